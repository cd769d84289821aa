"""Adds ``sha256_ints`` (sha256 of rows 1-5 of the full (6, N) float64 chain) to the metadata of every direct-path fixture.

The fixtures were recorded from the reference (make_golden.py) with ``sha256_chain`` = sha256 of the WHOLE chain, which
includes the libm-dependent time row, so a host whose ``log`` differs in the last bit cannot use it to compare the full
integer rows of the 100 000-event goldens (only their first/last 256 columns are stored).  This script re-runs each case
on the oracle, REQUIRES the oracle's chain to hash to the recorded ``sha256_chain`` (i.e. to be the reference's chain,
byte for byte) and only then stores the hash of its integer rows.  Run where the libm probe holds (the container the
fixtures were recorded in):   python tests/golden/add_int_hashes.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import helpers  # noqa: E402
import models  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    assert helpers.libm_matches_fixture_host(), "this host's log() differs from the fixture host's: cannot certify the chains"
    oracle.build()
    done = 0
    for name in models.CASES:
        path = os.path.join(HERE, name + ".npz")
        if not os.path.exists(path):
            continue
        z = dict(np.load(path))
        meta = json.loads(str(z["meta"]))
        if "sha256_chain" not in meta:
            continue
        sim = helpers.run_case_oracle(oracle, name)
        chain = np.ascontiguousarray(sim.simulation.events.as_array())
        got = hashlib.sha256(chain.tobytes()).hexdigest()
        if got != meta["sha256_chain"]:
            print("skip %s: the oracle's chain does not hash to the recorded reference chain" % name)
            continue
        meta["sha256_ints"] = hashlib.sha256(np.ascontiguousarray(chain[1:]).tobytes()).hexdigest()
        z["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(path, **z)
        done += 1
        print("%-20s sha256_ints %s" % (name, meta["sha256_ints"][:16]))
    print("%d fixtures updated" % done)


if __name__ == "__main__":
    main()
