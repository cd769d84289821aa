#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REFERENCE itself.

Usage (in the development container only; /root/reference does not exist on the GPU box):
    tests/golden/build_reference.sh /tmp/vgsim_ref_build
    PYTHONPATH=/tmp/vgsim_ref_build:/tmp/vgsim_ref_build/stubs python3 tests/golden/make_golden.py [case ...]

For every case of tests/models.py the reference's ``VGsim.Simulator`` is driven through the same
setter/simulate calls the parity tests use, and what the reference produced is recorded:
  * the event chain exactly as ``export_chain_events`` saves it ((6, size) float64): in full when it has
    at most FULL_CHAIN_LIMIT columns, otherwise its sha256 plus the first and last 256 columns;
  * the counters and scalars ``Stats`` prints (parsed from the reference's own stdout);
  * the final ``susceptible`` / ``infectious`` arrays.
Only data is written (inputs are the parameter calls in tests/models.py, outputs the arrays above);
no reference source travels.  See build_reference.sh for the mc_lib caveat (seed->stream: parity unpinned).
"""
import contextlib
import hashlib
import io
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import models  # noqa: E402  (tests/models.py)

import VGsim  # noqa: E402  (the reference build; see usage)

STAT_KEYS = {
    "Number of samples": "sCounter", "Total number of iterations": "ptr", "Success number": "good_attempt",
    "Epidemic time": "currentTime", "Number of infections": "bCounter", "Number of recoveries": "dCounter",
    "Number of mutations": "mCounter", "Number of accepted migrations": "migPlus",
    "Number of rejected migrations": "migNonPlus", "Number of immunity transitions": "iCounter",
}


def run_case(name):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sim, phases = models.build(VGsim.Simulator, name)
        for setup, kw in phases:
            setup(sim)
            sim.simulate(**kw)
        tmp = "/tmp/_golden_chain_%s" % name
        sim.export_chain_events(tmp)
    chain = np.load(tmp + ".npy")
    os.remove(tmp + ".npy")
    out = buf.getvalue()
    stats = {}
    # the LAST Stats block describes the final state
    for line in out.splitlines():
        m = re.match(r"^([A-Za-z ]+):\s*(\S+)$", line.strip())
        if m and m.group(1) in STAT_KEYS:
            key = STAT_KEYS[m.group(1)]
            stats[key] = float(m.group(2)) if key == "currentTime" else int(m.group(2))
    messages = [l for l in out.splitlines() if l.startswith(("Achieved", "Simulation finished"))]
    ptr = stats["ptr"]
    sus = np.asarray(sim.simulation.susceptible).astype(np.int64)
    inf = np.asarray(sim.simulation.infectious).astype(np.int64)
    meta = dict(case=name, size=int(chain.shape[1]), stats=stats, messages=messages,
                sha256_chain=hashlib.sha256(np.ascontiguousarray(chain).tobytes()).hexdigest(),
                sha256_infectious=hashlib.sha256(np.ascontiguousarray(inf).tobytes()).hexdigest(),
                numpy=np.__version__)
    arrays = dict(susceptible=sus)
    nz = np.argwhere(inf != 0)
    arrays["infectious_nz"] = np.concatenate([nz, inf[inf != 0][:, None]], axis=1).astype(np.int64)
    if chain.shape[1] <= models.FULL_CHAIN_LIMIT:
        arrays["times"] = chain[0]
        arrays["ints"] = chain[1:].astype(np.int64).astype(np.int32)
        assert (arrays["ints"].astype(float) == chain[1:]).all()
    else:
        arrays["head"] = chain[:, :256]
        arrays["tail"] = chain[:, ptr - 256:ptr] if ptr >= 256 else chain[:, :ptr]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), **arrays)
    print("%-18s ptr=%-8d size=%-8d sha=%s  attempts=%s  t=%r" % (
        name, ptr, chain.shape[1], meta["sha256_chain"][:16], stats.get("good_attempt"), stats.get("currentTime")))
    hist = np.bincount(chain[1, :ptr].astype(int), minlength=7).tolist()
    print("                   types", hist, "rejected", stats.get("migNonPlus"))


if __name__ == "__main__":
    names = sys.argv[1:] or list(models.CASES)
    for n in names:
        run_case(n)
