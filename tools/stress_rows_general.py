"""Development aid: ensembles of the random general models of tests/test_hip_fuzz.py (susceptibility groups, rate classes, NPIs, migration) —
the general row kernel against the one-replicate-per-wavefront kernel on the same seeds, two launches: python tools/stress_rows_general.py [first] [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import helpers
import test_hip_fuzz as fz
from vgsim_amd.ensemble import Ensemble


def check(seed):
    sim, n = fz.build(seed)
    R = 6
    seeds = 7000 * (seed + 1) + np.arange(R, dtype=np.int64)
    out = {}
    for kernel in ("wave", "quad"):
        ens = Ensemble(sim, R, seeds=seeds)
        try:
            a = ens.simulate(n, sample_size=10 ** 9, record_events=True, kernel=kernel)
            b = ens.simulate(n // 2 + 1, sample_size=10 ** 9, record_events=True, kernel=kernel)
            out[kernel] = (a.events.copy(), b.events.copy(), [ens.replicate_events(r) for r in range(R)], [ens.replicate_state(r) for r in range(R)])
        except Exception as ex:
            out[kernel] = str(ex)
        ens.close()
    w, q = out["wave"], out["quad"]
    if isinstance(w, str) or isinstance(q, str):
        # a model outside the row kernels' scope is refused (recombination, more than 128 demes ...); an abort of the reference's own must be the same
        same = isinstance(w, str) and isinstance(q, str) and w == q or (isinstance(q, str) and "not" in q and not isinstance(w, str))
        return same, "seed %d: %s | %s" % (seed, w if isinstance(w, str) else "ran", q if isinstance(q, str) else "ran")
    ok = np.array_equal(w[0], q[0]) and np.array_equal(w[1], q[1])
    for r in range(R):
        ok = ok and w[2][r].shape == q[2][r].shape and np.array_equal(w[2][r], q[2][r])
        ok = ok and np.array_equal(w[3][r].infectious, q[3][r].infectious) and np.array_equal(w[3][r].susceptible, q[3][r].susceptible)
    return bool(ok), "seed %d P %d S %d: %s" % (seed, sim.simulation.popNum, sim.simulation.susNum, "ok" if ok else "MISMATCH")


if __name__ == "__main__":
    first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    bad = 0
    for seed in range(first, first + count):
        good, text = check(seed)
        print(text, flush=True)
        bad += 0 if good else 1
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
