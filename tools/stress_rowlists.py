"""Development aid: random start states with lists around the tile boundaries (63, 64, 65, 127 ... entries), random mutation rates — the
row kernels (exact: long-list form with zero-count entries; FAST) against the one-replicate-per-wavefront kernel on the same seeds:
python tools/stress_rowlists.py [first_seed] [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import helpers
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble

edges = [1, 2, 15, 16, 17, 63, 64, 65, 66, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300, 511, 513]


def check(seed):
    """True when the row kernels reproduce the wave kernel on the random start state of `seed` (a line of text with it)."""
    rng = np.random.default_rng(seed)
    sites = int(rng.integers(5, 9))
    P = int(rng.choice([1, 3, 7, 16, 33, 64]))
    with helpers.quiet():
        sim = Simulator(number_of_sites=sites, populations_number=P, number_of_susceptible_groups=1, seed=seed)
    sim.set_transmission_rate(float(rng.choice([1.0, 2.5]))); sim.set_recovery_rate(0.9); sim.set_sampling_rate(0.1)
    sim.set_mutation_rate(float(rng.choice([0.01, 0.2, 0.6])))
    if P > 1:
        sim.set_total_migration_probability(float(rng.choice([0.01, 0.2])))
    sim.set_population_size(10 ** 6)
    m = sim.simulation
    H = m.hapNum
    for pn in range(P):
        occ = min(int(rng.choice(edges)), H)
        haps = rng.choice(H, size=occ, replace=False)
        if rng.random() < 0.3:
            haps[0] = H - 1
        m.infectious[pn, haps] = rng.integers(1, int(rng.choice([2, 4, 400])), size=occ)
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
    R, N = 6, int(rng.choice([500, 3000, 8000]))
    seeds = 1000 * seed + np.arange(R, dtype=np.int64)
    out = {}
    for tag, kernel, mode in (("wave", "wave", "exact"), ("quad", "quad", "exact"), ("fast", "quad", "fast")):
        ens = Ensemble(sim, R, seeds=seeds)
        try:
            res = ens.simulate(N, sample_size=10 ** 9, record_events=True, kernel=kernel, mode=mode)
            res2 = ens.simulate(N // 2, sample_size=10 ** 9, record_events=True, kernel=kernel, mode=mode)      # a second launch on the settled lists
            out[tag] = (res.events.copy(), res2.events.copy(), [ens.replicate_events(r) for r in range(R)], [ens.replicate_state(r) for r in range(R)])
        except Exception as ex:      # the reference's own abort (zero weight in fastChoose): every kernel must report the same
            out[tag] = str(ex)
        ens.close()
    if any(isinstance(v, str) for v in out.values()):
        same = len({str(v) if isinstance(v, str) else "ran" for v in out.values()}) == 1
        return same, "seed %d sites %d P %d N %d: %s %s" % (seed, sites, P, N, "ok (all kernels abort alike)" if same else "MISMATCH (abort)",
                                                              {k: (v if isinstance(v, str) else "ran") for k, v in out.items()})
    ok = True
    for tag in ("quad", "fast"):
        ok &= np.array_equal(out[tag][0], out["wave"][0]) and np.array_equal(out[tag][1], out["wave"][1])
        for r in range(R):
            a, b = out[tag][2][r], out["wave"][2][r]
            ok &= a.shape == b.shape and np.array_equal(a[1:], b[1:])
            if tag == "quad":
                ok &= np.array_equal(a[0], b[0])
            ok &= np.array_equal(out[tag][3][r].infectious, out["wave"][3][r].infectious)
    return bool(ok), "seed %d sites %d P %d N %d: %s" % (seed, sites, P, N, "ok" if ok else "MISMATCH")


if __name__ == "__main__":
    first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    bad = 0
    for seed in range(first, first + count):
        good, text = check(seed)
        print(text, flush=True)
        bad += 0 if good else 1
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
