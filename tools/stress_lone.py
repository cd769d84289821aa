"""Development aid: random general models (several rate classes, susceptibility groups, immunity transitions, migration, NPIs, up to 1024
haplotypes and 64 demes) — ONE trajectory on vgx_lone.hip (kernel='lone') against the CPU oracle, bit for bit, with a continued call:
python tools/stress_lone.py [first] [n]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import helpers
from oracle import oracle


LONG = bool(os.environ.get("VGX_STRESS_LONG"))   # long runs on 256-4096 haplotypes: long lists, new heap layouts, the full-heap way out


def build(seed):
    from vgsim_amd import Simulator
    rng = np.random.default_rng(10 ** 6 + seed)
    sites = int(rng.integers(4, 7)) if LONG else int(rng.integers(0, 6))
    P = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 64]))
    S = int(rng.integers(1, 5))
    with helpers.quiet():
        s = Simulator(number_of_sites=sites, populations_number=P, number_of_susceptible_groups=S, seed=int(rng.integers(0, 2 ** 31)))
    H = 4 ** sites
    s.set_transmission_rate(float(rng.uniform(1.5, 4.0)))
    s.set_recovery_rate(float(rng.uniform(0.3, 1.2)))
    s.set_sampling_rate(float(rng.uniform(0.01, 0.4)))
    for _ in range(int(rng.integers(0, 5))):
        h = int(rng.integers(0, H))
        s.set_transmission_rate(float(rng.uniform(0.5, 5.0)), haplotype=h)
        if rng.random() < 0.5:
            s.set_recovery_rate(float(rng.uniform(0.2, 1.5)), haplotype=h)
        if rng.random() < 0.3:
            s.set_sampling_rate(float(rng.uniform(0.0, 0.5)), haplotype=h)
    if sites:
        s.set_mutation_rate(float(rng.choice([0.0, 0.01, 0.2, 0.8, 3.0])))
        if rng.random() < 0.5:
            s.set_mutation_rate(float(rng.uniform(0.0, 0.5)), mutation=int(rng.integers(0, sites)))
        if rng.random() < 0.3:
            s.set_mutation_rate(float(rng.uniform(0.0, 2.0)), haplotype=int(rng.integers(0, H)))
        if rng.random() < 0.5:
            w = [int(x) for x in rng.integers(0, 4, size=4)]
            if sum(w) - max(w) > 0 and all(sum(w) - w[i] > 0 for i in range(4)):
                s.set_mutation_probabilities(w)
    for g in range(1, S):
        s.set_susceptibility(float(rng.uniform(0.0, 1.0)), susceptibility_type=g)
        if rng.random() < 0.4:
            s.set_susceptibility(float(rng.uniform(0.0, 1.0)), susceptibility_type=g, haplotype=int(rng.integers(0, H)))
        if rng.random() < 0.7:
            s.set_immunity_transition(float(rng.uniform(0.0, 0.1)), source=g, target=int(rng.integers(0, S)))
    if S > 1:
        s.set_susceptibility_type(int(rng.integers(0, S)))
        for _ in range(2):
            s.set_susceptibility_type(int(rng.integers(0, S)), haplotype=int(rng.integers(0, H)))
    s.set_population_size(int(rng.integers(2000, 200000)))
    if P > 1:
        s.set_population_size(int(rng.integers(500, 5000)), population=int(rng.integers(0, P)))
        s.set_total_migration_probability(float(rng.uniform(0.0, 0.3)))
        if rng.random() < 0.5:
            a, b = (int(x) for x in rng.choice(P, size=2, replace=False))
            s.set_migration_probability(float(rng.uniform(0.0, 0.002)), source=a, target=b)
        s.set_contact_density(float(rng.uniform(0.5, 2.0)), population=int(rng.integers(0, P)))
        s.set_sampling_multiplier(float(rng.uniform(0.5, 3.0)), population=int(rng.integers(0, P)))
    for _ in range(int(rng.integers(0, 3))):
        start = float(rng.uniform(0.001, 0.05))
        s.set_npi([float(rng.uniform(0.0, 0.8)), start, float(rng.uniform(0.0, start))], population=int(rng.integers(0, P)))
    return s, int(rng.integers(300, 6000)) * (12 if LONG else 1)


def check(seed):
    hip, n = build(seed)
    ref, _ = build(seed)
    m = ref.simulation
    tag = "seed %d P %d H %d S %d n %d" % (seed, m.popNum, m.hapNum, m.susNum, n)
    rc = oracle.run_direct(m, n, 10 ** 9, -1, 200)
    try:
        with helpers.quiet():
            hip.simulate(n, sample_size=10 ** 9, kernel="lone")
    except Exception as ex:
        return rc != 0, tag + (": both abort" if rc != 0 else ": engine error, oracle ran: %s" % ex)
    if rc != 0:
        return False, tag + ": oracle rc %d, engine ran" % rc
    try:
        helpers.assert_models_equal(hip.simulation, m, tag)
        # a continued call on the same engine state (vgx_api.hip runs it on the row kernels: the lists are no longer fresh) must go on
        # from where the first one ended
        rc = oracle.run_direct(m, n // 2 + 1, 10 ** 9, -1, 200)
        if rc == 0:
            with helpers.quiet():
                hip.simulate(n // 2 + 1, sample_size=10 ** 9)
            helpers.assert_models_equal(hip.simulation, m, tag + " (continued)")
    except AssertionError as ex:
        return False, tag + ": MISMATCH " + str(ex)[:300]
    return True, tag + ": ok (%d events, kernel %s)" % (m.events.ptr, hip.simulation._engine.last_kernel)


if __name__ == "__main__":
    oracle.build()
    first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    bad = 0
    for seed in range(first, first + count):
        good, text = check(seed)
        print(text, flush=True)
        bad += 0 if good else 1
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
