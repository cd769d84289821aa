"""Seeded random models (GPU): shapes, rate heterogeneity, mutation weights, susceptibility groups, immunity
transitions, migration matrices, sampling multipliers and lockdown thresholds drawn at random through the public
setters; the HIP engine must reproduce the oracle bit for bit in EXACT mode and on the integer columns in FAST mode.
Finds the corner cases hand-written models miss (empty populations, zero rates, single-group models, populations
beyond one 64-lane tile)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def build(seed):
    from vgsim_amd import Simulator
    rng = np.random.default_rng(seed)
    sites = int(rng.integers(0, 4))
    P = int(rng.choice([1, 2, 3, 5, 9, 66]))
    S = int(rng.integers(1, 4))
    with helpers.quiet():
        s = Simulator(number_of_sites=sites, populations_number=P, number_of_susceptible_groups=S, seed=int(rng.integers(0, 2 ** 31)))
    H = 4 ** sites
    s.set_transmission_rate(float(rng.uniform(1.5, 4.0)))
    s.set_recovery_rate(float(rng.uniform(0.3, 1.2)))
    s.set_sampling_rate(float(rng.uniform(0.01, 0.4)))
    for _ in range(int(rng.integers(0, 4))):          # a few haplotypes with their own rates -> several rate classes
        h = int(rng.integers(0, H))
        s.set_transmission_rate(float(rng.uniform(0.5, 5.0)), haplotype=h)
        if rng.random() < 0.5:
            s.set_recovery_rate(float(rng.uniform(0.2, 1.5)), haplotype=h)
    if sites:
        s.set_mutation_rate(float(rng.choice([0.0, 0.01, 0.2, 0.8])))
        if rng.random() < 0.5:
            s.set_mutation_rate(float(rng.uniform(0.0, 0.5)), mutation=int(rng.integers(0, sites)))
        if rng.random() < 0.5:
            w = [int(x) for x in rng.integers(0, 4, size=4)]
            if sum(w) - max(w) > 0 and all(sum(w) - w[i] > 0 for i in range(4)):
                s.set_mutation_probabilities(w)
    for g in range(1, S):
        s.set_susceptibility(float(rng.uniform(0.0, 1.0)), susceptibility_type=g)
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
    return s, int(rng.integers(300, 4000))


def drive_oracle(oracle_mod, sim, n):
    rc = oracle_mod.run_direct(sim.simulation, n, 10 ** 9, -1, 200)
    return rc


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("VGX_FUZZ_SEEDS", "40")))))
def test_random_model(oracle_mod, seed):
    hip, n = build(seed)
    ref, _ = build(seed)
    fast, _ = build(seed)
    rc = drive_oracle(oracle_mod, ref, n)
    if rc != 0:   # zero-weight abort of the reference (fast_choose.pxi:5-13): the engine must report it too
        with pytest.raises(RuntimeError):
            with helpers.quiet():
                hip.simulate(n, sample_size=10 ** 9)
        return
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9)
        fast.simulate(n, sample_size=10 ** 9, mode="fast")
    helpers.assert_models_equal(hip.simulation, ref.simulation, "fuzz %d" % seed)
    m0 = ref.simulation
    if m0.popNum <= 16 and m0.popNum * m0.hapNum <= 1024:   # the one-replicate-per-lane kernel on the same model
        lane, _ = build(seed)
        with helpers.quiet():
            lane.simulate(n, sample_size=10 ** 9, kernel="lane")
        helpers.assert_models_equal(lane.simulation, ref.simulation, "fuzz %d (lane kernel)" % seed)
    ptr = ref.simulation.events.ptr
    a, b = helpers.chain_of(fast.simulation), helpers.chain_of(ref.simulation)
    assert fast.simulation.events.ptr == ptr
    assert np.array_equal(a[1:, :ptr], b[1:, :ptr]), "fast: " + helpers.describe_first_diff(a[1:], b[1:], ptr)
    np.testing.assert_allclose(a[0, :ptr], b[0, :ptr], rtol=1e-9, atol=0.0)
    assert np.array_equal(fast.simulation.infectious, ref.simulation.infectious)
    # the backward pass continues the simulation's stream from the position the kernel reports
    if ref.simulation.sCounter >= 2:
        want = oracle_mod.run_genealogy(ref.simulation, None)
        if want["rc"] == 0:   # several roots (lineages that never coalesce) are outside what the pass defines
            with helpers.quiet():
                hip.genealogy(None)
            assert np.array_equal(hip.simulation.tree, want["tree"]) and np.array_equal(hip.simulation.times, want["times"])
            assert hip.simulation.mig.nodeId == want["mig_node"].tolist() and hip.simulation.mut.nodeId == want["mut_node"].tolist()


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("VGX_FUZZ_SEEDS", "40")))))
def test_random_model_tau_invariants(seed):
    """Tau-leaping on the same random models (after a short direct phase): the distribution is checked elsewhere;
    here every run must keep hosts conserved per population, compartments within [0, size], counters equal to the sums
    of the multievent rows it logged, and totals consistent."""
    sim, n = build(seed)
    m = sim.simulation
    with helpers.quiet():
        sim.simulate(min(n, 600), sample_size=10 ** 9)
    if m.globalInfectious == 0:
        return
    before = {k: getattr(m, k) for k in ("bCounter", "dCounter", "sCounter", "mCounter", "iCounter", "migPlus")}
    mv0 = m.multievents.ptr
    try:
        with helpers.quiet():
            sim.simulate(25, sample_size=10 ** 12, method="tau")
    except Exception as ex:
        if "tau underflow in the halving loop" not in str(ex) or m.popNum == 1:
            raise
        # Upstream's dead end: a source compartment that passed the check on the strength of its booked migrants (pyx:2473) was
        # applied below zero (pyx:2548), after which no try can pass and upstream halves tau for ever; the engine gives up after
        # 200 halvings.  That, and nothing else, must be what happened: the same call cut off after the steps it accepted (the
        # error names the step that could not be drawn) ends with such a compartment.
        import re
        k = int(re.search(r"\(step (\d+),", str(ex)).group(1))
        assert k > 0, "the halving loop gave up on the first step of the call: %s" % ex
        sim2, _ = build(seed)
        m2 = sim2.simulation
        with helpers.quiet():
            sim2.simulate(min(n, 600), sample_size=10 ** 9)
            sim2.simulate(k, sample_size=10 ** 12, method="tau")
        assert (m2.infectious < 0).any(), "the halving loop gave up although no compartment had been applied below zero"
        pytest.skip("seed %d runs into upstream's dead end (a compartment applied below zero: pyx:2473 vs 2548)" % seed)
    # upstream books a migrant on its SOURCE compartment in the bounds check (pyx:2473) but applies it to the target
    # (pyx:2548), so with migration a source compartment can pass the check and still end below zero: faithfully
    # reproduced (the oracle is pinned draw-exact on it), hence non-negativity is only an invariant without migration
    if m.popNum == 1:
        assert (m.infectious >= 0).all() and (m.susceptible >= 0).all()
    assert np.array_equal(m.susceptible.sum(axis=1) + m.infectious.sum(axis=1), m.sizes)
    assert m.globalInfectious == m.infectious.sum() and np.array_equal(m.totalInfectious, m.infectious.sum(axis=1))
    mv = m.multievents
    num, typ = mv.num[mv0:mv.ptr], mv.types[mv0:mv.ptr]
    assert (num > 0).all()
    for t, key in ((0, "bCounter"), (1, "dCounter"), (2, "sCounter"), (3, "mCounter"), (4, "iCounter"), (5, "migPlus")):
        assert int(num[typ == t].sum()) == getattr(m, key) - before[key], key
    ev = m.events
    multi = np.nonzero(ev.types[:ev.ptr] == 6)[0]
    if len(multi):   # contiguous, ordered row ranges
        lo, hi = ev.haplotypes[multi], ev.populations[multi]
        assert (hi >= lo).all() and (lo[1:] == hi[:-1]).all() and hi[-1] == mv.ptr


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("VGX_FUZZ_SEEDS", "40")))))
def test_random_model_tau_modes_agree(seed, monkeypatch):
    """(Step kernels of vgx_tau.hip, forced for these small models: VGX_TAU_STEP_KERNELS=1.)  The three ways a try of the halving loop can keep and check its deltas (vgx_run_opts.reserved[1]: list of moves /
    dense arrays with fused checks / dense arrays and a dense bounds check) make the same draws and the same decisions:
    bit for bit the same state, counters and multievent rows on every random model (shapes with rows shorter than a wave
    tile, several rate classes, several susceptibility groups, migration, lockdowns)."""
    from vgsim_amd import _capi
    monkeypatch.setenv("VGX_TAU_STEP_KERNELS", "1")
    out = []
    for mode in (0, 2, 1):
        sim, n = build(seed)
        m = sim.simulation
        with helpers.quiet():
            sim.simulate(min(n, 600), sample_size=10 ** 9)
            if m.globalInfectious == 0:
                return
            o = _capi.VgxRunOpts(); o.record_events = 1; o.reserved[1] = mode
            m.events.CreateEvents(25); m.events.CreateEvents(25); m.CheckSizes()
            m._get_engine().simulate_tau(m, 25, 10 ** 12, -1.0, 200, o)
        mv = m.multievents
        rows = np.stack([mv.num[:mv.ptr], mv.types[:mv.ptr], mv.haplotypes[:mv.ptr], mv.populations[:mv.ptr],
                         mv.newHaplotypes[:mv.ptr], mv.newPopulations[:mv.ptr]])
        # rows of one step are appended by concurrent wavefronts: compare them as a multiset per step
        out.append((m.infectious.copy(), m.susceptible.copy(), m.currentTime, [int(getattr(m, k)) for k in m.COUNTERS],
                    int(m.events.ptr), sorted(map(tuple, rows.T.tolist()))))
    for other in out[1:]:
        assert np.array_equal(out[0][0], other[0]) and np.array_equal(out[0][1], other[1])
        assert out[0][2] == other[2] and out[0][3] == other[3] and out[0][4] == other[4]
        assert out[0][5] == other[5]


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("VGX_FUZZ_SEEDS", "40")))))
def test_random_model_recombination_kernels_agree(oracle_mod, seed):
    """Recombinant births (pyx:575-596) on the random models with two sites or more: the occupancy-list kernel and the
    dense lane kernel give the oracle's log, records and state bit for bit."""
    sim0, n = build(seed)
    if sim0.simulation.sites < 2:
        return
    rng = np.random.default_rng(1000 + seed)
    prob = float(rng.uniform(0.05, 0.6))
    out = []
    for kernel in ("wave", "lane", None):
        sim, n = build(seed)
        m = sim.simulation
        m.recombination = prob
        with helpers.quiet():
            if kernel is None:
                oracle_mod.run_direct(m, min(n, 1500), 10 ** 9, -1, 200)
            else:
                sim.simulate(min(n, 1500), sample_size=10 ** 9, kernel=kernel)
        out.append(m)
    helpers.assert_models_equal(out[0], out[2], "fuzz %d recombination, wave kernel" % seed)
    helpers.assert_models_equal(out[1], out[2], "fuzz %d recombination, lane kernel" % seed)
