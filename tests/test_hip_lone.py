"""The latency kernel for ONE trajectory of a large haplotype space (vgx_lone.hip: occupancy lists resident in LDS, the haplotype
choice a ballot over the stored prefix sums) against the CPU oracle, bit for bit: every case of the suite in its scope (one rate class,
one susceptibility group, <= 64 populations, no possible lockdown switch) with the kernel forced, the reference's goldens, ensembles
whose replicates must equal single seeded runs (with trajectories: the device clock), BASELINE config 3 at full size in both occupancy
regimes, a heap that is laid out again and again, and the hand-over to the row kernel when the lists outgrow the heap."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu

LONE_OK = ["c2", "g1", "g1_short", "g5", "g5_short", "g6", "g6_short", "g8", "g8_short", "c3_s5_p16", "c3_s6_p8_spread",
           "sample_stop", "time_stop", "extinct", "extinct_restart"]


@pytest.mark.parametrize("name", LONE_OK)
def test_lone_kernel_bit_exact_vs_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name, kernel="lone").simulation
    assert hip._engine.last_kernel == "lone"
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", [n for n in LONE_OK if n != "c2"])
def test_lone_kernel_vs_reference_goldens(name):
    hip = helpers.run_case_hip(name, kernel="lone").simulation
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), leftovers=False)


DIRECT = [n for n, (_, ph) in models.CASES.items() if all(kw.get("method", "direct") == "direct" for _, kw in ph)]
# the general form: every direct case of the suite without recombination at up to 64 populations (p70: outside)
LONE_GENERAL = [n for n in DIRECT if n not in models.RECOMBINATION_CASES and n not in LONE_OK and n != "p70"]


@pytest.mark.parametrize("name", LONE_GENERAL + [n for n in models.ORACLE_ONLY_CASES if n != "p70"])
def test_lone_general_form_bit_exact_vs_oracle(oracle_mod, name):
    """Several rate classes (the class rides in the list entry's haplotype word), several susceptibility groups (immunity transitions,
    the stale susceptHapPopRate of Birth), lockdown switches with UpdateAllRates, Restarts that keep lockdown records."""
    hip = helpers.run_case_hip(name, kernel="lone").simulation
    assert hip._engine.last_kernel == "lone"
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", ["g2", "g3", "g4", "g7", "g9", "example", "stress_h64", "stress_h256", "continuation", "cmd_example"])
def test_lone_general_form_vs_reference_goldens(name):
    hip = helpers.run_case_hip(name, kernel="lone").simulation
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), rtol_time=1e-12, leftovers=False)


def test_lone_refuses_models_outside_its_scope():
    from vgsim_amd._capi import VgxError
    with pytest.raises(VgxError), helpers.quiet():
        helpers.run_case_hip("p70", kernel="lone")
    with pytest.raises(VgxError), helpers.quiet():
        helpers.run_case_hip("recomb_a", kernel="lone")


def _single(oracle_mod, name, seed, n_events, mut=None):
    from vgsim_amd import Simulator
    ctor, phases = models.CASES[name]
    with helpers.quiet():
        one = Simulator(**dict(ctor, seed=int(seed)))
    phases[0][0](one)
    if mut is not None:
        one.set_mutation_rate(mut)
    m = one.simulation
    assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200) == 0
    return m


@pytest.mark.parametrize("name,n_events,mut", [("c3_s5_p16", 3000, None), ("g6_short", 4000, None), ("c3_s5_p16", 2500, 0.5),
                                               ("extinct_restart", 1000, None)])
def test_lone_replicates_equal_single_runs(oracle_mod, name, n_events, mut):
    """One wavefront per replicate, with summary trajectories (the CLOCK instantiation: SampleTime's logarithm on the device)."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 7
    with helpers.quiet():
        sim, phases = models.build(Simulator, name)
    phases[0][0](sim)
    if mut is not None:
        sim.set_mutation_rate(mut)
    seeds = np.array([3, 4, 2021, 99, 100000, 12345678901, 1], dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    T = 17
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, traj_points=T, traj_window=(0.0, 6.0), kernel="lone")
    assert ens.engine.last_kernel == "lone"
    traj = ens.trajectories()
    for r in range(R):
        m = _single(oracle_mod, name, seeds[r], n_events, mut)
        assert res.events[r] == m.events.ptr, "replicate %d" % r
        chain = ens.replicate_events(r)
        assert np.array_equal(chain, m.events.as_array()[:, :m.events.ptr]), "replicate %d: %s" % (
            r, helpers.describe_first_diff(chain, m.events.as_array(), m.events.ptr))
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible)
        assert st.currentTime == m.currentTime and st.good_attempt == m.good_attempt
        for k in st.COUNTERS:
            assert getattr(st, k) == getattr(m, k), k
        tot_i = m.initial_infectious.sum(axis=1).astype(float)
        tot_s = m.initial_susceptible.sum(axis=1).astype(float)
        grid = np.linspace(0.0, 6.0, T)
        want = np.zeros((T, m.popNum, 2))
        j = 0
        for e in range(m.events.ptr):
            while j < T and grid[j] < m.events.times[e]:
                want[j, :, 0], want[j, :, 1] = tot_i, tot_s
                j += 1
            ty, pop, npop = m.events.types[e], m.events.populations[e], m.events.newPopulations[e]
            if ty == 0:
                tot_i[pop] += 1; tot_s[pop] -= 1
            elif ty in (1, 2):
                tot_i[pop] -= 1; tot_s[pop] += 1
            elif ty == 5:
                tot_i[npop] += 1; tot_s[npop] -= 1
        while j < T:
            want[j, :, 0], want[j, :, 1] = tot_i, tot_s
            j += 1
        if res.restarts[r] == 0:
            assert np.array_equal(traj[r], want), "trajectory of replicate %d" % r
    ens.close()


def _c3(seed, mut=0.01):
    from vgsim_amd import Simulator
    with helpers.quiet():
        s = Simulator(number_of_sites=8, populations_number=64, number_of_susceptible_groups=1, seed=seed)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1)
    s.set_mutation_rate(mut); s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    return s


@pytest.mark.parametrize("seed,mut,n,kernel", [(2020, 0.01, 40000, "lone"), (2021, 0.4, 12000, "lone"), (2020, 0.01, 40000, "auto")])
def test_lone_config3_bit_exact_vs_sparse_oracle(oracle_mod, seed, mut, n, kernel):
    """BASELINE config 3 at full size (65 536 haplotypes x 64 populations); with kernel='auto' a single Simulator.simulate() must
    take this kernel by itself."""
    hip = _c3(seed, mut)
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9, kernel=kernel)
    assert hip.simulation._engine.last_kernel == "lone"
    ref = _c3(seed, mut).simulation
    assert oracle_mod.run_direct(ref, n, 10 ** 9, -1, 200, sparse=True) == 0
    helpers.assert_models_equal(hip.simulation, ref, "config3 seed %d" % seed)
    assert (hip.simulation.infectious != 0).sum() > (50 if mut < 0.1 else 1000)


def _spread_state(sim, occ, rng):
    m = sim.simulation
    for pn in range(m.popNum):
        haps = rng.choice(m.hapNum, size=occ[pn], replace=False)
        if pn == 4:
            haps[0] = m.hapNum - 1          # the last haplotype occupied: the clamp of fastChoose is a valid pick there
        m.infectious[pn, haps] = rng.integers(1, 4, size=occ[pn])
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())


def test_lone_config3_long_lists_vs_sparse_oracle_and_row_kernel(oracle_mod):
    """Lists of up to 1000 entries (many tiles: the two-level choice, multi-tile shifts), a heap that is three quarters full at the
    start: replicate 0 against the oracle, all against the row kernel."""
    from vgsim_amd.ensemble import Ensemble
    sim = _c3(2020)
    m = sim.simulation
    rng = np.random.default_rng(5)
    occ = [700, 3, 64, 65, 130, 17, 16, 1000] + [int(v) for v in rng.integers(1, 90, size=56)]
    _spread_state(sim, occ, rng)
    m.set_mutation_rate(0.05, None, None)
    R, N = 3, 3000
    seeds = 900 + np.arange(R, dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    rq = ens.simulate(N, sample_size=10 ** 9, record_events=True, kernel="lone")
    assert ens.engine.last_kernel == "lone"
    chains = [ens.replicate_events(r) for r in range(R)]
    states = [ens.replicate_state(r) for r in range(R)]
    rw = ens.simulate(N, sample_size=10 ** 9, record_events=True, kernel="quad")
    for r in range(R):
        assert rq.events[r] == rw.events[r] and rq.loop_iterations[r] == rw.loop_iterations[r]
        assert np.array_equal(chains[r], ens.replicate_events(r)), "replicate %d differs from the row kernel: %s" % (
            r, helpers.describe_first_diff(chains[r], ens.replicate_events(r), rq.events[r]))
        sw = ens.replicate_state(r)
        assert np.array_equal(states[r].infectious, sw.infectious) and states[r].currentTime == sw.currentTime
    ens.close()
    import copy
    ref = copy.copy(m)
    for name in ("susceptible", "infectious", "initial_susceptible", "initial_infectious", "totalSusceptible", "totalInfectious",
                 "lockdownON", "contactDensity"):
        setattr(ref, name, getattr(m, name).copy())
    ref.events = type(m.events)()
    ref.user_seed = int(seeds[0])
    assert oracle_mod.run_direct(ref, N, 10 ** 9, -1, 200, sparse=True) == 0
    assert np.array_equal(chains[0], ref.events.as_array()[:, :ref.events.ptr]), helpers.describe_first_diff(
        chains[0], ref.events.as_array(), ref.events.ptr)
    assert np.array_equal(states[0].infectious, ref.infectious)


def test_lone_full_heap_hands_the_call_to_the_row_kernel(oracle_mod):
    """High mutation rate: the lists outgrow the LDS heap in mid-call.  Chosen automatically, the call then runs again on the row kernel
    from the same state and ends with the oracle's result; forced, it ends with a capacity error."""
    from vgsim_amd._capi import VgxError
    n = 60000
    hip = _c3(2021, 0.4)
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9)
    assert hip.simulation._engine.last_kernel == "quad"
    assert (hip.simulation.infectious != 0).sum() > 7600          # more entries than the heap has slots
    ref = _c3(2021, 0.4).simulation
    assert oracle_mod.run_direct(ref, n, 10 ** 9, -1, 200, sparse=True) == 0
    helpers.assert_models_equal(hip.simulation, ref, "config3 after a full heap")
    with pytest.raises(VgxError, match="capacity"), helpers.quiet():
        _c3(2021, 0.4).simulate(n, sample_size=10 ** 9, kernel="lone")


def test_lone_continued_calls(oracle_mod):
    """Two slices through the C ABI on device-resident state with the kernel forced (the first without a device clock, the second with a
    time limit): == the oracle's continuation."""
    import ctypes as C
    from vgsim_amd import Simulator, _capi
    ctor, phases = models.CASES["g5_short"]
    seeds = np.array([5, 6, 2020], dtype=np.int64)
    R, cap, s1, t2 = len(seeds), 500000, 1, 7.5
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    m = sim.simulation
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=R)
    m.events.CreateEvents(cap)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(seeds)
    o = _capi.VgxRunOpts(); o.record_events = 1
    o.kernel = 6
    eng._check(eng.lib.vgx_simulate_direct(eng.handle, cap, s1, -1.0, 200, C.byref(o)))
    first_ptr = [eng.counters(r).ev_ptr for r in range(R)]
    eng._check(eng.lib.vgx_simulate_direct(eng.handle, cap, 10 ** 9, t2, 200, C.byref(o)))
    assert eng.lib.vgx_last_direct_kernel(eng.handle) == 6
    for r in range(R):
        with helpers.quiet():
            one = Simulator(**dict(ctor, seed=int(seeds[r])))
        phases[0][0](one)
        om = one.simulation
        assert oracle_mod.run_direct(om, cap, s1, -1, 200) == 0
        p1 = om.events.ptr
        assert oracle_mod.run_direct(om, cap, 10 ** 9, t2, 200) == 0
        c = eng.counters(r)
        assert first_ptr[r] == p1 and c.ev_first_new == p1 and c.ev_ptr == om.events.ptr
        n = c.ev_ptr - p1
        times = np.zeros(n); cols = [np.zeros(n, dtype=np.int64) for _ in range(5)]
        eng._check(eng.lib.vgx_get_events(eng.handle, r, p1, n, _capi._p(times), *[_capi._p(x) for x in cols]))
        ref = om.events.as_array()[:, p1:om.events.ptr]
        assert np.array_equal(times, ref[0]), "replicate %d: times of the second slice" % r
        for k in range(5):
            assert np.array_equal(cols[k], ref[k + 1].astype(np.int64))
    eng.close()
