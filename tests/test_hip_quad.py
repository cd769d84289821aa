"""The four-replicates-per-wavefront kernel (vgx_quad.hip: one replicate per 16-lane DPP row) against the CPU oracle, bit
for bit: every eligible case of the suite (one rate class, one susceptibility group, <= 64 populations, no possible
lockdown switch) with the kernel forced, ensembles whose replicates must equal single seeded runs (rows of one wavefront
diverge: different event types, list lengths, restarts), and BASELINE config 3 at full size in both occupancy regimes."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu

QUAD_OK = ["c2", "g1", "g1_short", "g5", "g5_short", "g6", "g6_short", "g8", "g8_short", "c3_s5_p16", "c3_s6_p8_spread",
           "sample_stop", "time_stop", "extinct", "extinct_restart"]


@pytest.mark.parametrize("name", QUAD_OK)
def test_quad_kernel_bit_exact_vs_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name, kernel="quad").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", [n for n in QUAD_OK if n != "c2"])
def test_quad_kernel_vs_reference_goldens(name):
    hip = helpers.run_case_hip(name, kernel="quad").simulation
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), leftovers=False)


def test_quad_hands_models_outside_its_scope_to_the_general_form(oracle_mod):
    """Models the one-class form does not take go to the general form (vgx_quadg.hip, tests/test_hip_quadg.py) — since round 4 also
    those with a recombination probability; FAST mode with recombination stays refused."""
    from vgsim_amd._capi import VgxError
    hip = helpers.run_case_hip("recomb_pos", kernel="quad").simulation
    assert hip._engine.last_kernel == "quadg"
    helpers.assert_models_equal(hip, helpers.run_case_oracle(oracle_mod, "recomb_pos").simulation, "recomb_pos")
    with pytest.raises(VgxError), helpers.quiet():
        helpers.run_case_hip("recomb_pos", mode="fast")


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
def test_quad_replicates_equal_single_runs(oracle_mod, name, n_events, mut):
    """Rows of one wavefront run different trajectories (also: fewer replicates than rows in the last wavefront)."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 11
    with helpers.quiet():
        sim, phases = models.build(Simulator, name)
    phases[0][0](sim)
    if mut is not None:
        sim.set_mutation_rate(mut)
    seeds = np.array([3, 4, 5, 6, 7, 2021, 2022, 99, 100000, 12345678901, 1], dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    T = 17
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, traj_points=T, traj_window=(0.0, 6.0), kernel="quad")
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
        # trajectories: totals per population just before each grid time (replay of the oracle's log)
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
        if res.restarts[r] == 0:     # a Restart rewinds the grid (same rule as the wave kernel); compared without
            assert np.array_equal(traj[r], want), "trajectory of replicate %d" % r
    ens.close()


def _c3(seed, mut=0.01):
    from vgsim_amd import Simulator
    with helpers.quiet():
        s = Simulator(number_of_sites=8, populations_number=64, number_of_susceptible_groups=1, seed=seed)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1)
    s.set_mutation_rate(mut); s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    return s


@pytest.mark.parametrize("seed,mut,n", [(2020, 0.01, 40000), (2021, 0.4, 12000)])
def test_quad_config3_bit_exact_vs_sparse_oracle(oracle_mod, seed, mut, n):
    hip = _c3(seed, mut)
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9, kernel="quad")
    ref = _c3(seed, mut).simulation
    assert oracle_mod.run_direct(ref, n, 10 ** 9, -1, 200, sparse=True) == 0
    helpers.assert_models_equal(hip.simulation, ref, "config3 seed %d" % seed)


def test_quad_config3_spread_occupancy_vs_sparse_oracle_and_wave_kernel(oracle_mod):
    """Long occupancy lists (hundreds of entries per population, different lengths in the four rows): the chunked chain,
    the tile sums, 16-ary lower bound, insertions and removals with shifts.  Replicate 0 against the oracle, all
    replicates against the one-replicate-per-wavefront kernel."""
    from vgsim_amd.ensemble import Ensemble
    sim = _c3(2020)
    m = sim.simulation
    rng = np.random.default_rng(5)
    occ = [700, 3, 64, 65, 130, 17, 16, 1000] + [int(v) for v in rng.integers(1, 400, size=56)]
    for pn in range(64):
        haps = rng.choice(m.hapNum, size=occ[pn], replace=False)
        if pn == 4:
            haps[0] = m.hapNum - 1          # the last haplotype occupied: the clamp of fastChoose is a valid pick there
        m.infectious[pn, haps] = rng.integers(1, 4, size=occ[pn])
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
    m.set_mutation_rate(0.05, None, None)
    R, N = 6, 3000
    seeds = 900 + np.arange(R, dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    rq = ens.simulate(N, sample_size=10 ** 9, record_events=True, kernel="quad")
    chains = [ens.replicate_events(r) for r in range(R)]
    states = [ens.replicate_state(r) for r in range(R)]
    rw = ens.simulate(N, sample_size=10 ** 9, record_events=True, kernel="wave")
    for r in range(R):
        assert rq.events[r] == rw.events[r] and rq.loop_iterations[r] == rw.loop_iterations[r]
        assert np.array_equal(chains[r], ens.replicate_events(r)), "replicate %d differs from the wave kernel" % r
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
