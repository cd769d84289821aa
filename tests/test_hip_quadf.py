"""FAST mode on the row-per-replicate layout (vgx_quadf.hip: four replicates per wavefront, order-free sums — infection rate of a
population = tEvent x totalInfectious, haplotype by integer prefix search, factored BirthRate, tree prefix sums over the
populations) against the oracle on the same PCG64 stream: integer columns of the log, counters and compartments identical,
times within 1e-9 (the tolerance of FAST mode, tests/test_hip_fast.py), on every case the one-class row kernel takes, on
ensembles whose rows diverge, and on BASELINE config 3 in both occupancy regimes."""
import numpy as np
import pytest

import helpers
import models
from test_hip_fast import _assert_tier_b, RTOL_TIME
from test_hip_quad import QUAD_OK, _c3

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", QUAD_OK)
def test_fast_row_kernel_integer_columns_match_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name, mode="fast", kernel="quad").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    _assert_tier_b(hip, ref, name)


def test_fast_row_kernel_refuses_general_models():
    from vgsim_amd._capi import VgxError
    with pytest.raises(VgxError), helpers.quiet():
        helpers.run_case_hip("g9_short", mode="fast", kernel="quad")


@pytest.mark.parametrize("name,n_events,mut", [("c3_s5_p16", 3000, None), ("g6_short", 4000, None), ("c3_s5_p16", 2500, 0.5), ("extinct_restart", 1000, None)])
def test_fast_row_kernel_replicates_equal_single_oracle_runs(oracle_mod, name, n_events, mut):
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 11
    ctor, phases = models.CASES[name]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    if mut is not None:
        sim.set_mutation_rate(mut)
    seeds = np.array([3, 4, 5, 6, 7, 2021, 2022, 99, 100000, 12345678901, 1], dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, kernel="quad", mode="fast")
    for r in range(R):
        with helpers.quiet():
            one = Simulator(**dict(ctor, seed=int(seeds[r])))
        phases[0][0](one)
        if mut is not None:
            one.set_mutation_rate(mut)
        m = one.simulation
        assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200) == 0
        assert res.events[r] == m.events.ptr, "replicate %d" % r
        chain = ens.replicate_events(r)
        want = m.events.as_array()[:, :m.events.ptr]
        assert np.array_equal(chain[1:], want[1:]), "replicate %d: %s" % (r, helpers.describe_first_diff(chain[1:], want[1:], m.events.ptr))
        np.testing.assert_allclose(chain[0], want[0], rtol=RTOL_TIME, atol=0.0)
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible)
        assert st.good_attempt == m.good_attempt
        for k in st.COUNTERS:
            assert getattr(st, k) == getattr(m, k), k
    ens.close()


@pytest.mark.parametrize("seed,mut,n", [(2020, 0.01, 40000), (2021, 0.4, 12000)])
def test_fast_row_kernel_config3_vs_sparse_oracle(oracle_mod, seed, mut, n):
    """Config 3 at full size: natural occupancy and a high mutation rate (lists of hundreds of entries: tile sums)."""
    hip = _c3(seed, mut)
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9, kernel="quad", mode="fast")
    ref = _c3(seed, mut).simulation
    assert oracle_mod.run_direct(ref, n, 10 ** 9, -1, 200, sparse=True) == 0
    _assert_tier_b(hip.simulation, ref, "config3 seed %d" % seed)


def test_fast_row_kernel_spread_occupancy_zero_count_entries_and_continuation():
    """Long occupancy lists with many counts of 1 (removals leave zero-count entries, insertions ripple to the next free slot
    across tiles, the lists are settled after every launch): integer columns and the final compartments equal the exact row
    kernel's over three launches, the middle one by the exact kernel on the lists the FAST launch left."""
    from vgsim_amd.ensemble import Ensemble
    sim = _c3(2020)
    m = sim.simulation
    rng = np.random.default_rng(5)
    occ = [700, 3, 64, 65, 130, 17, 16, 1000] + [int(v) for v in rng.integers(1, 400, size=56)]
    for pn in range(64):
        haps = rng.choice(m.hapNum, size=occ[pn], replace=False)
        if pn == 4:
            haps[0] = m.hapNum - 1
        m.infectious[pn, haps] = rng.integers(1, 3, size=occ[pn])
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
    m.set_mutation_rate(0.2, None, None)
    R = 7
    seeds = 900 + np.arange(R, dtype=np.int64)
    plan = [(3000, "fast"), (1500, "exact"), (3000, "fast")]
    ens = Ensemble(sim, R, seeds=seeds)
    got = []
    for n_ev, mode in plan:
        res = ens.simulate(n_ev, sample_size=10 ** 9, record_events=True, kernel="quad", mode=mode)
        got.append((res.events.copy(), [ens.replicate_events(r) for r in range(R)], [ens.replicate_state(r) for r in range(R)]))
    ens.close()
    ens = Ensemble(sim, R, seeds=seeds)
    for (n_ev, _), (events, chains, states) in zip(plan, got):
        res = ens.simulate(n_ev, sample_size=10 ** 9, record_events=True, kernel="quad", mode="exact")
        assert np.array_equal(res.events, events)
        for r in range(R):
            want = ens.replicate_events(r)
            assert np.array_equal(chains[r][1:], want[1:]), "replicate %d: %s" % (r, helpers.describe_first_diff(chains[r][1:], want[1:], want.shape[1]))
            np.testing.assert_allclose(chains[r][0], want[0], rtol=RTOL_TIME, atol=0.0)
            sw = ens.replicate_state(r)
            assert np.array_equal(states[r].infectious, sw.infectious) and np.array_equal(states[r].susceptible, sw.susceptible)
    ens.close()


def test_fast_row_kernel_one_haplotype_several_populations(oracle_mod):
    """Lists of ONE slot (no sites) filled by migrants: insertions into lists shorter than a lane's four entries."""
    from vgsim_amd import Simulator

    def build(seed):
        with helpers.quiet():
            s = Simulator(number_of_sites=0, populations_number=3, number_of_susceptible_groups=1, seed=seed)
        s.set_transmission_rate(3.0); s.set_recovery_rate(1.0); s.set_sampling_rate(0.2)
        s.set_population_size(5000); s.set_total_migration_probability(0.2)
        return s
    for seed in (5, 6, 7):
        hip = build(seed)
        with helpers.quiet():
            hip.simulate(6000, sample_size=10 ** 9, kernel="quad", mode="fast")
        ref = build(seed).simulation
        assert oracle_mod.run_direct(ref, 6000, 10 ** 9, -1, 200) == 0
        _assert_tier_b(hip.simulation, ref, "one haplotype, seed %d" % seed)
        assert ref.migPlus > 0


def test_fast_row_kernel_squeezes_a_full_list(oracle_mod, monkeypatch):
    """List capacity forced down to 192 entries (VGX_LIST_CAP): with a high mutation rate the lists fill up with zero-count
    entries, the kernel squeezes them out in place when an insertion finds the list full, and the run equals the exact
    kernel's with the default capacity."""
    from vgsim_amd.ensemble import Ensemble

    def run(mode, cap):
        if cap:
            monkeypatch.setenv("VGX_LIST_CAP", str(cap))
        else:
            monkeypatch.delenv("VGX_LIST_CAP", raising=False)
        sim = _c3(2020)
        m = sim.simulation
        rng = np.random.default_rng(11)
        for pn in range(64):
            haps = rng.choice(m.hapNum, size=100, replace=False)
            m.infectious[pn, haps] = 1
            m.susceptible[pn, 0] -= 100
        sim.set_transmission_rate(1.0)        # balanced: about a hundred occupied haplotypes per population throughout
        m.set_mutation_rate(0.6, None, None)
        R = 5
        ens = Ensemble(sim, R, seeds=300 + np.arange(R, dtype=np.int64))
        # 1250 events per population, two thirds of them mutations: every one leaves a zero-count entry, and an insertion behind the
        # list's last zero-count entry appends, so the lists reach the 192 slots several times over
        res = ens.simulate(80000, sample_size=10 ** 9, record_events=True, kernel="quad", mode=mode)
        out = (res.events.copy(), [ens.replicate_events(r) for r in range(R)], [ens.replicate_state(r) for r in range(R)])
        ens.close()
        return out
    ev_f, ch_f, st_f = run("fast", 192)
    ev_e, ch_e, st_e = run("exact", 0)
    assert np.array_equal(ev_f, ev_e)
    for r in range(len(ch_f)):
        assert np.array_equal(ch_f[r][1:], ch_e[r][1:]), "replicate %d" % r
        assert np.array_equal(st_f[r].infectious, st_e[r].infectious)
        assert (st_f[r].infectious != 0).sum(axis=1).max() <= 192


@pytest.mark.parametrize("seed", [385, 3, 17, 55, 120, 233])
def test_row_kernels_equal_the_wave_kernel_on_random_start_states(seed):
    """tools/stress_rowlists.py: random start states with lists around the tile boundaries, random mutation and migration rates, two
    launches each — the exact row kernels (bit for bit) and the FAST row kernel (integer rows, compartments) against the
    one-replicate-per-wavefront kernel.  Seed 385 is the state on which a squeezed list once kept stale tile sums behind its new end."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("stress_rowlists", os.path.join(os.path.dirname(__file__), "..", "tools", "stress_rowlists.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    good, text = mod.check(seed)
    assert good, text
