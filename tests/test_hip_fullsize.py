"""Parity at BASELINE.json's full config-3 size (65 536 haplotypes x 64 populations).  The dense oracle would
need ~6 ms per event there, so the check uses (a) the oracle's occupied-only mode, which tests/test_oracle_golden.py
proves bit-identical to the dense reference order, on one replicate, and (b) size-independent invariants on a
whole ensemble: host conservation per population, counters vs compartments, event-type bookkeeping."""
import contextlib
import hashlib
import io

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def c3(seed, mut=0.01):
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=8, populations_number=64, number_of_susceptible_groups=1, seed=seed)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1)
    s.set_mutation_rate(mut); s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    return s


@pytest.mark.parametrize("seed,mut,n", [(2020, 0.01, 40000), (2021, 0.4, 12000)])
def test_config3_bit_exact_vs_sparse_oracle(oracle_mod, seed, mut, n):
    hip = c3(seed, mut)
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9)
    ref = c3(seed, mut).simulation
    assert oracle_mod.run_direct(ref, n, 10 ** 9, -1, 200, sparse=True) == 0
    helpers.assert_models_equal(hip.simulation, ref, "config3 seed %d" % seed)
    assert (hip.simulation.infectious != 0).sum() > (50 if mut < 0.1 else 1000)   # the lists are really exercised


def test_config3_ensemble_invariants():
    from vgsim_amd.ensemble import Ensemble
    sim = c3(7)
    R, N = 256, 20000
    ens = Ensemble(sim, R)
    res = ens.simulate(N, sample_size=10 ** 12, record_events=True)
    assert (res.events == N).all() or (res.events <= N).all()
    sizes = sim.simulation.sizes
    for r in (0, 17, 255):
        st = ens.replicate_state(r)
        assert np.array_equal(st.susceptible.sum(axis=1) + st.infectious.sum(axis=1), sizes)      # hosts conserved per deme
        assert st.globalInfectious == st.infectious.sum() and np.array_equal(st.totalInfectious, st.infectious.sum(axis=1))
        chain = ens.replicate_events(r)
        hist = np.bincount(chain[1].astype(int), minlength=7)
        assert (hist[0], hist[1], hist[2], hist[3], hist[5]) == (st.bCounter, st.dCounter, st.sCounter, st.mCounter, st.migPlus)
        assert 1 + st.bCounter + st.migPlus - st.dCounter - st.sCounter == st.globalInfectious    # index case + births - removals
        assert (np.diff(chain[0]) >= 0).all() and chain[0, -1] == st.currentTime
        # every loop iteration either records an event or is a rejected migration; failed attempts add theirs
        assert res.loop_iterations[r] >= res.events[r] + st.migNonPlus
        if res.restarts[r] == 0:
            assert res.loop_iterations[r] == res.events[r] + st.migNonPlus
    ens.close()


def test_config3_spread_occupancy_exact_oracle_and_fast():
    """4096 occupied haplotypes per population (lists of 64 tiles: tile sums, two-level lower bound, row-wise running
    sums, chunked tree scans): EXACT is bit-identical to the oracle's occupied-only mode, FAST has the same integer
    columns and compartments."""
    from vgsim_amd import _capi
    from oracle import oracle as oracle_mod
    oracle_mod.build()

    def spread(seed):
        s = c3(seed)
        m = s.simulation
        rng = np.random.default_rng(99)
        for pn in range(64):
            haps = rng.choice(m.hapNum, size=4096, replace=False)
            m.infectious[pn, haps] = rng.integers(1, 4, size=4096)
            m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
        return s

    n = 3000
    ex = spread(11)
    with helpers.quiet():
        ex.simulate(n, sample_size=10 ** 9)
    ref = spread(11).simulation
    assert oracle_mod.run_direct(ref, n, 10 ** 9, -1, 200, sparse=True) == 0
    helpers.assert_models_equal(ex.simulation, ref, "config3 spread")
    fa = spread(11)
    with helpers.quiet():
        fa.simulate(n, sample_size=10 ** 9, mode="fast")
    a, b = helpers.chain_of(fa.simulation), helpers.chain_of(ex.simulation)
    assert np.array_equal(a[1:, :n], b[1:, :n])
    np.testing.assert_allclose(a[0, :n], b[0, :n], rtol=1e-9, atol=0.0)
    assert np.array_equal(fa.simulation.infectious, ex.simulation.infectious)


def test_config4_tau_invariants_at_full_size():
    """BASELINE config 4 at its full size (2^20 haplotypes x 256 populations, migration, dense occupancy): the reference
    cannot construct this shape, so the check is on size-independent properties of three leaps — hosts conserved per
    population, no negative compartment, totals equal to the compartments' sums, counters equal to the net change — and on the
    two ways a try keeps its deltas (list of moves / dense arrays) giving the same 2^28 compartments bit for bit."""
    import ctypes as C
    from vgsim_amd import Simulator, _capi
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    m = s.simulation
    m.infectious[:] = 3
    m.susceptible[:, 0] -= 3 * m.hapNum
    before_I = int(m.infectious.sum()) + 1       # + the index case the first call adds (pyx:435-448)
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
    m.events.CreateEvents(3); m.events.ptr = 1; m.events.CreateEvents(3)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([2020], dtype=np.int64))
    o = _capi.VgxRunOpts(); o.record_events = 0
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, 3, 10 ** 15, -1.0, 1, C.byref(o)))
    eng.get_state(m, 0)
    c = eng.counters(0)
    eng.close()
    # ... and the same three leaps with the dense delta arrays and the fused per-compartment tests (reserved[1] = 2): the same
    # draws and decisions, so the 2^28 compartments, the counters and the clock must come out bit for bit the same
    keep = (hashlib.sha256(m.infectious.tobytes()).hexdigest(), m.susceptible.copy(), m.currentTime, int(c.reserved[0]),
            [int(getattr(m, k)) for k in m.COUNTERS])
    with contextlib.redirect_stdout(io.StringIO()):
        s2 = Simulator(number_of_sites=10, populations_number=256, seed=2020)
    s2.set_transmission_rate(2.5); s2.set_recovery_rate(0.9); s2.set_sampling_rate(0.1); s2.set_mutation_rate(0.01)
    s2.set_total_migration_probability(0.01); s2.set_population_size(10 ** 7)
    m2 = s2.simulation
    m2.infectious[:] = 3
    m2.susceptible[:, 0] -= 3 * m2.hapNum
    eng2 = _capi.HipEngine(m2.sites, m2.hapNum, m2.popNum, m2.susNum, n_replicates=1)
    m2.events.CreateEvents(3); m2.events.ptr = 1; m2.events.CreateEvents(3)
    eng2.set_params(m2); eng2.set_state(m2); eng2.set_seeds(np.array([2020], dtype=np.int64))
    o2 = _capi.VgxRunOpts(); o2.record_events = 0; o2.reserved[1] = 2
    eng2._check(eng2.lib.vgx_simulate_tau(eng2.handle, 3, 10 ** 15, -1.0, 1, C.byref(o2)))
    eng2.get_state(m2, 0)
    c2 = eng2.counters(0)
    eng2.close()
    assert keep == (hashlib.sha256(m2.infectious.tobytes()).hexdigest(), keep[1], m2.currentTime, int(c2.reserved[0]),
                    [int(getattr(m2, k)) for k in m2.COUNTERS]) and np.array_equal(keep[1], m2.susceptible)
    del m2, s2
    assert c.loop_iterations == 3 and c.reserved[0] > 5 * 10 ** 6                      # millions of events per leap
    assert c.reserved[0] == m.bCounter + m.dCounter + m.sCounter + m.mCounter + m.migPlus
    assert m.infectious.min() >= 0 and m.susceptible.min() >= 0
    assert np.array_equal(m.totalInfectious, m.infectious.sum(axis=1)) and m.globalInfectious == int(m.infectious.sum())
    assert np.array_equal(m.susceptible.sum(axis=1) + m.infectious.sum(axis=1), m.sizes)     # per population: migrants infect in place
    assert int(m.infectious.sum()) == before_I + m.bCounter + m.migPlus - m.dCounter - m.sCounter
    assert m.currentTime > 0
