"""The single-trajectory latency kernel (vgx_solo.hip: one replicate per wavefront, the whole dense model in LDS and registers;
hapNum <= 64, popNum <= 128) against the CPU oracle and the reference's goldens, bit for bit: every direct case of the suite it
takes with the kernel forced (``kernel='solo'``), single runs through the automatic choice (what ``Simulator.simulate()`` now
runs on), small ensembles whose replicates must equal single seeded runs, the model of the reference's published benchmark
(data/Table 3: 2, 10 and 100 demes), and the reciprocal division of BirthRate's terms against the division."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu


def _takes(name):
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    return (ctor.get("number_of_sites", 0) <= 3 and ctor.get("populations_number", 1) <= 128
            and all(kw.get("method", "direct") == "direct" for _, kw in phases))


SOLO = [n for n in models.CASES if _takes(n)]
GOLDEN = [n for n in ("g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "example", "p70", "stress_h64", "continuation", "cmd_example",
                      "sample_stop", "time_stop", "extinct", "extinct_restart") if n in SOLO]


@pytest.mark.parametrize("name", SOLO + [n for n in models.ORACLE_ONLY_CASES if _takes(n)])
def test_solo_kernel_bit_exact_vs_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name, kernel="solo").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", [n for n in SOLO if not n.startswith("g") or n.endswith("_short")])
def test_solo_general_layout_bit_exact_vs_oracle(oracle_mod, name, monkeypatch):
    """The general BirthRate layout (one pass per segment) on models the compact layout would take."""
    monkeypatch.setenv("VGX_SOLO_GENERAL", "1")
    hip = helpers.run_case_hip(name, kernel="solo").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", GOLDEN)
def test_solo_kernel_matches_the_goldens(name):
    hip = helpers.run_case_hip(name, kernel="solo").simulation
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), rtol_time=1e-12, leftovers=False)


def _program_model(P, S, migration, seed, per_hap):
    """sites = 2 (16 haplotypes), P > 16 populations: BirthRate runs as the program of chain segments over the population lanes."""
    from vgsim_amd import Simulator
    with helpers.quiet():
        s = Simulator(number_of_sites=2, populations_number=P, number_of_susceptible_groups=S, seed=seed)
    s.set_transmission_rate(2.6); s.set_recovery_rate(0.8); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
    s.set_transmission_rate(3.4, haplotype="GG")
    s.set_population_size(40000)
    for g in range(1, S):
        s.set_susceptibility(0.25 * g, susceptibility_type=g)
        s.set_immunity_transition(0.01, source=g, target=0)
    if S > 1:
        s.set_susceptibility_type(1)
    for h, g, v in per_hap:
        s.set_susceptibility(v, susceptibility_type=g, haplotype=h)
    if migration:
        s.set_total_migration_probability(migration)
    return s


@pytest.mark.parametrize("P,S,migration,per_hap", [
    (20, 2, 0.0, []),                                   # two segments in a row, no migration: every pass a chain alone
    (20, 2, 0.05, []),                                  # the migration rates' sum beside the first segment, the second alone
    (33, 3, 0.02, [("C*", 1, 0.6), ("G*", 2, 0.9)]),    # a tree of segments: siblings share a pass
    (70, 3, 0.03, [("A*", 1, 0.0), ("T*", 2, 0.7)]),    # two registers of population lanes; a class without its group-1 segment
    (100, 1, 0.01, []),                                 # one segment
    (17, 4, 0.0, [("AC", 3, 1.0), ("GT", 2, 0.0)])])
def test_birthrate_program_shapes_bit_exact_vs_oracle(oracle_mod, P, S, migration, per_hap):
    """vgx_solo's BirthRate beyond 16 populations: the host's schedule of passes (two independent chains per pass, a chain alone where it
    has no partner, the migration rates' sum riding along) on segment trees of several shapes, one trajectory against the oracle."""
    hip, ref = _program_model(P, S, migration, 77 + P, per_hap), _program_model(P, S, migration, 77 + P, per_hap)
    assert oracle_mod.run_direct(ref.simulation, 6000, 10 ** 9, -1, 200) == 0
    with helpers.quiet():
        hip.simulate(6000, sample_size=10 ** 9, kernel="solo")
    assert hip.simulation._engine.last_kernel == "solo"
    helpers.assert_models_equal(hip.simulation, ref.simulation, "program P=%d S=%d" % (P, S))


def test_single_runs_take_the_solo_kernel_automatically():
    sim = helpers.run_case_hip("g9_short")
    assert sim.simulation._engine.last_kernel == "solo"


def test_larger_shapes_are_refused():
    from vgsim_amd._capi import VgxError
    with pytest.raises(VgxError), helpers.quiet():
        helpers.run_case_hip("stress_h256", kernel="solo")      # 256 haplotypes


@pytest.mark.parametrize("name", models.RECOMBINATION_CASES)
def test_recombinant_births_on_the_solo_kernel(oracle_mod, name):
    """The recombination branch of Birth (pyx:575-596) on the single-trajectory kernel — what a Simulator with a
    recombination probability now runs on: second parent, breakpoint, records (kept across Restarts like upstream's), log and state
    equal the oracle's, which is pinned on fixtures recorded from the reference; also against those fixtures directly."""
    hip = helpers.run_case_hip(name).simulation
    assert hip._engine.last_kernel == "solo"
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)
    assert len(hip.rec.his) > 0
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), rtol_time=1e-12, leftovers=False)


def _single(oracle_mod, name, seed, n_events):
    from vgsim_amd import Simulator
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    with helpers.quiet():
        one = Simulator(**dict(ctor, seed=int(seed)))
    phases[0][0](one)
    m = one.simulation
    assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200) == 0
    return m


@pytest.mark.parametrize("name,n_events", [("g9_short", 4000), ("g4_short", 3000), ("p70", 2500), ("lockdown_restart", 1500),
                                           ("example", 3000), ("stress_h64", 3000)])
def test_solo_replicates_equal_single_runs(oracle_mod, name, n_events):
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 5
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    seeds = np.array([3, 2021, 99, 12345678901, 1], dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, kernel="solo")
    for r in range(R):
        m = _single(oracle_mod, name, seeds[r], n_events)
        assert res.events[r] == m.events.ptr, "replicate %d" % r
        chain = ens.replicate_events(r)
        assert np.array_equal(chain, m.events.as_array()[:, :m.events.ptr]), "replicate %d: %s" % (
            r, helpers.describe_first_diff(chain, m.events.as_array(), m.events.ptr))
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible)
        assert np.array_equal(st.lockdownON, m.lockdownON) and np.array_equal(st.contactDensity, m.contactDensity)
        assert st.currentTime == m.currentTime and st.good_attempt == m.good_attempt
        for k in st.COUNTERS:
            assert getattr(st, k) == getattr(m, k), k
    ens.close()


@pytest.mark.parametrize("name,n_events", [("g1", 800), ("g5", 800), ("g6_short", 800)])
def test_one_class_ensembles_below_8192_replicates_take_the_latency_kernel(oracle_mod, name, n_events):
    """The automatic choice (vgx_api.hip, direct dispatch): a one-class model both this kernel and the one-class row kernel take
    runs here below 8192 replicates; a few replicates out of 4096 against single oracle runs, and the row kernel from 8192 on."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 4096
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    seeds = np.arange(R, dtype=np.int64) * 7919 + 11
    ens = Ensemble(sim, R, seeds=seeds)
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True)
    assert ens.engine.last_kernel == "solo"
    for r in (0, 1, 63, 64, 2047, 4095):
        m = _single(oracle_mod, name, seeds[r], n_events)
        assert res.events[r] == m.events.ptr, "replicate %d" % r
        chain = ens.replicate_events(r)
        assert np.array_equal(chain, m.events.as_array()[:, :m.events.ptr]), "replicate %d: %s" % (
            r, helpers.describe_first_diff(chain, m.events.as_array(), m.events.ptr))
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible)
        assert st.currentTime == m.currentTime
    ens.close()
    ens = Ensemble(sim, 8192, seeds=np.arange(8192, dtype=np.int64) + 5)
    ens.simulate(50, sample_size=10 ** 9)
    assert ens.engine.last_kernel == "quad"
    ens.close()


def _table3(K, M, seed):
    import bench
    return bench.make_table3(K, M, seed)


@pytest.mark.parametrize("K,M,n", [(2, 0.001, 60000), (2, 0.1, 30000), (10, 0.001, 30000), (10, 0.1, 20000), (100, 0.001, 6000), (100, 0.1, 6000)])
def test_table3_model_single_trajectory(oracle_mod, K, M, n):
    """The reference's published benchmark model (data/Table 3), one trajectory: solo kernel == oracle, both division forms."""
    import os
    ref = _table3(K, M, 2023).simulation
    assert oracle_mod.run_direct(ref, n, 10 ** 12, -1, 200) == 0
    for plain in (False, True):
        if plain:
            os.environ["VGX_SOLO_PLAIN_DIV"] = "1"
        try:
            sim = _table3(K, M, 2023)
            with helpers.quiet():
                sim.simulate(n, sample_size=10 ** 12, kernel="solo")
        finally:
            os.environ.pop("VGX_SOLO_PLAIN_DIV", None)
        helpers.assert_models_equal(sim.simulation, ref, "table3 K=%d M=%g plain_div=%s" % (K, M, plain))


def test_table3_small_demes_switch_lockdowns(oracle_mod):
    """Demes small enough for the NPI to switch on and off (UpdateAllRates inside the loop, lockdown log, host clock keys)."""
    import bench
    from vgsim_amd import Simulator
    def build():
        s = bench.make_table3(4, 0.05, 11)
        s.set_population_size(3000)
        s.set_npi([0.2, 0.02, 0.005])
        return s
    ref = build().simulation
    assert oracle_mod.run_direct(ref, 40000, 10 ** 12, -1, 200) == 0
    sim = build()
    with helpers.quiet():
        sim.simulate(40000, sample_size=10 ** 12, kernel="solo")
    assert ref.swapLockdown > 0
    helpers.assert_models_equal(sim.simulation, ref, "table3 small demes")


def test_solo_without_event_log_and_with_trajectories(oracle_mod):
    """record_events = 0 (device clock) and trajectory bins: the CLOCK instantiation."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    ctor, phases = models.CASES["g9_short"]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    seeds = np.array([7, 8, 9], dtype=np.int64)
    a = Ensemble(sim, 3, seeds=seeds)
    ra = a.simulate(4000, sample_size=10 ** 9, record_events=False, traj_points=33, traj_window=(0.0, 8.0), kernel="solo")
    b = Ensemble(sim, 3, seeds=seeds)
    rb = b.simulate(4000, sample_size=10 ** 9, record_events=False, traj_points=33, traj_window=(0.0, 8.0), kernel="wave")
    assert np.array_equal(ra.events, rb.events)
    assert np.array_equal(a.trajectories(), b.trajectories())
    for r in range(3):
        sa, sb = a.replicate_state(r), b.replicate_state(r)
        assert np.array_equal(sa.infectious, sb.infectious) and np.array_equal(sa.susceptible, sb.susceptible)
        assert sa.currentTime == sb.currentTime
    a.close(); b.close()


def test_reciprocal_division_equals_the_division():
    """x / actualSizes through the correctly rounded reciprocal and two residual corrections (vgx_solo.hip div_by_const), and the
    division sequence without range scaling (fdiv), are the IEEE quotient: random operands over the ranges the kernel's operands
    take, plus adversarial significands."""
    from vgsim_amd import _capi
    lib = _capi.load_library()
    rng = np.random.default_rng(12345)
    n = 1 << 22
    num = np.concatenate([rng.random(n) * 10.0 ** rng.integers(-12, 12, n), rng.integers(0, 1 << 52, n).astype(np.float64),
                          np.ldexp(1.0 + rng.integers(0, 64, n) * 2.0 ** -52, rng.integers(-40, 40, n)), np.zeros(16)])
    den = np.concatenate([rng.random(n) * 10.0 ** rng.integers(0, 12, n) + 1e-3, rng.integers(1, 1 << 40, n).astype(np.float64),
                          np.ldexp(2.0 - rng.integers(1, 64, n) * 2.0 ** -52, rng.integers(-20, 40, n)), rng.random(16) + 0.5])
    num = np.ascontiguousarray(num); den = np.ascontiguousarray(den)
    q1 = np.empty_like(num); q2 = np.empty_like(num); q3 = np.empty_like(num)
    import ctypes as C
    F = C.POINTER(C.c_double)
    rc = lib.vgx_test_div_by_const(num.ctypes.data_as(F), den.ctypes.data_as(F), len(num), q1.ctypes.data_as(F), q3.ctypes.data_as(F),
                                   q2.ctypes.data_as(F))
    assert rc == 0
    assert np.array_equal(q2, num / den)          # the device's division is the IEEE quotient
    for name, q in (("reciprocal sequence", q1), ("lean division", q3)):
        bad = np.nonzero(q != q2)[0]
        assert len(bad) == 0, "%s, first mismatch: %r / %r -> %r vs %r" % (name, num[bad[0]], den[bad[0]], q[bad[0]], q2[bad[0]])
