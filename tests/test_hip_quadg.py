"""The GENERAL four-replicates-per-wavefront kernel (vgx_quadg.hip: several susceptibility groups, several rate classes,
lockdown switches, up to 128 populations) against the CPU oracle and the reference's goldens, bit for bit: every direct case of
the suite with the kernel forced (``kernel='quadg'``; ``'quad'`` takes it wherever the one-class form refuses; models with a
recombination probability run the instantiations that carry the recombination branch of Birth), ensembles whose replicates must equal single seeded runs (the four rows of a wavefront take different branches:
immunity transitions, births, migrations, lockdown switches, restarts), and the model of the reference's published benchmark
(data/Table 3)."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu

DIRECT = [n for n, (_, ph) in models.CASES.items() if all(kw.get("method", "direct") == "direct" for _, kw in ph)]
QUADG = [n for n in DIRECT if n not in models.RECOMBINATION_CASES]      # (those: test_recombinant_births_on_the_general_row_kernel)
# the models the one-class form refuses (what `kernel='quad'` now hands to the general form)
GENERAL_ONLY = ["g2", "g3", "g4", "g7", "g9", "example", "p70", "stress_h64", "stress_h256", "continuation", "cmd_example"]


@pytest.mark.parametrize("name", QUADG + list(models.ORACLE_ONLY_CASES))
def test_general_row_kernel_bit_exact_vs_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name, kernel="quadg").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", GENERAL_ONLY)
def test_kernel_quad_takes_the_general_form_and_matches_the_goldens(name):
    hip = helpers.run_case_hip(name, kernel="quad").simulation
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), rtol_time=1e-12, leftovers=False)


@pytest.mark.parametrize("name", models.RECOMBINATION_CASES)
def test_recombinant_births_on_the_general_row_kernel(oracle_mod, name):
    """The recombination branch of Birth (pyx:575-596) on the row-per-replicate layout (`vgx_quadg_kernel_*_rec`): the second parent
    by fastChoose over birthInf in list order, breakpoint, the newborn's haplotype as a list operation, the forward records (kept
    across Restarts like upstream's) — log, state and records equal the oracle's and the fixtures recorded from the reference."""
    hip = helpers.run_case_hip(name, kernel="quadg").simulation
    assert hip._engine.last_kernel == "quadg"
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)
    assert len(hip.rec.his) > 0
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), rtol_time=1e-12, leftovers=False)


@pytest.mark.parametrize("name,n_events", [("recomb_a", 3000), ("recomb_restart", 2500)])
def test_recombination_in_an_ensemble_on_the_general_row_kernel(oracle_mod, name, n_events):
    """Four rows of a wavefront with recombinant births at different events: every replicate equals its single seeded oracle run
    (a recombinant birth's row names the second parent; the newborn's haplotype shows in the compartments)."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 6
    ctor, phases = models.CASES[name]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    seeds = np.array([3, 4, 2021, 99, 12345678901, 1], dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, kernel="quadg")
    assert ens.engine.last_kernel == "quadg"
    for r in range(R):
        m = _single(oracle_mod, name, seeds[r], n_events)
        assert res.events[r] == m.events.ptr, "replicate %d" % r
        chain = ens.replicate_events(r)
        assert np.array_equal(chain, m.events.as_array()[:, :m.events.ptr]), "replicate %d: %s" % (
            r, helpers.describe_first_diff(chain, m.events.as_array(), m.events.ptr))
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible)
        for k in st.COUNTERS:
            assert getattr(st, k) == getattr(m, k), k
    ens.close()


def _single(oracle_mod, name, seed, n_events):
    from vgsim_amd import Simulator
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    with helpers.quiet():
        one = Simulator(**dict(ctor, seed=int(seed)))
    phases[0][0](one)
    m = one.simulation
    assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200) == 0
    return m


@pytest.mark.parametrize("name,n_events", [("g9_short", 4000), ("g4_short", 3000), ("stress_h256", 2500), ("p70", 2500),
                                           ("lockdown_restart", 1500), ("example", 3000)])
def test_general_row_kernel_replicates_equal_single_runs(oracle_mod, name, n_events):
    """Rows of one wavefront run different trajectories (also: fewer replicates than rows in the last wavefront)."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 7
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    seeds = np.array([3, 4, 5, 2021, 99, 12345678901, 1], dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, kernel="quadg")
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


@pytest.mark.parametrize("name,n_events", [("g9_short", 4000), ("stress_h256", 2500), ("p70", 2000), ("lockdown_restart", 1500)])
def test_counter_based_stream_on_the_general_row_kernel(oracle_mod, name, n_events):
    """``mode='fast_philox'`` on a general model: the general row kernel's (or, for small models, the latency kernel's) exact arithmetic on the Philox stream (iteration i of an attempt
    takes outputs 2 i and 2 i + 1 of the stream of (seed, attempt)) — every replicate equals the ORACLE fed with the same stream, integer
    rows, counters and compartments bit for bit, times to 1e-9 (the device clock).  2048 replicates: the automatic choice."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    R = 2048
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    seeds = 5000 + np.arange(R, dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, mode="fast_philox")
    assert ens.engine.last_kernel in ("quadg", "solo")      # (small models in the compact layout: two wavefronts per SIMD of the latency kernel)
    for r in (0, 1, 777, R - 1):
        with helpers.quiet():
            one = Simulator(**dict(ctor, seed=int(seeds[r])))
        phases[0][0](one)
        m = one.simulation
        assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200, log_mode=oracle_mod.RNG_PHILOX) == 0
        assert res.events[r] == m.events.ptr, "replicate %d" % r
        chain, want = ens.replicate_events(r), m.events.as_array()[:, :m.events.ptr]
        assert np.array_equal(chain[1:], want[1:]), "replicate %d: %s" % (r, helpers.describe_first_diff(chain[1:], want[1:], m.events.ptr))
        np.testing.assert_allclose(chain[0], want[0], rtol=1e-9, atol=0.0)
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible)
        assert np.array_equal(st.lockdownON, m.lockdownON) and st.good_attempt == m.good_attempt
        for k in st.COUNTERS:
            assert getattr(st, k) == getattr(m, k), k
    ens.close()


def _table3(K, M, seed):
    """data/Table 3/Table 3.py:5-22 through today's setters (bench.py::make_table3)."""
    import bench
    return bench.make_table3(K, M, seed=seed)


@pytest.mark.parametrize("K,M,n,size", [(2, 0.1, 6000, None), (10, 0.001, 6000, None), (10, 0.1, 4000, 3000), (100, 0.1, 1500, 400)])
def test_table3_model_bit_exact_vs_oracle(oracle_mod, K, M, n, size):
    """The reference's published benchmark model (16 haplotypes, 3 susceptibility groups, 4 rate classes, NPI on every deme):
    single runs vs the oracle, also with demes small enough for the NPI to switch on and off during the run."""
    hip = _table3(K, M, 2023)
    ref = _table3(K, M, 2023)
    if size is not None:
        for s in (hip, ref):
            s.set_population_size(size)
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9, kernel="quadg")
    assert oracle_mod.run_direct(ref.simulation, n, 10 ** 9, -1, 200) == 0
    helpers.assert_models_equal(hip.simulation, ref.simulation, "table3 K=%d M=%g" % (K, M))
    if size is not None:
        assert hip.simulation.swapLockdown > 0
