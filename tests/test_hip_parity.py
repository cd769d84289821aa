"""Parity tests proper (GPU): the HIP engine, called through the C ABI exactly as ``Simulator.simulate``
does, against (a) the CPU oracle on the same seeded inputs — bit for bit on all six log rows, counters,
lockdown log and final compartments — and (b) the golden vectors recorded from the reference itself: all six
rows bit for bit (sha256 of the full (6, N) chain for the 100 000-event goldens g1..g9, the reference's own
``testing/check_simulator.py`` criterion) wherever this host's libm reproduces the fixture host's ``log``
(tests/golden/libm_probe.json); elsewhere the integer rows exactly and the time row within 1e-12 relative.
Event times come from the engine's host clock (vgx_api.hip: host_clock): accumulated with the host's libm from the
logged denominators, exactly as the reference's SampleTime does (pyx:476-478)."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu

DIRECT = [n for n, (_, ph) in models.CASES.items() if all(kw.get("method", "direct") == "direct" for _, kw in ph)]


LANE_OK = [n for n in DIRECT if not n.startswith(("stress_h256", "c3_", "p70"))]   # popNum <= 16, popNum * hapNum <= 1024


@pytest.mark.parametrize("name", DIRECT + list(models.ORACLE_ONLY_CASES))
def test_direct_bit_exact_vs_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name).simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", DIRECT + list(models.ORACLE_ONLY_CASES))
def test_wave_kernel_bit_exact_vs_oracle(oracle_mod, name):
    """The one-replicate-per-wavefront kernel (vgx_direct.hip) forced for every case: the automatic choice hands single runs of
    small general models to the row kernel (vgx_quadg.hip) since round 3."""
    hip = helpers.run_case_hip(name, kernel="wave").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


def test_which_time_row_branch_runs_on_this_host():
    """Records (in pytest's warnings summary, which ends the run's output) which branch the golden tests below take here:
    libm_probe_match=True -> sha256 of the full (6, N) chain incl. the time row; False -> integer rows exact, times 1e-12."""
    import warnings
    warnings.warn("libm_probe_match=%s (time-row branch of the golden tests: %s)" % (
        helpers.libm_matches_fixture_host(), "bit-exact sha256" if helpers.libm_matches_fixture_host() else "rtol 1e-12"))


@pytest.mark.parametrize("name", DIRECT)
def test_direct_vs_reference_goldens(name):
    print("libm_probe_match=%s" % helpers.libm_matches_fixture_host())
    hip = helpers.run_case_hip(name).simulation
    helpers.check_against_golden(hip, name, exact_time=helpers.libm_matches_fixture_host(), rtol_time=1e-12, leftovers=False)


@pytest.mark.parametrize("name", LANE_OK + list(models.ORACLE_ONLY_CASES))
def test_lane_kernel_bit_exact_vs_oracle(oracle_mod, name):
    """The one-replicate-per-lane kernel (vgx_lanes.hip: the reference's serial loops on dense per-replicate state),
    forced for every small case of the suite."""
    hip = helpers.run_case_hip(name, kernel="lane").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", models.RECOMBINATION_CASES)
def test_recombination_on_the_wave_kernel_bit_exact_vs_oracle(oracle_mod, name):
    """Recombinant births (pyx:575-596) on the occupancy-list kernel (what large haplotype spaces use): the second parent is
    chosen over the list in haplotype order with the reference's weights; log, records, counters and state equal the
    oracle's, which is pinned on fixtures recorded from the reference."""
    hip = helpers.run_case_hip(name, kernel="wave").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    helpers.assert_models_equal(hip, ref, name)
    assert len(hip.rec.his) > 0


def test_recombination_is_refused_in_fast_mode():
    from vgsim_amd._capi import VgxError
    with helpers.quiet():
        sim, phases = models.build(__import__("vgsim_amd").Simulator, "recomb_a")
        phases[0][0](sim)
    for kw in (dict(mode="fast"), dict(mode="fast_philox")):
        with pytest.raises(VgxError), helpers.quiet():
            sim.simulate(100, **kw)
    assert len(helpers.run_case_hip("recomb_a").simulation.rec.his) > 1000


@pytest.mark.parametrize("sites,seed,mut,n", [(1, 2021, 0.3, 6000), (3, 7, 0.3, 6000), (4, 11, 0.3, 8000), (5, 4, 0.05, 8000)])
def test_memory_optimization_vs_oracle_table_code(oracle_mod, sites, seed, mut, n):
    """``memory_optimization=True`` (one population): the engine against the ORACLE'S op-for-op restatement of the reference's
    table code (AddMemory pyx:264-274, AddHaplotype pyx:355-377, the lookup of Mutation pyx:651-660): event chain, compartments
    and the table itself (numToHap, hapToNum, currentHapNum, maxHapNum grown in addMemoryNum steps)."""
    from vgsim_amd import Simulator

    def make():
        with helpers.quiet():
            s = Simulator(number_of_sites=sites, populations_number=1, number_of_susceptible_groups=2, seed=seed, memory_optimization=True)
        s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(mut)
        s.set_transmission_rate(3.5, haplotype=2)
        s.set_susceptibility_type(1); s.set_susceptibility(0.4, susceptibility_type=1)
        s.set_immunity_transition(0.05, source=1, target=0)
        return s
    hip, ref = make(), make()
    with helpers.quiet():
        hip.simulate(n, sample_size=10 ** 9)
    assert oracle_mod.run_direct_memopt(ref.simulation, n, 10 ** 9, -1, 200) == 0
    h, r = hip.simulation, ref.simulation
    assert r.good_attempt == 1, "pick a seed whose first attempt survives (see _refresh_haplotype_table)"
    helpers.assert_models_equal(h, r, "memory_optimization sites=%d" % sites)
    tb = r._memopt
    cur = tb.currentHapNum
    assert h.currentHapNum == cur and h.maxHapNum == tb.maxHapNum and len(h.numToHap) == tb.maxHapNum
    assert np.array_equal(h.numToHap[:cur], tb.numToHap[:cur])
    assert np.array_equal(h.hapToNum[tb.numToHap[:cur]], tb.hapToNum[tb.numToHap[:cur]])
    if sites >= 5:
        assert 4 ** (sites - 2) < tb.maxHapNum < 4 ** sites      # the table grew through AddMemory, not to the full space


def test_memory_optimization_where_upstream_corrupts_its_state(monkeypatch):
    """Several populations / tau with memory_optimization=True: the reference accepts the call (and corrupts its state); here it runs
    as the plain layout's trajectory with a warning, single runs and ensembles alike; VGX_STRICT_MEMOPT=1 refuses instead."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    with helpers.quiet():
        two = Simulator(2, 2, 1, seed=11, memory_optimization=True)
        plain = Simulator(2, 2, 1, seed=11)
        one = Simulator(2, 1, 1, seed=11, memory_optimization=True)
    for s in (two, plain):
        s.set_mutation_rate(0.3); s.set_migration_probability(0.05)
    with pytest.warns(RuntimeWarning, match="memory_optimization"), helpers.quiet():
        two.simulate(3000)
    with helpers.quiet():
        plain.simulate(3000)
    assert np.array_equal(two.simulation.events.as_array(), plain.simulation.events.as_array())
    assert np.array_equal(two.simulation.infectious, plain.simulation.infectious)
    seen = np.nonzero(two.simulation.infectious.any(axis=0))[0]
    tb = two.simulation
    assert set(seen) <= set(tb.numToHap[:tb.currentHapNum])            # the table's bookkeeping covers what is there
    with pytest.warns(RuntimeWarning, match="memory_optimization"), helpers.quiet():
        one.simulate(10, method="tau")
    with pytest.warns(RuntimeWarning, match="memory_optimization"), helpers.quiet():
        Ensemble(two, 2).close()
    monkeypatch.setenv("VGX_STRICT_MEMOPT", "1")
    with pytest.raises(ValueError, match="one population"), helpers.quiet():
        two.simulate(100)
