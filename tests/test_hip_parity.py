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


@pytest.mark.parametrize("name", DIRECT)
def test_direct_vs_reference_goldens(name):
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


def test_memory_optimization_flag_changes_bookkeeping_only():
    """The engine is sparse in the haplotype dimension by construction: ``memory_optimization=True`` yields the chain of
    the plain layout and maintains the reference's haplotype table (sorted program numbers, pyx:355-377)."""
    from vgsim_amd import Simulator

    def run(flag):
        with helpers.quiet():
            s = Simulator(4, 2, 1, seed=11, memory_optimization=flag)
            s.set_mutation_rate(0.2)
            s.set_migration_probability(0.01)
            s.simulate(4000)
            after_direct.append(s.simulation.currentHapNum)
            s.simulate(300, method="tau", sample_size=10 ** 9)
        return s.simulation
    after_direct = []
    a, b = run(False), run(True)
    assert after_direct[0] == 256 and 1 < after_direct[1] < 256     # the table grows with the haplotypes seen
    assert np.array_equal(a.events.as_array(), b.events.as_array()) and np.array_equal(a.infectious, b.infectious)
    n = b.currentHapNum
    seen = np.unique(np.concatenate(([0], b.events.newHaplotypes[:b.events.ptr][b.events.types[:b.events.ptr] == 3],
                                     b.multievents.newHaplotypes[:b.multievents.ptr][b.multievents.types[:b.multievents.ptr] == 3])))
    assert after_direct[1] <= n == len(seen) <= 256 and np.array_equal(b.numToHap[:n], seen)
    assert np.array_equal(b.hapToNum[seen], np.arange(n)) and n <= b.maxHapNum <= 256 and len(b.numToHap) == b.maxHapNum
    assert a.currentHapNum == 256 and np.array_equal(a.numToHap, np.arange(256))
