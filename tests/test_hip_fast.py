"""FAST mode of the direct path (GPU): order-free sums (class-aggregated infection rate, integer prefix
search over the occupancy list, factored BirthRate, tree scans over populations) with the SAME PCG64 stream
and event semantics as EXACT mode.  Reordered floating-point sums perturb rates at the 1e-16 level, so on
the same seed the INTEGER columns of the event log, the counters and the final compartments must be
identical to the oracle's (SURVEY.md §7.1 "Tier B", §7.4 row "factored BirthRate"), and the time column
must agree within 1e-9 relative (tolerance of this mode).  Holds for any number of rate classes: the occupancy
lists keep haplotype order, so the same uniform selects the same haplotype."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu

# every direct case of the parity suite: one rate class (class-aggregated rate + integer prefix search) and several
# classes (tree scans over the occupancy list in haplotype order)
DIRECT = [n for n, (_, ph) in models.CASES.items() if all(kw.get("method", "direct") == "direct" for _, kw in ph)
          and n not in models.RECOMBINATION_CASES]   # recombination runs in exact mode only (lane kernel)
RTOL_TIME = 1e-9


def _assert_tier_b(got, want, what):
    assert got.events.ptr == want.events.ptr, "%s events.ptr %d != %d" % (what, got.events.ptr, want.events.ptr)
    a, b = helpers.chain_of(got), helpers.chain_of(want)
    ptr = want.events.ptr
    assert np.array_equal(a[1:, :ptr], b[1:, :ptr]), what + " " + helpers.describe_first_diff(a[1:], b[1:], ptr)
    np.testing.assert_allclose(a[0, :ptr], b[0, :ptr], rtol=RTOL_TIME, atol=0.0, err_msg=what + " times")
    for k in got.COUNTERS + ("good_attempt", "globalInfectious"):
        assert getattr(got, k) == getattr(want, k), "%s %s: %r != %r" % (what, k, getattr(got, k), getattr(want, k))
    assert abs(got.currentTime - want.currentTime) <= RTOL_TIME * abs(want.currentTime)
    assert np.array_equal(got.susceptible, want.susceptible), what + " susceptible"
    assert np.array_equal(got.infectious, want.infectious), what + " infectious"
    assert np.array_equal(got.lockdownON, want.lockdownON), what + " lockdownON"
    assert got.loc.states == want.loc.states and got.loc.populationsId == want.loc.populationsId, what + " lockdown log"
    np.testing.assert_allclose(got.loc.times, want.loc.times, rtol=RTOL_TIME, atol=0.0)


@pytest.mark.parametrize("name", DIRECT)
def test_fast_integer_columns_match_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name, mode="fast").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    _assert_tier_b(hip, ref, name)


def test_fast_ensemble_matches_exact_ensemble():
    """Replicate ensembles: FAST and EXACT agree on every replicate's counters and final compartments."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    with helpers.quiet():
        sim, phases = models.build(Simulator, "c3_s5_p16")
        phases[0][0](sim)
    R = 48
    out = {}
    for mode in ("exact", "fast"):
        ens = Ensemble(sim, R)
        res = ens.simulate(4000, sample_size=10 ** 9, record_events=True, traj_points=33, traj_window=(0.0, 8.0), mode=mode)
        out[mode] = (res.events.copy(), res.loop_iterations.copy(), res.restarts.copy(),
                     [ens.replicate_state(r).infectious.copy() for r in (0, 7, R - 1)],
                     [ens.replicate_events(r)[1:] for r in (0, 7, R - 1)], ens.trajectories().copy())
        ens.close()
    e, f = out["exact"], out["fast"]
    for i in range(3):
        assert np.array_equal(e[i], f[i])
    for a, b in zip(e[3], f[3]):
        assert np.array_equal(a, b)
    for a, b in zip(e[4], f[4]):
        assert np.array_equal(a, b)
    assert np.array_equal(e[5], f[5])
