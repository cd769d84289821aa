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


@pytest.mark.parametrize("name", DIRECT)
def test_fast_wave_kernel_integer_columns_match_oracle(oracle_mod, name):
    """The one-replicate-per-wavefront kernel's FAST path forced for every case (the automatic choice gives the one-class cases to
    the FAST row kernel, vgx_quadf.hip, at every ensemble size since round 3)."""
    hip = helpers.run_case_hip(name, mode="fast", kernel="wave").simulation
    ref = helpers.run_case_oracle(oracle_mod, name).simulation
    _assert_tier_b(hip, ref, name)


@pytest.mark.parametrize("name", ["g9_short", "p70", "stress_h256"])
def test_fast_request_on_a_general_model_runs_the_exact_fast_kernels(oracle_mod, name):
    """A general model (several classes / groups / NPIs: what the FAST row kernel refuses) asked for in FAST mode: the exact row /
    latency kernels run it — their output is what FAST promises, and they are the faster path (2.5e8 -> 1.4e9 events/s on the
    Table-3 model; tools/probe_fast_general.py).  Here: the kernel that ran, and results equal to the exact oracle bit for bit."""
    hip = helpers.run_case_hip(name, mode="fast").simulation
    assert hip._engine.last_kernel in ("solo", "quadg", "lone"), hip._engine.last_kernel
    helpers.assert_models_equal(hip, helpers.run_case_oracle(oracle_mod, name).simulation, name)


def test_ensembles_with_recombination_do_not_take_the_lane_kernel():
    """Round 3's automatic choice gave ensembles of models with recombination to the lane kernel (1.0e7 events/s at 16 384 replicates
    against 7.5e8 on the latency kernel and 6.4e8 on the general row kernel: tools/probe_recomb_ens.py)."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    ctor, phases = models.CASES["recomb_a"]
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    for R in (8, 4096):
        ens = Ensemble(sim, R)
        ens.simulate(500, sample_size=10 ** 9)
        assert ens.engine.last_kernel in ("solo", "quadg"), (R, ens.engine.last_kernel)
        ens.close()


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


# ---- FAST mode with the counter-based random stream (vgx_run_opts.mode = 2) -------------------------------------------
# Other random numbers than the reference's, so the chain cannot equal a reference fixture; the checker is the oracle fed
# with the same Philox stream (oracle.RNG_PHILOX: the two uniform draws of an iteration come from outputs 2 i and 2 i + 1
# of the stream of (seed, attempt)).  Same tier as FAST: integer columns, counters and compartments identical, times 1e-9.
@pytest.mark.parametrize("name", DIRECT)
def test_fast_philox_matches_oracle_on_the_same_stream(oracle_mod, name):
    hip = helpers.run_case_hip(name, mode="fast_philox").simulation
    ref = helpers.run_case_oracle(oracle_mod, name, log_mode=oracle_mod.RNG_PHILOX).simulation
    _assert_tier_b(hip, ref, name)


def test_fast_philox_is_another_trajectory_of_the_same_model():
    """Not the PCG64 chain, reproducible, and the stream's definition: the first uniform of an attempt is the low half of
    the Philox block with counter (0, 0, attempt, 'VGXs') and the seed as key."""
    import ctypes as C
    from vgsim_amd import _capi
    a = helpers.run_case_hip("g1", mode="fast_philox").simulation
    b = helpers.run_case_hip("g1", mode="fast_philox").simulation
    c = helpers.run_case_hip("g1", mode="fast").simulation
    assert np.array_equal(helpers.chain_of(a), helpers.chain_of(b))
    assert not np.array_equal(helpers.chain_of(a)[1:, :a.events.ptr], helpers.chain_of(c)[1:, :c.events.ptr])
    lib = _capi.load_library()
    # (the attempt that succeeded: earlier ones went extinct within 100 events and were restarted, pyx:414-418)
    att_a, att_c = a.good_attempt - 1, c.good_attempt - 1
    ctr, key, o = (C.c_uint32 * 4)(0, 0, att_a, 0x56475873), (C.c_uint32 * 2)(a.user_seed & 0xFFFFFFFF, a.user_seed >> 32), (C.c_uint32 * 4)()
    assert lib.vgx_test_philox(0, C.byref(ctr), C.byref(key), C.byref(o)) == 0
    u0 = float(((o[1] << 32) | o[0]) >> 11) / 2.0 ** 53
    # the first event's time is -log(u0) / (totalRate + totalMigrationRate) with the start state's rates: the same
    # denominator the PCG64 run divided by
    seq = np.random.Generator(np.random.PCG64(np.random.SeedSequence(c.user_seed, spawn_key=(att_c,))))
    u0_pcg = seq.random()
    ta, tc = helpers.chain_of(a)[0, 0], helpers.chain_of(c)[0, 0]
    assert ta / tc == pytest.approx(np.log(u0) / np.log(u0_pcg), rel=1e-9)
