"""The oracle's restatement of the absent third-party RNG wrapper (mc_lib.rndm.RndmWrapper) and of numpy's
random_poisson, checked against numpy itself; the portable logarithm against libm."""
import ctypes as C
import math

import numpy as np
import pytest


@pytest.mark.parametrize("seed,attempt", [(2020, 0), (2020, 2), (0, 0), (1, 7), (1234, 199), (2 ** 31 + 5, 3),
                                          (2 ** 40 + 12345, 1), (2 ** 63 - 1, 4)])
def test_pcg64_seeding_and_stream_match_numpy(oracle_mod, seed, attempt):
    state, inc, doubles = oracle_mod.pcg64_stream(seed, attempt, 64)
    bg = np.random.PCG64(np.random.SeedSequence(seed, spawn_key=(attempt,)))
    st = bg.state["state"]
    assert st["state"] == state and st["inc"] == inc
    assert np.array_equal(doubles, np.random.Generator(bg).random(64))


def test_known_answer_seed_2020_attempt_2(oracle_mod):
    """SURVEY.md §8(c) G15."""
    state, inc, d = oracle_mod.pcg64_stream(2020, 2, 4)
    assert state == 0x5b28c82152932c207cecf11bb281bc6c and inc == 0x06b1cc5a588721b52dc6c8ea86b185f7
    assert d.tolist() == [0.44016044650747377, 0.08583779961275151, 0.20313668834338472, 0.3831094889306431]


def test_pcg64_advance(oracle_mod):
    L = oracle_mod.lib()
    g1, g2 = oracle_mod.VgoPcg64(), oracle_mod.VgoPcg64()
    L.vgo_pcg64_seed(C.byref(g1), 77, 1)
    L.vgo_pcg64_seed(C.byref(g2), 77, 1)
    for _ in range(1000):
        L.vgo_pcg64_next64(C.byref(g1))
    L.vgo_pcg64_advance(C.byref(g2), 0, 1000)
    assert L.vgo_pcg64_next64(C.byref(g1)) == L.vgo_pcg64_next64(C.byref(g2))


@pytest.mark.parametrize("lam", [0.0, 1e-9, 0.3, 2.5, 9.999, 10.0, 37.2, 1e3, 1e6, 3.3e9])
def test_poisson_is_draw_exact_with_numpy(oracle_mod, lam):
    L = oracle_mod.lib()
    g = oracle_mod.VgoPcg64()
    L.vgo_pcg64_seed(C.byref(g), 99, 5)
    gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(99, spawn_key=(5,))))
    mine = [L.vgo_poisson(C.byref(g), lam) for _ in range(2000)]
    ref = gen.poisson(lam, 2000).tolist()
    assert mine == ref
    assert L.vgo_pcg64_next64(C.byref(g)) == int(gen.bit_generator.random_raw())  # same stream position


def test_portable_log_within_one_ulp(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.Generator(np.random.PCG64(7))
    xs = np.concatenate([rng.random(20000), rng.random(2000) * 1e-12, 1.0 - rng.random(2000) * 1e-9,
                         np.array([2.0 ** -53, 0.5, 1.0 - 2.0 ** -53, 0.999, 1e-300])])
    worst = 0.0
    for x in xs:
        a, b = L.vgo_portable_log(float(x)), math.log(float(x))
        if b != 0.0:
            worst = max(worst, abs(a - b) / math.ulp(b))
        else:
            assert a == 0.0
    assert worst <= 1.0, worst
    assert L.vgo_portable_log(0.0) == -math.inf


def test_oracle_counter_based_stream_matches_the_library_definition(oracle_mod):
    """The oracle's optional Philox stream (oracle.RNG_PHILOX: the checker of the engine's FAST mode 2) is the stream
    include/vgx.h defines: output 0 of (seed, attempt) = low 64 bits of Philox4x32-10 with counter (0, 0, attempt, 'VGXs')
    and the seed as key (libvgx's host Philox, itself pinned on Random123 vectors in test_samplers.py).  The first event's
    time is -log(u0) / (totalRate + totalMigrationRate) with the start state's rates, the same denominator in both runs."""
    import helpers
    import models
    from vgsim_amd import Simulator, _capi

    def first_event(log_mode):
        with helpers.quiet():
            sim, phases = models.build(Simulator, "g1")
            phases[0][0](sim)
        m = sim.simulation
        assert oracle_mod.run_direct(m, 2000, 10 ** 9, -1, 200, log_mode=log_mode) == 0
        return m

    a, b = first_event(oracle_mod.RNG_PHILOX), first_event(0)
    lib = _capi.load_library()
    ctr = (C.c_uint32 * 4)(0, 0, a.good_attempt - 1, 0x56475873)
    key = (C.c_uint32 * 2)(a.user_seed & 0xFFFFFFFF, a.user_seed >> 32)
    o = (C.c_uint32 * 4)()
    assert lib.vgx_test_philox(0, C.byref(ctr), C.byref(key), C.byref(o)) == 0
    u0 = float(((o[1] << 32) | o[0]) >> 11) / 2.0 ** 53
    u0_pcg = np.random.Generator(np.random.PCG64(np.random.SeedSequence(b.user_seed, spawn_key=(b.good_attempt - 1,)))).random()
    assert a.events.times[0] / b.events.times[0] == pytest.approx(math.log(u0) / math.log(u0_pcg), rel=1e-12)
    assert not np.array_equal(a.events.types[:a.events.ptr], b.events.types[:b.events.ptr])
