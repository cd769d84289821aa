"""The samplers of the tau-leap kernels on their own (test hooks of include/vgx.h).

* Philox4x32-10: the known-answer vectors of the Random123 distribution (Salmon, Moraes, Dror, Shaw, SC'11), for the host
  build of vgx_rng.h (CPU suite) and for the device's (GPU suite).
* Poisson: the device sampler replaces numpy's ``random_poisson`` (pyx:2531-2532) with inversion below a mean of 10 and PTRS
  (Hoermann 1993) from 10 on; both branches are tested against the exact probability mass function with a chi-square
  statistic over 2^18 draws per mean (means on both sides of the switch at 10 included)."""
import ctypes as C
import math

import numpy as np
import pytest

KAT = [  # counter, key, output (Random123 kat_vectors, philox4x32 10 rounds)
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def _philox(on_device, ctr, key):
    from vgsim_amd import _capi
    lib = _capi.load_library()
    c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
    assert lib.vgx_test_philox(int(on_device), C.byref(c), C.byref(k), C.byref(o)) == 0
    return tuple(o)


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers_host(ctr, key, want):
    assert _philox(False, ctr, key) == want


@pytest.mark.gpu
@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers_device(ctr, key, want):
    assert _philox(True, ctr, key) == want


def _poisson_pmf(lam, kmax):
    logp = -lam + np.arange(kmax + 1) * math.log(lam) - np.array([math.lgamma(k + 1.0) for k in range(kmax + 1)])
    return np.exp(logp)


@pytest.mark.gpu
@pytest.mark.parametrize("lam", [0.01, 1.0, 9.9, 10.0, 50.0, 1.0e4])
def test_device_poisson_matches_the_exact_pmf(lam):
    from vgsim_amd import _capi
    lib = _capi.load_library()
    n = 1 << 18
    out = np.zeros(n, dtype=np.int64)
    assert lib.vgx_test_poisson(float(lam), n, 20201103, out.ctypes.data_as(C.POINTER(C.c_int64))) == 0
    assert (out >= 0).all()
    # moments: mean and variance of a Poisson are both lam
    se_mean = math.sqrt(lam / n)
    assert abs(out.mean() - lam) <= 4.5 * se_mean, (out.mean(), lam)
    se_var = math.sqrt((lam + 2 * lam * lam) / n)           # Var(s^2) ~ (mu4 - sigma^4)/n, mu4 = lam + 3 lam^2
    assert abs(out.var(ddof=1) - lam) <= 4.5 * se_var, (out.var(ddof=1), lam)
    # chi-square against the exact pmf, tail classes merged until every expected count is >= 8
    lo, hi = max(int(lam - 8 * math.sqrt(lam) - 10), 0), int(lam + 8 * math.sqrt(lam) + 20)
    pmf = _poisson_pmf(lam, hi)
    exp = pmf[lo:hi + 1] * n
    exp[0] += pmf[:lo].sum() * n
    exp[-1] += max(1.0 - pmf.sum(), 0.0) * n
    obs = np.bincount(np.clip(out, lo, hi) - lo, minlength=hi - lo + 1).astype(float)
    # merge from both ends
    while len(exp) > 2 and exp[0] < 8:
        exp[1] += exp[0]; obs[1] += obs[0]; exp, obs = exp[1:], obs[1:]
    while len(exp) > 2 and exp[-1] < 8:
        exp[-2] += exp[-1]; obs[-2] += obs[-1]; exp, obs = exp[:-1], obs[:-1]
    chi2 = float(((obs - exp) ** 2 / exp).sum())
    dof = len(exp) - 1
    # Wilson-Hilferty bound at about 4.5 sigma: a correct sampler exceeds it with probability < 1e-5
    bound = dof * (1.0 - 2.0 / (9 * dof) + 4.5 * math.sqrt(2.0 / (9 * dof))) ** 3
    assert chi2 <= bound, "chi-square %.1f over %d classes (bound %.1f)" % (chi2, dof + 1, bound)
