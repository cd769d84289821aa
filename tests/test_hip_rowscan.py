"""K3, the dense propensity row pass (vgx_rowscan.hip behind ``vgx_propensity_scan``), on the GPU:
  * against the oracle's dense rate caches after ``UpdateAllRates`` (the reference's own arrays eventHapPopRate,
    tEventHapPopRate, hapPopRate, susceptHapPopRate, infectPopRate of src/_BirthDeath.pyx:279-351 in the reference's
    operation order): per-haplotype rates within 1e-13 relative (BirthRate is factored through the row's contact sum),
    row totals within 1e-12 (tree-order sums);
  * the choice against ``fastChoose`` restated in numpy on the kernel's own hapPopRate: the index is the first whose serial
    running sum reaches r (entries within 1e-12 of the threshold may go either way), the recycled random number follows
    fast_choose.pxi:31;
  * ragged shapes (H not a multiple of the tile, several susceptibility groups, zero rows)."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu


def _contact(m):
    """K[pi] = sum_pn m[pi,pn]^2 * cd[pn] / actualSizes[pn], with the recomputed diagonal (pyx:289-297)."""
    P = m.popNum
    mig = m.migrationRates.copy()
    asz = np.zeros(P)
    for p1 in range(P):
        mig[p1, p1] = 1.0 - (mig[p1].sum() - mig[p1, p1])
    for p1 in range(P):
        asz[p1] = sum(mig[p2, p1] * m.sizes[p2] for p2 in range(P))
    return (mig ** 2 * m.contactDensity[None, :] / asz[None, :]).sum(axis=1)


def _check_choice(hpr, total, u, chosen, rn):
    r = total * u
    pre = np.cumsum(hpr)
    k = int(chosen)
    assert hpr[k] > 0.0
    before = pre[k] - hpr[k]
    eps = 1e-12 * max(total, 1e-300)
    assert before < r + eps and pre[k] >= r - eps, (k, before, pre[k], r)
    assert rn == pytest.approx((r - before) / hpr[k], abs=1e-6)


@pytest.mark.parametrize("name,n_events", [("stress_h64", 20000), ("stress_h256", 8000), ("g9_short", 3000), ("c3_s5_p16", 4000)])
def test_rows_match_the_oracles_dense_rate_caches(oracle_mod, name, n_events):
    from vgsim_amd import Simulator, _capi
    with helpers.quiet():
        sim, phases = models.build(Simulator, name)
        phases[0][0](sim)
    m = sim.simulation
    assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200) == 0      # a populated state
    st = oracle_mod.update_all_rates(m)                                    # the reference's caches for that state
    P, H, S = m.popNum, m.hapNum, m.susNum
    rng = np.random.default_rng(1)
    u = rng.random(P)
    out = _capi.propensity_scan(m.infectious, st.eventHapPopRate[:, :, 1:4], m.numToHap[:H], m.bRate, m.susceptibility,
                                m.susceptible.astype(float), _contact(m), u)
    np.testing.assert_allclose(out["susceptHapPopRate"], st.susceptHapPopRate, rtol=1e-15, atol=0)
    np.testing.assert_allclose(out["birthRate"], st.eventHapPopRate[:, :, 0], rtol=1e-13, atol=0)
    np.testing.assert_allclose(out["tEvent"], st.tEventHapPopRate, rtol=1e-13, atol=0)
    np.testing.assert_allclose(out["hapPopRate"], st.hapPopRate, rtol=1e-13, atol=0)
    np.testing.assert_allclose(out["rowTotal"], st.infectPopRate, rtol=1e-12, atol=0)
    for pi in range(P):
        if out["rowTotal"][pi] > 0.0:
            _check_choice(out["hapPopRate"][pi], out["rowTotal"][pi], u[pi], out["chosen"][pi], out["rnOut"][pi])


@pytest.mark.parametrize("rows,H,S", [(3, 1, 1), (5, 1023, 1), (4, 1025, 2), (2, 4099, 3), (7, 65536, 1)])
def test_rows_against_numpy_ragged_shapes(rows, H, S):
    from vgsim_amd import _capi
    rng = np.random.default_rng(rows * 1000 + H)
    inf = rng.integers(0, 4, size=(rows, H)) * (rng.random((rows, H)) < 0.3)
    inf[0] = 0 if rows > 2 else inf[0]                       # an empty row: total 0
    r123 = rng.random((rows, H, 3))
    n2h = rng.permutation(H)
    b, sig = 1.0 + rng.random(H), rng.random((H, S))
    sus, K, u = rng.integers(1, 10 ** 6, size=(rows, S)).astype(float), rng.random(rows) * 1e-6, rng.random(rows)
    out = _capi.propensity_scan(inf, r123, n2h, b, sig, sus, K, u)
    x = sus[:, None, :] * sig[n2h][None, :, :]
    ws = np.zeros((rows, H))
    for sn in range(S):
        ws = ws + x[:, :, sn]
    birth = b[n2h][None, :] * (ws * K[:, None])
    te = ((birth + r123[:, :, 0]) + r123[:, :, 1]) + r123[:, :, 2]
    hpr = te * inf
    assert np.array_equal(out["susceptHapPopRate"], x) and np.array_equal(out["birthRate"], birth)
    assert np.array_equal(out["tEvent"], te) and np.array_equal(out["hapPopRate"], hpr)        # elementwise: bit for bit
    np.testing.assert_allclose(out["rowTotal"], hpr.sum(axis=1), rtol=1e-12, atol=0)
    for r in range(rows):
        if out["rowTotal"][r] > 0.0:
            _check_choice(hpr[r], out["rowTotal"][r], u[r], out["chosen"][r], out["rnOut"][r])
