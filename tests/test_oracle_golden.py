"""The CPU oracle against the golden vectors recorded from the reference itself (tests/golden/*.npz,
made by tests/golden/make_golden.py).  This is what pins the oracle: all six rows of every chain,
counters, success attempt, epidemic time and final compartments, for the nine reference goldens and
the additional cases of tests/models.py."""
import numpy as np
import pytest

import helpers
import models

DIRECT = [n for n in models.CASES if not n.startswith("tau_")]
TAU = [n for n in models.CASES if n.startswith("tau_")]
SLOW = {"example", "c2"}


@pytest.mark.parametrize("name", DIRECT)
def test_direct_matches_reference(oracle_mod, name):
    sim = helpers.run_case_oracle(oracle_mod, name)
    helpers.check_against_golden(sim.simulation, name, exact_time=helpers.libm_matches_fixture_host())


@pytest.mark.parametrize("name", TAU)
def test_tau_matches_reference(oracle_mod, name):
    """Direct warm-up then Poisson tau-leaping, draw-exact (same PCG64 stream, numpy's random_poisson
    restated): MULTITYPE records incl. step times and dense multievent index ranges, all counters."""
    sim = helpers.run_case_oracle(oracle_mod, name)
    helpers.check_against_golden(sim.simulation, name, exact_time=helpers.libm_matches_fixture_host())


@pytest.mark.parametrize("name", [n for n in DIRECT if n not in SLOW])
def test_sparse_is_bit_identical(oracle_mod, name):
    """Visiting only occupied haplotypes (what the HIP engine does) changes no bit of the output."""
    sim = helpers.run_case_oracle(oracle_mod, name, sparse=True)
    helpers.check_against_golden(sim.simulation, name, exact_time=helpers.libm_matches_fixture_host())


@pytest.mark.parametrize("name", ["g9_short", "stress_h64", "c3_s5_p16"])
def test_portable_log_time_tolerance(oracle_mod, name):
    """With the portable log (the device's), integer rows stay exact and times stay within 1e-12."""
    sim = helpers.run_case_oracle(oracle_mod, name, log_mode=1)
    helpers.check_against_golden(sim.simulation, name, exact_time=False, rtol_time=1e-12)


REF_TESTING = "/root/reference/testing"


@pytest.mark.parametrize("k", range(1, 10))
def test_upstream_reference_blobs_if_present(oracle_mod, k):
    """The reference's OWN goldens ``testing/reference_{1..9}.npy`` (``check_simulator.py:6-32`` compares all six rows
    with ``!=``) are missing from the checkout (``.MISSING_LARGE_BLOBS``).  They are the only witnesses of the seed ->
    uniform-stream mapping of the absent third-party ``mc_lib.rndm.RndmWrapper``; wherever they are available this test
    pins the oracle's restatement of it.  Skipped while they are absent: parity stays "unpinned at the
    RNG wrapper" (DESIGN.md §2)."""
    import os
    path = os.path.join(REF_TESTING, "reference_%d.npy" % k)
    if not os.path.exists(path) or os.path.getsize(path) < 1000:
        pytest.skip("upstream golden %s is not in this checkout" % path)
    want = np.load(path)
    sim = helpers.run_case_oracle(oracle_mod, "g%d" % k)
    got = sim.simulation.events.as_array()
    assert got.shape == want.shape
    assert np.array_equal(got[1:], want[1:]), "integer rows differ from the upstream golden"
    if helpers.libm_matches_fixture_host():
        assert np.array_equal(got[0], want[0]), "time row differs from the upstream golden"
    else:
        np.testing.assert_allclose(got[0], want[0], rtol=1e-12, atol=0)
