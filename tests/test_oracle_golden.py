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
