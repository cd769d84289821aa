"""Parity tests proper (GPU): the HIP engine, called through the C ABI exactly as ``Simulator.simulate``
does, against (a) the CPU oracle on the same seeded inputs — bit for bit on all six log rows, counters,
lockdown log and final compartments (the oracle running the same portable log as the device) — and
(b) the golden vectors recorded from the reference itself — integer rows exact, time row within 1e-12
relative (the reference's glibc log vs the device's fdlibm-style log differ by <= 1 ulp per step)."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu

DIRECT = [n for n, (_, ph) in models.CASES.items() if all(kw.get("method", "direct") == "direct" for _, kw in ph)]


LANE_OK = [n for n in DIRECT if not n.startswith(("stress_h256", "c3_", "p70"))]   # popNum <= 16, popNum * hapNum <= 1024


@pytest.mark.parametrize("name", DIRECT)
def test_direct_bit_exact_vs_oracle(oracle_mod, name):
    hip = helpers.run_case_hip(name).simulation
    ref = helpers.run_case_oracle(oracle_mod, name, log_mode=oracle_mod.LOG_PORTABLE).simulation
    helpers.assert_models_equal(hip, ref, name)


@pytest.mark.parametrize("name", DIRECT)
def test_direct_vs_reference_goldens(name):
    hip = helpers.run_case_hip(name).simulation
    helpers.check_against_golden(hip, name, exact_time=False, rtol_time=1e-12)


@pytest.mark.parametrize("name", LANE_OK)
def test_lane_kernel_bit_exact_vs_oracle(oracle_mod, name):
    """The one-replicate-per-lane kernel (vgx_lanes.hip: the reference's serial loops on dense per-replicate state),
    forced for every small case of the suite."""
    hip = helpers.run_case_hip(name, kernel="lane").simulation
    ref = helpers.run_case_oracle(oracle_mod, name, log_mode=oracle_mod.LOG_PORTABLE).simulation
    helpers.assert_models_equal(hip, ref, name)
