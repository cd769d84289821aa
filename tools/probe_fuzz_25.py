"""Development aid: a seeded random model of tests/test_hip_fuzz.py whose tau call gives up ("tau underflow"): the state after the
steps that were accepted before.  python tools/probe_fuzz_25.py SEED STEPS"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_hip_fuzz as F, helpers
seed, k = int(sys.argv[1]), int(sys.argv[2])
sim, n = F.build(seed)
m = sim.simulation
with helpers.quiet():
    sim.simulate(min(n, 600), sample_size=10 ** 9)
with helpers.quiet():
    sim.simulate(k, sample_size=10 ** 12, method="tau")
I = m.infectious
print(seed, "after", k, "steps: t=%.6f infected %d min I %d (cells below zero: %d) min S %d" % (m.currentTime, I.sum(), I.min(), (I < 0).sum(), m.susceptible.min()),
      "migration diag %.3f" % m.migrationRates[0, 0])
