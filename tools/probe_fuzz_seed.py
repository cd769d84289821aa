"""Development aid: one seeded random model of tests/test_hip_fuzz.py, tau step by step: python tools/probe_fuzz_seed.py SEED [ENV=VALUE ...]"""
import sys, os
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1); os.environ[k] = v
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_hip_fuzz as F, helpers
seed = int(sys.argv[1])
sim, n = F.build(seed)
m = sim.simulation
with helpers.quiet():
    sim.simulate(min(n, 600), sample_size=10 ** 9)
print("dims sites %d H %d P %d S %d, infected %d, sizes %s" % (m.sites, m.hapNum, m.popNum, m.susNum, m.globalInfectious, m.sizes[:8]))
for k in range(1, 27):
    try:
        with helpers.quiet():
            sim.simulate(1, sample_size=10 ** 12, method="tau")
    except Exception as ex:
        print("step", k, "FAILED", str(ex)[:200]); break
    I = m.infectious
    print("step", k, "t=%.4f" % m.currentTime, "infected", int(I.sum()), "min I", int(I.min()), "min S", int(m.susceptible.min()),
          "totI", m.totalInfectious[:6], "S+I==size", bool(np.array_equal(m.susceptible.sum(axis=1) + I.sum(axis=1), m.sizes)), "lockdown", m.lockdownON[:6])
print("S", m.susceptible[:4]); print("I rows sums", m.infectious.sum(axis=1)[:6])
