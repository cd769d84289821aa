import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import models, helpers
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
for name in ("g6_short", "g1", "g5", "c3_s5_p16"):
    if name not in models.CASES: continue
    ctor, phases = models.CASES[name]
    for R in (2048, 4096, 8192, 16384):
        row = {}
        for kernel in ("auto", "solo", "quad"):
            try:
                with helpers.quiet():
                    sim = Simulator(**ctor)
                phases[0][0](sim)
                ens = Ensemble(sim, R)
                res = ens.simulate(20000, sample_size=10 ** 12, kernel=kernel)
                row[kernel] = "%.3g (%s)" % (res.total_events / (res.kernel_ms * 1e-3), ens.engine.last_kernel)
                ens.close()
            except Exception as ex:
                row[kernel] = "refused"
        m = sim.simulation
        print("%s (H %d P %d S %d) R=%d: %s" % (name, m.hapNum, m.popNum, m.susNum, R, row), flush=True)
