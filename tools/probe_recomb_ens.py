"""Development aid: ensembles of a model with recombination on the kernels that take it (events/s of device time).
python tools/probe_recomb_ens.py [R] [events]"""
import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import models, helpers
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
for name in ("recomb_a", "recomb_pos"):
    ctor, phases = models.CASES[name]
    out = {}
    for kernel in ("auto", "quadg", "solo", "lane", "wave"):
        with helpers.quiet():
            sim = Simulator(**ctor)
        phases[0][0](sim)
        ens = Ensemble(sim, R)
        try:
            ens.simulate(2000, sample_size=10 ** 12, kernel=kernel)      # warm-up launch
            ens2 = Ensemble(sim, R)
            res = ens2.simulate(N, sample_size=10 ** 12, kernel=kernel)
            out[kernel] = "%.3g ev/s (%s)" % (res.total_events / (res.kernel_ms * 1e-3), ens2.engine.last_kernel)
            ens2.close()
        except Exception as ex:
            out[kernel] = "refused: " + str(ex)[:60]
        ens.close()
    m = sim.simulation
    print(name, "sites %d H %d P %d S %d" % (m.sites, m.hapNum, m.popNum, m.susNum), out, flush=True)
