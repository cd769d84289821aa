"""Development aid: BASELINE config 2 (one haplotype, one population) per kernel and ensemble size (events/s of device time).
python tools/probe_config2.py"""
import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
for R, n in ((4096, 50000), (16384, 50000), (65536, 20000), (262144, 10000)):
    row = {}
    for kernel in ("auto", "solo", "quad", "lane"):
        with contextlib.redirect_stdout(io.StringIO()):
            s = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
        s.set_transmission_rate(4.0); s.set_recovery_rate(1.5); s.set_sampling_rate(0.3); s.set_population_size(10 ** 6)
        try:
            ens = Ensemble(s, R)
            res = ens.simulate(n, sample_size=10 ** 12, kernel=kernel)
            row[kernel] = "%.3g (%s)" % (res.total_events / (res.kernel_ms * 1e-3), ens.engine.last_kernel)
            ens.close()
        except Exception as ex:
            row[kernel] = "refused"
    print("config 2, R=%d: %s" % (R, row), flush=True)
