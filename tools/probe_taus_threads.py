"""Development aid: the on-device tau loop (vgx_taus.hip) at workgroup sizes 64 / 256 / 512 over ensemble sizes (steps/s of device time, all
replicates together; VGX_TAUS_THREADS forces a size).  python tools/probe_taus_threads.py"""
import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
for sites, pops in ((2, 3), (3, 4), (4, 5)):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=pops, seed=7)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
    s.set_total_migration_probability(0.002); s.set_population_size(10 ** 6)
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(2000, sample_size=10 ** 12)
    for R in (64, 256, 512, 1024, 2048, 4096):
        row = {}
        for tt in ("64", "256", "512", ""):
            if tt:
                os.environ["VGX_TAUS_THREADS"] = tt
            else:
                os.environ.pop("VGX_TAUS_THREADS", None)
            ens = Ensemble(s, R)
            for it in range(2):
                res = ens.simulate_tau(300, sample_size=10 ** 15, seeds=7 + it * R + np.arange(R, dtype=np.int64))
            row[tt or "auto"] = "%.3g" % (float(res.loop_iterations.sum()) / (res.kernel_ms * 1e-3))
            ens.close()
        print("%dx%d R=%d: %s" % (4 ** sites, pops, R, row), flush=True)
