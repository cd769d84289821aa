"""Development aid: where the row kernels overtake vgx_lone as the ensemble grows (config 3 and its general variant):
python tools/probe_lone_crossover.py [events]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
for name, make, row in (("config3", lambda: bench.make_simulator(2020), "quad"), ("config3_general", bench.make_general_c3, "quadg")):
    for R in (512, 768, 1024, 1536, 2048):
        ens = Ensemble(make(), R)
        out = []
        for kernel in ("lone", row):
            res = None
            for it in range(2):
                res = ens.simulate(N, sample_size=10 ** 12, record_events=True, seeds=2020 + np.arange(R, dtype=np.int64), kernel=kernel)
            out.append("%s %.3e" % (ens.engine.last_kernel, res.total_events / (res.kernel_ms * 1e-3)))
        print("%-16s R=%5d  %s" % (name, R, "   ".join(out)), flush=True)
        ens.close()
