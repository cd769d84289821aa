"""Development aid: ONE trajectory of the Table-3 models on vgx_lone.hip beside vgx_solo.hip: python tools/probe_lone_table3.py [events]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
for K in (2, 10, 50):
    for M in (0.001, 0.1):
        ens = Ensemble(bench.make_table3(K, M), 1)
        for kernel in ("lone", "solo"):
            res = None
            for it in range(2):
                res = ens.simulate(N, sample_size=10 ** 12, record_events=True, seeds=np.array([2023], dtype=np.int64), kernel=kernel)
            print("K=%3d M=%.3f %-5s -> %-5s %.3e events/s (%.1f ms)" % (K, M, kernel, ens.engine.last_kernel, res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms), flush=True)
        ens.close()
