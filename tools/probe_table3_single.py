import sys, os
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
from vgsim_amd.ensemble import Ensemble
for K in (2, 10, 50):
    sim = bench.make_table3(K, 0.001)
    ens = Ensemble(sim, 1)
    for kernel in ("solo", "lone"):
        for it in range(2):
            res = ens.simulate(200000, sample_size=10 ** 12, record_events=True, seeds=np.array([2023 + it], dtype=np.int64), kernel=kernel)
        print("K=%3d %-5s -> %-5s %.3e events/s" % (K, kernel, ens.engine.last_kernel, res.total_events / (res.kernel_ms * 1e-3)), flush=True)
    ens.close()
