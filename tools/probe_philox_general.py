"""Development aid: the counter-based stream (mode='fast_philox') on general models — 16 384 replicates of the Table-3 model on the
general row kernel (its exact arithmetic on the Philox stream) beside the exact mode: python tools/probe_philox_general.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
for K, R in ((2, 16384), (10, 16384), (100, 4096)):
    ens = Ensemble(bench.make_table3(K, 0.001), R)
    for mode in ("exact", "fast", "fast_philox"):
        res = None
        for it in range(2):
            res = ens.simulate(50000, sample_size=10 ** 12, record_events=True, mode=mode, seeds=2023 + np.arange(R, dtype=np.int64))
        print("K=%3d R=%5d %-11s -> %-6s %.3e events/s" % (K, R, mode, ens.engine.last_kernel, res.total_events / (res.kernel_ms * 1e-3)), flush=True)
    ens.close()
