"""Ensembles on the single-trajectory kernel beside the row kernels: events/s of device time (development aid)."""
import json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
out = {}
for K, R, n in ((2, 16384, 20000), (10, 16384, 20000), (100, 4096, 5000)):
    for k in sys.argv[1].split(","):
        ens = Ensemble(bench.make_table3(K, 0.001), R)
        for it in range(2):
            res = ens.simulate(n, sample_size=10 ** 12, record_events=True, seeds=2023 + it * R + np.arange(R, dtype=np.int64), kernel=k)
        out["K=%d R=%d %s" % (K, R, k)] = res.total_events / (res.kernel_ms * 1e-3)
        ens.close()
    print(json.dumps(out), flush=True)
