"""Development aid: one trajectory / small ensembles of BASELINE config 3 as a GENERAL model (bench.make_general_c3) on the kernels that take it:
python tools/probe_lone_general.py [events]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
for R in (1, 64, 256, 512):
    ens = Ensemble(bench.make_general_c3(), R)
    for kernel in ("lone", "quadg", "wave", "auto"):
        res = None
        for it in range(2):
            res = ens.simulate(N, sample_size=10 ** 12, record_events=True, seeds=2020 + np.arange(R, dtype=np.int64), kernel=kernel)
        print("R=%4d %-6s -> %-6s %.3e events/s (%.1f ms)" % (R, kernel, ens.engine.last_kernel, res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms), flush=True)
    ens.close()
