"""Development aid: mode='fast' on general models (what the FAST row kernel refuses): events/s per kernel choice.
python tools/probe_fast_general.py [R] [events]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
for K in (2, 10, 100):
    out = {}
    for mode in ("exact", "fast"):
        sim = bench.make_table3(K, 0.001, 2023)
        RR = R if K < 100 else R // 4
        ens = Ensemble(sim, RR)
        ens.simulate(2000, sample_size=10 ** 12, mode=mode)
        ens2 = Ensemble(sim, RR)
        res = ens2.simulate(N, sample_size=10 ** 12, mode=mode)
        out[mode] = "%.3g ev/s (%s)" % (res.total_events / (res.kernel_ms * 1e-3), ens2.engine.last_kernel)
        ens.close(); ens2.close()
    print("Table 3 K=%d" % K, out, flush=True)
