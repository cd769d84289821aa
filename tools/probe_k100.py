import sys, os
sys.path.insert(0, "/root/repo")
import bench
from vgsim_amd.ensemble import Ensemble
for R in ((4096,) if len(sys.argv) > 1 and sys.argv[1] == "one" else (4096, 5120, 8192, 16384)):
    sim = bench.make_table3(100, 0.001, 2023)
    ens = Ensemble(sim, R)
    res = ens.simulate(20000, sample_size=10 ** 12)
    print("K=100 R=%d: %.3g ev/s (%s)" % (R, res.total_events / (res.kernel_ms * 1e-3), ens.engine.last_kernel), flush=True)
    ens.close()
