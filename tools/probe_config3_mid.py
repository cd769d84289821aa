"""Development aid: BASELINE config 3 (65 536 haplotypes x 64 populations) at ensemble sizes around the automatic choice's
thresholds, per kernel (events/s of device time).  python tools/probe_config3_mid.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
for R, n in ((256, 20000), (1024, 20000), (2048, 20000), (4096, 20000), (8192, 20000)):
    row = {}
    for kernel in ("auto", "wave", "quad"):
        s = bench.make_simulator(2020)
        try:
            ens = Ensemble(s, R)
            res = ens.simulate(n, sample_size=10 ** 12, kernel=kernel)
            row[kernel] = "%.3g (%s)" % (res.total_events / (res.kernel_ms * 1e-3), ens.engine.last_kernel)
            ens.close()
        except Exception as ex:
            row[kernel] = "refused: %s" % str(ex)[:60]
    print("config 3, R=%d: %s" % (R, row), flush=True)
