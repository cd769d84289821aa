"""Development aid: lane-per-replicate vs wave-per-replicate kernel on small models (device time of one launch)."""
import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import models
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble

def sim_of(name):
    with contextlib.redirect_stdout(io.StringIO()):
        s, ph = models.build(Simulator, name)
        ph[0][0](s)
    return s

for name, R, N in (("c2", 1, 200000), ("c2", 16384, 100000), ("c2", 262144, 20000), ("g9", 1, 100000), ("g9", 16384, 20000), ("g9", 131072, 5000),
                   ("stress_h64", 16384, 10000), ("c3_s5_p16", 4096, 6000)):
    for kernel in ("wave", "lane"):
        try:
            ens = Ensemble(sim_of(name), R)
            res = None
            for it in range(2):
                res = ens.simulate(N, sample_size=10 ** 12, record_events=True, seeds=np.arange(R) + 77 + it * R, kernel=kernel)
            print("%-10s R=%-7d N=%-7d %-5s %9.2f ms  %.3e events/s" % (name, R, N, kernel, res.kernel_ms, res.total_events / (res.kernel_ms * 1e-3)), flush=True)
            ens.close()
        except Exception as ex:
            print(name, R, kernel, "ERR", repr(ex)[:150])
