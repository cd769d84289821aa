"""One trajectory at a time on each kernel: python tools/probe_single.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
def c2():
    import contextlib, io
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
    s.set_transmission_rate(4.0); s.set_recovery_rate(1.5); s.set_sampling_rate(0.3)
    return s
for name, mk in (("table3 K=2", lambda: bench.make_table3(2, 0.001)), ("table3 K=10", lambda: bench.make_table3(10, 0.001)),
                 ("config2", c2), ("config3", lambda: bench.make_simulator(2020))):
    for kernel in (sys.argv[1:] or ["wave", "quad", "quadg", "lane"]):
        for R in (1, 4):
            try:
                ens = Ensemble(mk(), R)
                res = None
                for it in range(2):
                    res = ens.simulate(200000, sample_size=10 ** 12, record_events=True, seeds=2020 + it * R + np.arange(R, dtype=np.int64), kernel=kernel)
                print("%-12s %-6s R=%d  %.3e ev/s per replicate %.3e" % (name, kernel, R, res.total_events / (res.kernel_ms * 1e-3), res.total_events / (res.kernel_ms * 1e-3) / R), flush=True)
                ens.close()
            except Exception as ex:
                print("%-12s %-6s R=%d  refused: %s" % (name, kernel, R, str(ex)[:60]), flush=True)
