"""Diagnostic: per-phase shader cycles of the quad kernel (libvgx built with -DVGX_PROFILE -> vgsim_amd/libvgx_prof.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
R, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ens = Ensemble(bench.make_simulator(2020), R)
res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=1001, traj_window=(0.0, 12.0), kernel="quad")
tot = np.zeros(16)
eng = ens.engine
for rep in range(0, R, 4 * 37):
    out = np.zeros(16, dtype=np.int64)
    eng.lib.vgx_get_profile(eng.handle, rep, out.ctypes.data_as(C.POINTER(C.c_int64)))
    tot += out
names = ["loop top", "front", "rng", "time+traj", "pop select", "hap select", "class+apply", "mutation", "migration", "list ops", "add event",
         "birth rate", "row sum", "cum scan", "mig sum", "tail"]
iters = res.loop_iterations[::4 * 37].sum()
print("%.3e ev/s, %.1f ms; cycles per wave-iteration: %.0f" % (res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms, tot.sum() / iters))
long_iters = tot[15]          # slot 15 counts iterations, not cycles: the haplotype choice took the long-list form
tot[15] = 0
for n, v in zip(names, tot):
    print("%-12s %6.1f %%  %8.0f cycles/iteration" % (n, 100 * v / tot.sum(), v / iters))
print("iterations whose haplotype choice streamed a list of more than 64 entries: %.3f %%" % (100.0 * long_iters / iters))
