"""Diagnostic: per-phase shader cycles of the general row kernel (VGX_LIBRARY=vgsim_amd/libvgx_prof.so, `make -C vgsim_amd/csrc prof`) on the
Table-3 model: python tools/profile_quadg.py K M replicates events"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
K, M, R, N = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ens = Ensemble(bench.make_table3(K, M), R)
res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=0, kernel="quadg")
tot = np.zeros(16)
eng = ens.engine
step = 4 * max(1, R // 4 // 64)
for rep in range(0, R, step):
    out = np.zeros(16, dtype=np.int64)
    eng.lib.vgx_get_profile(eng.handle, rep, out.ctypes.data_as(C.POINTER(C.c_int64)))
    tot += out
names = ["after pass -> loop top", "front", "rng + time", "population choice", "migration: populations", "cold record load", "immunity transition",
         "haplotype choice", "class + apply + mutation", "migration: haplotype, thinning", "list operations", "add event", "lockdown check + eff",
         "BirthRate segments", "infectPopRate + sums", "-"]
iters = res.loop_iterations[::step].sum()
print("%.3e ev/s, %.1f ms; cycles per wave-iteration: %.0f" % (res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms, tot.sum() / iters))
for n, v in zip(names, tot):
    if v: print("%-32s %6.1f %%  %8.0f cycles/iteration" % (n, 100 * v / tot.sum(), v / iters))
