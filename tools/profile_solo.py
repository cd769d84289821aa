"""Diagnostic: per-phase shader cycles of the single-trajectory kernel (VGX_LIBRARY=vgsim_amd/libvgx_prof.so, `make -C vgsim_amd/csrc prof`):
python tools/profile_solo.py K M events   (K = 0: BASELINE config 2)"""
import ctypes as C, os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
K, M, N = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
if K == 0:
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        sim = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
    sim.set_transmission_rate(4.0); sim.set_recovery_rate(1.5); sim.set_sampling_rate(0.3)
else:
    sim = bench.make_table3(K, M)
ens = Ensemble(sim, 1)
res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=0, kernel="solo", seeds=np.array([2023], dtype=np.int64))
eng = ens.engine
out = np.zeros(16, dtype=np.int64)
eng.lib.vgx_get_profile(eng.handle, 0, out.ctypes.data_as(C.POINTER(C.c_int64)))
names = ["loop control + refill", "uniform fetch (+ clock)", "population choice", "rescale, immune test", "row switch", "haplotype + type choice",
         "group choice + apply", "-> tail", "BirthRate", "row refresh + scan", "immune sum", "popRate scan", "migration rates", "log + counters",
         "after event: flush, extinction, lockdown", "-"]
iters = float(res.loop_iterations.sum())
print("%.3e ev/s, %.1f ms; stamped cycles per iteration: %.0f (stamps cost ~40 cycles each)" % (res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms, out.sum() / iters))
for n, v in zip(names, out):
    if v: print("%-42s %6.1f %%  %8.0f cycles/iteration" % (n, 100 * v / out.sum(), v / iters))
