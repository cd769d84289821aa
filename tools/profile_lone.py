"""Diagnostic: per-phase shader cycles of the latency kernel for large haplotype spaces (VGX_LIBRARY=vgsim_amd/libvgx_prof.so,
`make -C vgsim_amd/csrc prof`):  python tools/profile_lone.py [events] [replicates] [model]   (model: c3 = BASELINE config 3 (default),
c3g = bench.make_general_c3, tK = the Table-3 model with K demes, e.g. t10)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
MODEL = sys.argv[3] if len(sys.argv) > 3 else "c3"
sim = bench.make_simulator(2020) if MODEL == "c3" else bench.make_general_c3() if MODEL == "c3g" else bench.make_table3(int(MODEL[1:]), 0.001)
ens = Ensemble(sim, R)
res = None
for it in range(2):
    res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=0, kernel="lone", seeds=2020 + np.arange(R, dtype=np.int64))
eng = ens.engine
out = np.zeros(16, dtype=np.int64)
eng.lib.vgx_get_profile(eng.handle, 0, out.ctypes.data_as(C.POINTER(C.c_int64)))
names = ["SampleTime, uniforms, loop counters", "population choice", "haplotype + class choice", "event: counts, list operations", "BirthRate",
         "migration rates", "list refresh (prefix sums)", "popRate scan", "log + counters", "after event: flush, extinction, loop control",
         "general form: choices + event + list operations", "general form: UpdateRates + log"]
iters = float(res.loop_iterations[0])
print("%.3e ev/s, %.1f ms; stamped cycles per iteration: %.0f (stamps cost ~40 cycles each)" % (res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms, out.sum() / iters))
for n, v in zip(names, out):
    if v: print("%-46s %6.1f %%  %8.0f cycles/iteration" % (n, 100 * v / out.sum(), v / iters))
