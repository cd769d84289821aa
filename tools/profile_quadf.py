"""Diagnostic: per-phase shader cycles of the FAST row kernel (VGX_LIBRARY=vgsim_amd/libvgx_prof.so, `make -C vgsim_amd/csrc prof`):
python tools/profile_quadf.py R events lo hi occupied   (occupied = 0: the headline's index-case start)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
R, N, lo, hi, occ = (int(x) for x in sys.argv[1:6])
mode = sys.argv[6] if len(sys.argv) > 6 else "fast"     # "exact": the stamps of vgx_quad(_long)_kernel
sim = bench.make_simulator(2020)
m = sim.simulation
if occ:
    rng = np.random.default_rng(2020)
    for pn in range(bench.POPS):
        haps = rng.choice(m.hapNum, size=occ, replace=False)
        m.infectious[pn, haps] = rng.integers(lo, hi, size=occ)
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
ens = Ensemble(sim, R)
res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=0, mode=mode)
tot = np.zeros(16)
eng = ens.engine
for rep in range(0, R, 4 * 37):
    out = np.zeros(16, dtype=np.int64)
    eng.lib.vgx_get_profile(eng.handle, rep, out.ctypes.data_as(C.POINTER(C.c_int64)))
    tot += out
names = ["loop top", "-", "front+rng+time", "pop select", "hap select", "class+apply", "mutation+migration", "sync", "lower bound", "tile sums",
         "shift", "add event", "-", "rates+tail", "-", "-"]
if mode == "exact":
    names = ["loop top", "front", "rng", "time+traj", "pop select", "hap select", "class+apply", "mutation", "migration", "list ops", "add event",
             "birth rate", "row sum", "cum scan", "mig sum", "tail"]
iters = res.loop_iterations[::4 * 37].sum()
print("%.3e ev/s, %.1f ms; cycles per wave-iteration: %.0f" % (res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms, tot.sum() / iters))
for n, v in zip(names, tot):
    if v: print("%-20s %6.1f %%  %8.0f cycles/iteration" % (n, 100 * v / tot.sum(), v / iters))
