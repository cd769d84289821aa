"""Development probe: events/s of the quad and wave kernels on BASELINE config 3 (natural and spread occupancy)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble

def natural(kernel, R, N, T=1001):
    ens = Ensemble(bench.make_simulator(2020), R)
    res = None
    for it in range(2):
        res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=T, traj_window=(0.0, 12.0),
                           seeds=2020 + it * R + np.arange(R, dtype=np.int64), kernel=kernel)
    ens.close()
    return res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms

def spread(kernel, R, N, occupied=4096):
    sim = bench.make_simulator(2020)
    m = sim.simulation
    rng = np.random.default_rng(2020)
    for pn in range(bench.POPS):
        haps = rng.choice(m.hapNum, size=occupied, replace=False)
        m.infectious[pn, haps] = rng.integers(1, 4, size=occupied)
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
    ens = Ensemble(sim, R)
    res = None
    for it in range(2):
        res = ens.simulate(N, sample_size=10 ** 12, record_events=True, seeds=5000 + it * R + np.arange(R, dtype=np.int64), kernel=kernel)
    ens.close()
    return res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "natural"):
        for kernel, R, N in (("wave", 4096, 50000), ("quad", 4096, 50000), ("quad", 8192, 50000), ("quad", 16384, 25000), ("quad", 32768, 12500)):
            v, ms = natural(kernel, R, N)
            print("natural %-5s R=%-6d N=%-6d  %.3e ev/s  %.1f ms" % (kernel, R, N, v, ms), flush=True)
    if what in ("all", "spread"):
        for kernel, R, N in (("wave", 4096, 2500), ("quad", 4096, 2500), ("quad", 8192, 2500), ("quad", 16384, 1500)):
            v, ms = spread(kernel, R, N)
            print("spread  %-5s R=%-6d N=%-6d  %.3e ev/s  %.1f ms  (%.2f TB/s of 8 B/entry count stream)" % (kernel, R, N, v, ms, v * 12 * 4096 / 1e12), flush=True)
