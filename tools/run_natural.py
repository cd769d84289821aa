"""Runs the headline workload once per kernel choice (for rocprofv3): python tools/run_natural.py quad 8192 20000 [spread]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
kernel, R, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spread = len(sys.argv) > 4 and sys.argv[4] == "spread"
sim = bench.make_simulator(2020)
if spread:
    m = sim.simulation
    rng = np.random.default_rng(2020)
    for pn in range(bench.POPS):
        haps = rng.choice(m.hapNum, size=4096, replace=False)
        m.infectious[pn, haps] = rng.integers(1, 4, size=4096)
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
ens = Ensemble(sim, R)
res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=0 if spread else 1001, traj_window=(0.0, 12.0), kernel=kernel)
print("events", res.total_events, "iterations", int(res.loop_iterations.sum()), "ms", res.kernel_ms, "ev/s %.3e" % (res.total_events / res.kernel_ms * 1e3))
