"""Spread-occupancy probe: python tools/probe_spread.py R:events:mode:lo:hi[:occupied] ...
Every population starts with `occupied` distinct haplotypes with counts in [lo, hi)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble

for spec in sys.argv[1:]:
    f = spec.split(":")
    R, events, mode, lo, hi = int(f[0]), int(f[1]), f[2], int(f[3]), int(f[4])
    occupied = int(f[5]) if len(f) > 5 else 4096
    sim = bench.make_simulator(2020)
    m = sim.simulation
    rng = np.random.default_rng(2020)
    for pn in range(bench.POPS):
        haps = rng.choice(m.hapNum, size=occupied, replace=False)
        m.infectious[pn, haps] = rng.integers(lo, hi, size=occupied)
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
    ens = Ensemble(sim, R, device=0)
    seeds = 5000 + np.arange(R, dtype=np.int64)
    for it in range(2):
        res = ens.simulate(events, sample_size=10 ** 12, record_events=True, traj_points=0, seeds=seeds + it * R, mode=mode)
    print(spec, "%.3e events/s  %.2f ms" % (res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms), flush=True)
    ens.close()
