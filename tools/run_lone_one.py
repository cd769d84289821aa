"""One BASELINE config-3 trajectory (or R of them) on a chosen direct kernel, for rocprofv3 passes:
python3 tools/run_lone_one.py events [kernel] [replicates] [traj_points]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
n = int(sys.argv[1])
kernel = sys.argv[2] if len(sys.argv) > 2 else "lone"
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1
T = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ens = Ensemble(bench.make_simulator(2020), R)
res = ens.simulate(n, sample_size=10 ** 12, record_events=True, seeds=2020 + np.arange(R, dtype=np.int64), kernel=kernel,
                   traj_points=T, traj_window=(0.0, 12.0))
print(json.dumps({"events": int(res.total_events), "iterations": int(res.loop_iterations.sum()), "kernel_ms": res.kernel_ms,
                  "events_per_s": res.total_events / (res.kernel_ms * 1e-3), "kernel": ens.engine.last_kernel}))
