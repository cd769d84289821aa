import sys, os
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from vgsim_amd.ensemble import Ensemble
cases = [a.split(":") for a in sys.argv[1:]] or [("quad", 16384, 100000), ("quadg", 16384, 100000)]
for case in cases:
    kernel, R, N = case[0], int(case[1]), int(case[2])
    mode = case[3] if len(case) > 3 else "exact"
    ens = Ensemble(bench.make_simulator(2020), R)
    res = None
    for it in range(2):
        res = ens.simulate(N, sample_size=10 ** 12, record_events=True, traj_points=1001, traj_window=(0.0, 12.0),
                           seeds=2020 + it * R + np.arange(R, dtype=np.int64), kernel=kernel, mode=mode)
    st = ens.replicate_state(0)
    print(mode, "%s R=%d N=%d  %.3e ev/s  %.1f ms  nocc %.1f  bytes %.1f GB" % (kernel, R, N, res.total_events / (res.kernel_ms * 1e-3), res.kernel_ms,
          float((st.infectious != 0).sum(axis=1).mean()), ens.engine.device_bytes / 1e9), flush=True)
    ens.close()
