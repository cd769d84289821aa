"""One Table-3 trajectory on the latency kernel (for rocprofv3 passes): python3 tools/run_solo_one.py K M events [kernel]"""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd.ensemble import Ensemble
K, M, n = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
kernel = sys.argv[4] if len(sys.argv) > 4 else "solo"
R = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if K == 0:
    import contextlib, io
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        sim = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
    sim.set_transmission_rate(4.0); sim.set_recovery_rate(1.5); sim.set_sampling_rate(0.3)
else:
    sim = bench.make_table3(K, M)
ens = Ensemble(sim, R)
res = ens.simulate(n, sample_size=10 ** 12, record_events=True, seeds=2023 + np.arange(R, dtype=np.int64), kernel=kernel)
print(json.dumps({"events": int(res.total_events), "iterations": int(res.loop_iterations.sum()), "kernel_ms": res.kernel_ms,
                  "events_per_s": res.total_events / (res.kernel_ms * 1e-3)}))
