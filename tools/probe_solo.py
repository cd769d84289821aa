"""One trajectory at a time on the latency kernel (vgx_solo.hip) beside the row kernels: events/s of device time."""
import contextlib, io, json, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble

def one(sim, n, kernel, R=1):
    ens = Ensemble(sim, R)
    res = None
    for it in range(2):
        res = ens.simulate(n, sample_size=10 ** 12, record_events=True, seeds=2023 + it * R + np.arange(R, dtype=np.int64), kernel=kernel)
    v = res.total_events / (res.kernel_ms * 1e-3)
    ens.close()
    return v

out = {}
with contextlib.redirect_stdout(io.StringIO()):
    c2 = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
c2.set_transmission_rate(4.0); c2.set_recovery_rate(1.5); c2.set_sampling_rate(0.3)
kernels = sys.argv[1].split(",") if len(sys.argv) > 1 else ["solo", "quadg"]
for k in kernels:
    out["config2/" + k] = one(c2, 200000, k if k != "quadg" else "quad")
    for K, M in ((2, 0.001), (10, 0.001), (100, 0.001), (10, 0.1)):
        out["table3 K=%d M=%g/%s" % (K, M, k)] = one(bench.make_table3(K, M), 200000 if K < 100 else 60000, k)
    print(json.dumps(out), flush=True)
for R in (64, 1024, 4096, 16384):
    for k in kernels:
        out["table3 K=10 R=%d/%s" % (R, k)] = one(bench.make_table3(10, 0.001), 20000, k, R)
    print(json.dumps(out), flush=True)
