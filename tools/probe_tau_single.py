"""Development aid: tau trajectories of small and mid-size models on the step kernels and on the on-device loop (steps/s of device
time, all replicates together), over model sizes and ensemble sizes.  python tools/probe_tau_single.py"""
import os, sys, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
for sites, pops in ((3, 4), (4, 5), (4, 8), (4, 16), (5, 8)):
  for R in (1, 8, 32, 128, 512):
    row = {}
    for env in ("1", "0"):
        os.environ["VGX_TAU_STEP_KERNELS"] = env
        with contextlib.redirect_stdout(io.StringIO()):
            s = Simulator(number_of_sites=sites, populations_number=pops, seed=7)
        s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
        s.set_total_migration_probability(0.002); s.set_population_size(10 ** 6)
        with contextlib.redirect_stdout(io.StringIO()):
            s.simulate(2000, sample_size=10 ** 12)
        try:
            ens = Ensemble(s, R)
            for it in range(2):
                res = ens.simulate_tau(200, sample_size=10 ** 15, seeds=7 + it * R + np.arange(R, dtype=np.int64))
            row["step kernels" if env == "1" else "on-device loop"] = float(res.loop_iterations.sum()) / (res.kernel_ms * 1e-3)
            ens.close()
        except Exception as ex:
            row["step kernels" if env == "1" else "on-device loop"] = str(ex)[:50]
    m = s.simulation
    print("%d haplotypes x %d populations (%d cells), R=%d: %s" % (m.hapNum, m.popNum, m.hapNum * m.popNum, R, {k: ("%.3g" % v if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
