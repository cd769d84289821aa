import os, sys, contextlib, io
sys.path.insert(0, "/root/repo")
import numpy as np
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
for warm in (2000, 200000, 3000000):
    row = {}
    for env in ("1", "0"):
        os.environ["VGX_TAU_STEP_KERNELS"] = env
        with contextlib.redirect_stdout(io.StringIO()):
            s = Simulator(number_of_sites=4, populations_number=5, seed=7)
        s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
        s.set_total_migration_probability(0.002); s.set_population_size(10 ** 8)
        with contextlib.redirect_stdout(io.StringIO()):
            s.simulate(warm, sample_size=10 ** 12)
        ens = Ensemble(s, 1)
        for it in range(2):
            res = ens.simulate_tau(300, sample_size=10 ** 15, seeds=np.array([7 + it], dtype=np.int64))
        row["step kernels" if env == "1" else "on-device loop"] = "%.3g" % (float(res.loop_iterations.sum()) / (res.kernel_ms * 1e-3))
        ens.close()
    print("256 x 5 after %d direct events (%d infected): %s" % (warm, s.simulation.globalInfectious, row), flush=True)
