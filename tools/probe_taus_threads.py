import os, sys, contextlib, io
sys.path.insert(0, "/root/repo")
import numpy as np
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
for sites, pops, reps in ((2, 3, 2048), (3, 4, 512), (4, 5, 512)):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=pops, seed=7)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
    s.set_total_migration_probability(0.002); s.set_population_size(10 ** 6)
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(2000, sample_size=10 ** 12)
    row = {}
    for R in (1, reps):
        ens = Ensemble(s, R)
        for it in range(2):
            res = ens.simulate_tau(1000, sample_size=10 ** 15, seeds=7 + it * R + np.arange(R, dtype=np.int64))
        row["R=%d" % R] = "%.3g" % (float(res.loop_iterations.sum()) / (res.kernel_ms * 1e-3))
        ens.close()
    print(os.environ.get("VGX_LIBRARY", "default"), "%dx%d" % (4 ** sites, pops), row, flush=True)
