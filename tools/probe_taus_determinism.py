import os, sys, contextlib, io
sys.path.insert(0, "/root/repo")
import numpy as np
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
def run(tt, sites, pops):
    os.environ["VGX_TAUS_THREADS"] = str(tt)
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=pops, number_of_susceptible_groups=2, seed=7)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
    s.set_total_migration_probability(0.002); s.set_population_size(10 ** 6)
    s.set_susceptibility_type(1); s.set_susceptibility(0.3, susceptibility_type=1); s.set_immunity_transition(0.02, source=1, target=0)
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(2000, sample_size=10 ** 12)
    ens = Ensemble(s, 3)
    res = ens.simulate_tau(300, sample_size=10 ** 15, seeds=np.array([5, 6, 7], dtype=np.int64))
    out = []
    for r in range(3):
        st = ens.replicate_state(r)
        out.append((st.infectious.copy(), st.susceptible.copy(), st.currentTime))
    ens.close()
    return out
for sites, pops in ((2, 3), (3, 4), (4, 5)):
    a, b, c = run(512, sites, pops), run(64, sites, pops), run(256, sites, pops)
    same = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] for x, y in zip(a, b)) and \
           all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] for x, y in zip(a, c))
    print("%dx%d: 64 / 256 / 512 threads give the same runs: %s (t = %.6f)" % (4 ** sites, pops, same, a[0][2]), flush=True)
