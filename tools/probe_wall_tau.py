import sys, os, time, io, contextlib
sys.path.insert(0, "/root/repo")
os.environ["VGX_TIMING"] = "1"
from vgsim_amd import Simulator
with contextlib.redirect_stdout(io.StringIO()):
    s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.4)
s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
for n in (500, 500):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(n, sample_size=10 ** 15, method="tau", record_multievents=False)
    t1 = time.perf_counter()
    print("simulate(%d, tau): wall %.2f s, kernel %.1f ms" % (n, t1 - t0, s.simulation._engine.last_kernel_ms), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    s.simulate(500, sample_size=10 ** 15, method="tau", record_multievents=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
