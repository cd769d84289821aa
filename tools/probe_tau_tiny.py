"""Development aid: the smallest case of tools/probe_tau_small.py alone (16 haplotypes x 3 populations), for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import helpers
from vgsim_amd import Simulator
with helpers.quiet():
    s = Simulator(number_of_sites=2, populations_number=3, seed=7)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
s.set_total_migration_probability(0.02); s.set_population_size(10 ** 7)
with helpers.quiet():
    s.simulate(2000, sample_size=10 ** 12)
    s.simulate(300, sample_size=10 ** 12, method="tau", record_multievents=False)
    t0 = time.time(); s.simulate(2000, sample_size=10 ** 12, method="tau", record_multievents=False); t1 = time.time()
print("steps/s", 2000 / (t1 - t0), "kernel ms", s.simulation._engine.last_kernel_ms)
