"""Development aid: a large epidemic on 65536 haplotypes x 16 populations (10^7 hosts each), tau steps from an index-case
warm-up; for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import helpers
from vgsim_amd import Simulator
with helpers.quiet():
    s = Simulator(number_of_sites=8, populations_number=16, seed=7)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
s.set_total_migration_probability(0.02); s.set_population_size(10 ** 7)
with helpers.quiet():
    s.simulate(2000, sample_size=10 ** 12)
    s.simulate(1500, sample_size=10 ** 12, method="tau", record_multievents=False)
    t0 = time.time(); s.simulate(500, sample_size=10 ** 12, method="tau", record_multievents=False); t1 = time.time()
m = s.simulation
print("steps/s", 500 / (t1 - t0), "infected", m.globalInfectious, "occupied", int((m.infectious > 0).sum()), "max", int(m.infectious.max()))
