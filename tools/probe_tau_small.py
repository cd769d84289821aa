"""Development aid: tau-leaping steps per second on small models (launch-bound regime), GPU engine vs the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import helpers
from vgsim_amd import Simulator

def model(sites, pops, size):
    with helpers.quiet():
        s = Simulator(number_of_sites=sites, populations_number=pops, seed=7)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
    if pops > 1:
        s.set_total_migration_probability(0.02)
    s.set_population_size(size)
    return s

for sites, pops, size in ((2, 3, 10 ** 6), (4, 5, 10 ** 6), (6, 8, 10 ** 7)):
    s = model(sites, pops, size)
    with helpers.quiet():
        s.simulate(2000, sample_size=10 ** 12)            # seed the epidemic with the direct method
        t0 = time.time(); s.simulate(300, sample_size=10 ** 12, method="tau", record_multievents=False); t1 = time.time()   # warm-up (allocations)
        t0 = time.time(); s.simulate(2000, sample_size=10 ** 12, method="tau", record_multievents=False); t1 = time.time()
    m = s.simulation
    line = "sites %d pops %d: GPU %.0f steps/s wall, %.0f steps/s device (%d events in the log, infected %d)" % (
        sites, pops, 2000 / (t1 - t0), 2000 / (m._engine.last_kernel_ms * 1e-3), m.events.ptr, m.globalInfectious)
    try:
        from oracle import oracle
        oracle.build()
        o = model(sites, pops, size)
        om = o.simulation
        oracle.run_direct(om, 2000, 10 ** 12, -1, 200)
        t2 = time.time(); oracle.run_tau(om, 200, 10 ** 12, -1, 200); t3 = time.time()
        line += "; CPU oracle %.0f steps/s" % (200 / (t3 - t2))
    except Exception as e:   # the oracle is test infrastructure: optional here
        line += "; (oracle: %r)" % (e,)
    print(line, flush=True)
