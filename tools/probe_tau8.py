"""Config-4 tau leg variants for profiling the byte drift pass: python tools/probe_tau8.py [nomig] [steps]"""
import sys, os, io, contextlib, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from vgsim_amd import Simulator, _capi
nomig = "nomig" in sys.argv
steps = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 6
with contextlib.redirect_stdout(io.StringIO()):
    s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
if not nomig:
    s.set_total_migration_probability(0.01)
s.set_population_size(10 ** 7)
m = s.simulation
m.infectious[:] = 3
m.susceptible[:, 0] -= 3 * m.hapNum
eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
m.events.CreateEvents(steps); m.events.ptr = 1; m.events.CreateEvents(steps)
eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([2020], dtype=np.int64))
o = _capi.VgxRunOpts(); o.record_events = 0
eng._check(eng.lib.vgx_simulate_tau(eng.handle, steps, 10 ** 15, -1.0, 1, C.byref(o)))
c = eng.counters(0)
print("nomig" if nomig else "mig", "steps", int(c.loop_iterations), "ms/step", eng.last_kernel_ms / max(int(c.loop_iterations), 1), "events", int(c.reserved[0]))
