"""Development probe: tries left out by the tau sieve for a dense synthetic state (tools/, not shipped)."""
import contextlib, ctypes as C, io, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgsim_amd import Simulator, _capi


def run(sites, P, N, steps=6, off=0, first=False):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=P, seed=31)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01); s.set_population_size(N)
    m = s.simulation
    m.infectious[:] = 3
    m.susceptible[:, 0] -= 3 * m.hapNum
    if first:
        m.totalInfectious[:] = 3 * m.hapNum
        m.totalSusceptible[:] = m.susceptible.sum(axis=1)
        m.globalInfectious = int(m.totalInfectious.sum())
        m.first_simulation = True
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
    m.events.CreateEvents(steps); m.events.ptr = 1; m.events.CreateEvents(steps)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([31], dtype=np.int64))
    o = _capi.VgxRunOpts(); o.record_events = 0; o.reserved[0] = off
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, steps, 10 ** 15, -1.0, 1, C.byref(o)))
    c = eng.counters(0)
    eng.get_state(m, 0)
    print(sites, P, N, "first" if first else "", "off" if off else "on", "steps", c.loop_iterations, "skipped", c.reserved[3], "drawn", c.reserved[0],
          "launches", eng.lib.vgx_last_kernel_launches(eng.handle), "t", m.currentTime, "ms", eng.last_kernel_ms)
    eng.close()


if __name__ == "__main__":
    run(8, 16, 10 ** 8)
    run(8, 16, 10 ** 8, first=True)
    run(8, 16, 10 ** 8, off=1)
    run(10, 64, 10 ** 7)
