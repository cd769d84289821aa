"""Timing probe (development aid): config-3 shaped direct runs, single trajectory and ensembles."""
import contextlib, io, sys, time, os
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgsim_amd import Simulator
from vgsim_amd import _capi


def c3_model(sites=8, P=64, seed=2020, mut=0.01):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=P, number_of_susceptible_groups=1, seed=seed)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1)
    s.set_mutation_rate(mut)
    if P > 1:
        s.set_total_migration_probability(0.01)
    s.set_population_size(10 ** 7)
    return s


def run(sites, P, R, N, mut=0.01, record=0):
    s = c3_model(sites, P, mut=mut)
    m = s.simulation
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=R)
    m.events.CreateEvents(N)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.arange(2020, 2020 + R))
    o = _capi.VgxRunOpts(); o.record_events = record
    t = time.time()
    rc = eng.lib.vgx_simulate_direct(eng.handle, N, 10 ** 12, -1.0, 200, C.byref(o))
    wall = time.time() - t
    eng._check(rc)
    ev = sum(eng.counters(r).ev_ptr for r in range(R))
    loops = sum(eng.counters(r).loop_iterations for r in range(R))
    ms = eng.last_kernel_ms
    print("sites=%d P=%d R=%d N=%d mut=%g: events=%d loops=%d kernel=%.1f ms wall=%.2f s -> %.3g ev/s (%.2f us/event/replicate) devMB=%.0f" % (
        sites, P, R, N, mut, ev, loops, ms, wall, ev / (ms * 1e-3), ms * 1e3 / max(ev / R, 1), eng.device_bytes / 1e6), flush=True)
    eng.close()


if __name__ == "__main__":
    for args in [(0, 1, 1, 200000), (2, 3, 1, 100000), (8, 64, 1, 20000), (8, 64, 256, 20000), (8, 64, 1024, 20000),
                 (8, 64, 4096, 20000), (0, 1, 4096, 100000)]:
        run(*args)
