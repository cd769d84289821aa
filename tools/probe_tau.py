"""Development aid: tau-leap step timing at large, densely occupied shapes (config 3 / config 4 of BASELINE.json)."""
import contextlib, io, sys, os, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgsim_amd import Simulator, _capi


def run(sites, P, steps, per_cell=3, R=1):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=P, seed=2020)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    m = s.simulation
    H = m.hapNum
    m.infectious[:] = per_cell                      # dense ("spread") occupancy, set directly on the host arrays
    m.susceptible[:, 0] -= per_cell * H
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=R)
    m.events.CreateEvents(steps); m.events.CreateEvents(steps)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.arange(2020, 2020 + R))
    o = _capi.VgxRunOpts(); o.record_events = 0
    t = time.time()
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, steps, 10 ** 15, -1.0, 1, C.byref(o)))
    wall = time.time() - t
    c = eng.counters(0)
    drawn = c.reserved[0]
    ms = eng.last_kernel_ms
    fused_min = 16.0 * P * H  # bytes/step: read + write infectious once
    print("tau sites=%d H=%d P=%d R=%d: %d steps, device %.1f ms (%.2f ms/step), wall %.1f s, events drawn %.3g -> %.3g ev/s; "
          "fused-minimum traffic %.2f GB/step -> %.0f GB/s equivalent; devMB=%.0f" % (
              sites, H, P, R, c.loop_iterations, ms, ms / max(c.loop_iterations, 1), wall, drawn, drawn / (ms * 1e-3),
              fused_min / 1e9, fused_min * c.loop_iterations / (ms * 1e-3) / 1e9, eng.device_bytes / 1e6), flush=True)
    eng.close()


if __name__ == "__main__":
    run(4, 8, 20)
    run(8, 64, 5)
    if len(sys.argv) > 1:
        run(10, 256, 2)
