"""Host-side phases of vgx_simulate_tau at BASELINE config 4 (VGX_TIMING=1 prints them on stderr): python tools/probe_tau_wall.py [steps]"""
import contextlib, io, os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VGX_TIMING"] = "1"
from vgsim_amd import Simulator, _capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with contextlib.redirect_stdout(io.StringIO()):
    s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
m = s.simulation
m.infectious[:] = 3
m.susceptible[:, 0] -= 3 * m.hapNum
eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
m.events.CreateEvents(steps); m.events.ptr = 1; m.events.CreateEvents(steps)
t = time.perf_counter(); eng.set_params(m); print("set_params %.1f ms" % (1e3 * (time.perf_counter() - t)))
t = time.perf_counter(); eng.set_state(m); print("set_state %.1f ms" % (1e3 * (time.perf_counter() - t)))
eng.set_seeds(np.array([2020], dtype=np.int64))
o = _capi.VgxRunOpts(); o.record_events = 0
for k in range(3):
    eng.set_state(m)
    t = time.perf_counter(); eng.stage_tau(); print("stage_tau %.1f ms" % (1e3 * (time.perf_counter() - t)))
    t = time.perf_counter()
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, steps, 10 ** 15, -1.0, 1, C.byref(o)))
    w = time.perf_counter() - t
    c = eng.counters(0)
    print("call %d: wall %.1f ms, device %.1f ms, %d steps -> wall %.2f ms/step, device %.2f ms/step" % (
        k, 1e3 * w, eng.last_kernel_ms, c.loop_iterations, 1e3 * w / max(c.loop_iterations, 1), eng.last_kernel_ms / max(c.loop_iterations, 1)), flush=True)
    m.events.CreateEvents(steps)
