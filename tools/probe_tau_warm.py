"""config 4 started by SURVEY.md 8(d)'s recipe instead of a uniform fill: index case, high-mutation tau warm-up, then the timed steps.
python tools/probe_tau_warm.py warm_steps [timed_steps]"""
import contextlib, io, os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgsim_amd import Simulator, _capi
W = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
with contextlib.redirect_stdout(io.StringIO()):
    s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.4)
s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
m = s.simulation
t = time.time()
done = 0
while done < W:
    n = min(2000, W - done)
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(n, sample_size=10 ** 15, method="tau", record_multievents=False)
    done += n
    occ = int((m.infectious != 0).sum())
    print("warm-up %d steps: t=%.3f infected=%d occupied=%d (%.2f%%) max=%d  [%.1f s]" % (done, m.currentTime, m.globalInfectious, occ, 100.0 * occ / m.infectious.size,
          int(m.infectious.max()), time.time() - t), flush=True)
s.set_mutation_rate(0.01)
eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
m.events.CreateEvents(N)
eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([2020], dtype=np.int64)); eng.stage_tau()
o = _capi.VgxRunOpts(); o.record_events = 0
t = time.perf_counter()
eng._check(eng.lib.vgx_simulate_tau(eng.handle, N, 10 ** 15, -1.0, 1, C.byref(o)))
w = time.perf_counter() - t
c = eng.counters(0)
print("timed: %d steps, device %.2f ms/step, wall %.2f ms/step, %.3g events drawn -> %.3g ev/s, tries skipped %d" % (
    c.loop_iterations, eng.last_kernel_ms / max(c.loop_iterations, 1), 1e3 * w / max(c.loop_iterations, 1), c.reserved[0], c.reserved[0] / (eng.last_kernel_ms * 1e-3), c.reserved[3]))
