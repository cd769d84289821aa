"""Development aid: where the wall time of a Simulator.simulate() call goes besides the kernel (per phase, averaged).
python tools/probe_call_profile.py [events]"""
import os, sys, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd import _capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sim = bench.make_table3(2, 0.001, 2023)
with contextlib.redirect_stdout(io.StringIO()):
    sim.simulate(n, sample_size=10 ** 12)
m = sim.simulation
eng = m._engine
T = {}
def lap(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
calls = 20
for _ in range(calls):
    t = time.perf_counter(); m.events.CreateEvents(n); lap("CreateEvents", t)
    t = time.perf_counter(); eng.set_params(m); lap("set_params", t)
    t = time.perf_counter(); eng.set_state(m); lap("set_state", t)
    t = time.perf_counter(); eng.set_seeds([m.user_seed]); lap("set_seeds", t)
    t = time.perf_counter(); eng._check(eng.lib.vgx_simulate_direct(eng.handle, n, 10 ** 12, -1.0, 200, None)); lap("vgx_simulate_direct", t)
    T["kernel"] = T.get("kernel", 0.0) + eng.last_kernel_ms * 1e-3
    t = time.perf_counter(); eng.get_state(m, 0); lap("get_state", t)
    c = eng.counters(0)
    t = time.perf_counter(); eng.fetch_events(m.events, 0, c.ev_first_new, c.ev_ptr - c.ev_first_new); lap("fetch_events", t)
    m.events.ptr = c.ev_ptr
for k, v in T.items():
    print("%-22s %8.3f ms per call" % (k, 1e3 * v / calls))
