"""Development aid: tau invariants of the seeded random models of tests/test_hip_fuzz.py, per mode of the tries
(vgx_run_opts.reserved[1]): python tools/dbg_tau_fuzz.py [seed ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import helpers
import test_hip_fuzz as tf
from vgsim_amd import _capi

for seed in [int(x) for x in sys.argv[1:]] or [0]:
    for mode in (0, 2, 1):
        sim, n = tf.build(seed)
        m = sim.simulation
        with helpers.quiet():
            sim.simulate(min(n, 600), sample_size=10 ** 9)
        if m.globalInfectious == 0:
            print(seed, "extinct"); break
        o = _capi.VgxRunOpts(); o.record_events = 1; o.reserved[1] = mode
        m.events.CreateEvents(25); m.events.CreateEvents(25); m.CheckSizes()
        e = m._get_engine()
        e.simulate_tau(m, 25, 10 ** 12, -1.0, 200, o)
        c = e.last_counters
        d = m.susceptible.sum(axis=1) + m.infectious.sum(axis=1) - m.sizes
        print("seed", seed, "mode", mode, "H", m.hapNum, "P", m.popNum, "S", m.susNum, "deficit", d.tolist(), "drawn", int(c.reserved[0]), "ptr", int(c.ev_ptr),
              "counters", [int(getattr(m, k)) for k in m.COUNTERS])
