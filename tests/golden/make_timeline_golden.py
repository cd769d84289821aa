#!/usr/bin/env python3
"""Golden vectors for the log replays get_data_infectious / get_data_susceptible (reference
src/_BirthDeath.pyx:1967-2045), recorded from the REFERENCE itself (development container only; usage as
make_golden.py).  Inputs are the cases of tests/models.py plus the (population, haplotype|group, step_num) queries
listed here; outputs the arrays the reference returned."""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import models  # noqa: E402

import VGsim  # noqa: E402

QUERIES = {"g9_short": dict(inf=[(0, 0), (1, 12), (2, 3)], sus=[(0, 0), (1, 1), (2, 2)], steps=40),
           "stress_h64": dict(inf=[(0, 0), (1, 37)], sus=[(0, 0), (1, 1)], steps=25),
           "tau_b": dict(inf=[(0, 0), (1, 5), (2, 15)], sus=[(0, 0), (2, 1)], steps=30)}

for name, q in QUERIES.items():
    with contextlib.redirect_stdout(io.StringIO()):
        sim, phases = models.build(VGsim.Simulator, name)
        for setup, kw in phases:
            setup(sim)
            sim.simulate(**kw)
    m = sim.simulation
    out = {}
    for k, (p, h) in enumerate(q["inf"]):
        data, sample, tp, ld = m.get_data_infectious(p, h, q["steps"])
        out["inf%d_data" % k], out["inf%d_sample" % k] = np.asarray(data), np.asarray(sample)
        out["inf%d_tp" % k] = np.asarray(tp, dtype=float)
        out["inf%d_ld" % k] = np.asarray([[float(a), float(b)] for a, b in ld], dtype=float).reshape(-1, 2)
    for k, (p, s) in enumerate(q["sus"]):
        data, tp, ld = m.get_data_susceptible(p, s, q["steps"])
        out["sus%d_data" % k] = np.asarray(data)
    np.savez_compressed(os.path.join(HERE, "timeline_%s.npz" % name), meta=json.dumps(dict(case=name, **q)), **out)
    print(name, "ok")
