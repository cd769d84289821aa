"""Development aid: the automatic kernel choice of vgx_simulate_direct against every kernel that takes the model, over model shapes and
ensemble sizes (events/s of device time; '!' marks a forced kernel that beats the automatic choice by more than 25 %).
python tools/probe_dispatch.py [events]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import models, helpers, bench
from vgsim_amd import Simulator
from vgsim_amd.ensemble import Ensemble
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8000


def case(name):
    ctor, phases = models.CASES[name] if name in models.CASES else models.ORACLE_ONLY_CASES[name]
    def make():
        with helpers.quiet():
            s = Simulator(**ctor)
        phases[0][0](s)
        return s
    return make


SHAPES = [("table3 K=2", lambda: bench.make_table3(2, 0.001, 2023)), ("table3 K=10", lambda: bench.make_table3(10, 0.001, 2023)),
          ("table3 K=100", lambda: bench.make_table3(100, 0.001, 2023)), ("g9_short", case("g9_short")), ("p70", case("p70")),
          ("stress_h64", case("stress_h64")), ("stress_h256", case("stress_h256")), ("recomb_a", case("recomb_a")),
          ("lockdown_restart", case("lockdown_restart")), ("example", case("example"))]
for label, make in SHAPES:
    for R in (1, 64, 512, 4096, 16384):
        if "K=100" in label and R > 4096:
            continue
        row = {}
        for kernel in ("auto", "solo", "quadg", "quad", "wave", "lane"):
            try:
                sim = make()
                ens = Ensemble(sim, R)
                res = ens.simulate(N, sample_size=10 ** 12, kernel=kernel)
                row[kernel] = (res.total_events / max(res.kernel_ms, 1e-6) * 1e3, ens.engine.last_kernel)
                ens.close()
            except Exception:
                pass
        base = row.get("auto", (0.0, "?"))
        txt = " ".join("%s %.3g%s" % (k, v[0], "!" if k != "auto" and v[0] > 1.25 * base[0] else "") for k, v in row.items() if k != "auto")
        print("%-18s R=%-6d auto -> %-6s %.3g | %s" % (label, R, base[1], base[0], txt), flush=True)
