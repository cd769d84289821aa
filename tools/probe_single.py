import contextlib, io, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import models
from vgsim_amd import Simulator
for name in ("g9", "g1", "g5", "stress_h64"):
    for kernel in ("wave", "lane"):
        with contextlib.redirect_stdout(io.StringIO()):
            sim, phases = models.build(Simulator, name)
            phases[0][0](sim)
            sim.simulate(100000, kernel=kernel)
        m = sim.simulation
        print(name, kernel, "events", m.events.ptr, "kernel ms %.1f" % m._engine.last_kernel_ms, "-> %.3g ev/s" % (m.events.ptr / (m._engine.last_kernel_ms * 1e-3)), flush=True)
