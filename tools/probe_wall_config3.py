import sys, time, io, contextlib
sys.path.insert(0, "/root/repo")
import bench
s = bench.make_simulator(2020)
with contextlib.redirect_stdout(io.StringIO()):
    s.simulate(1000, sample_size=10**12)
for n in (100000, 100000):
    m = s.simulation
    p0 = m.events.ptr
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(n, sample_size=10**12)
    tw = time.perf_counter() - t0
    print("simulate(%d): wall %.1f ms, kernel %.1f ms (%s), events %d -> %.3e ev/s wall" % (n, 1e3*tw, m._engine.last_kernel_ms, m._engine.last_kernel, m.events.ptr - p0, (m.events.ptr-p0)/tw))
import os
os.environ["VGX_TIMING"] = "1"
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
with contextlib.redirect_stdout(io.StringIO()):
    s.simulate(100000, sample_size=10**12)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
