"""Development aid: wall time of repeated short Simulator.simulate() calls (the per-call cost around the kernel).
python tools/probe_call_overhead.py"""
import os, sys, time, contextlib, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for K, n in ((2, 1000), (2, 100000), (10, 1000)):
    sim = bench.make_table3(K, 0.001, 2023)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.simulate(1000, sample_size=10 ** 12)
    t = time.perf_counter()
    calls = 50
    kms = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(calls):
            sim.simulate(n, sample_size=10 ** 12)
            kms += sim.simulation._engine.last_kernel_ms
    w = (time.perf_counter() - t) / calls
    print("Table 3 K=%d, %d events per call: %.2f ms wall per call, %.2f ms of it in the kernel" % (K, n, 1e3 * w, kms / calls), flush=True)
