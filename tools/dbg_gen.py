import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, helpers
from oracle import oracle
from vgsim_amd import _capi
oracle.build()
for name in ("g9_short", "stress_h64"):
    hip = helpers.run_case_hip(name); ref = helpers.run_case_oracle(oracle, name, log_mode=oracle.LOG_PORTABLE)
    c = hip.simulation._engine.last_counters
    st = oracle.get_state(ref.simulation)
    print(name, "kernel last_att", c.reserved[1], "loops", c.reserved[2], "total loops", c.loop_iterations, "restarts", c.restarts,
          "oracle good_attempt", ref.simulation.good_attempt, "oracle iterations_done", st.iterations_done)
    out = (C.c_uint64 * 4)()
    _capi.load_library().vgx_rng_position(ref.simulation.user_seed, int(c.reserved[1]), 2 * int(c.reserved[2]), C.byref(out))
    print("  product pos", [hex(int(x)) for x in out]); print("  oracle  pos", [hex(int(x)) for x in st.rng_final])
    for gs in (None, 4711):
        h2 = helpers.run_case_hip(name); r2 = helpers.run_case_oracle(oracle, name, log_mode=oracle.LOG_PORTABLE)
        with helpers.quiet(): h2.genealogy(gs)
        w = oracle.run_genealogy(r2.simulation, gs)
        print("  seed", gs, "tree equal", np.array_equal(h2.simulation.tree, w["tree"]), "times equal", np.array_equal(h2.simulation.times, w["times"]))
