import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
os.environ["VGX_TIMING"] = "1"
import numpy as np
import test_hip_tau as T, helpers
for fill, sites, P in ((T._fill_sparse, 9, 2), (T._fill_one_region, 9, 2), (T._fill_small, 9, 3)):
    s = T._filled(sites, P, 1, 900 + sites, fill, True)
    with helpers.quiet():
        s.simulate(8, sample_size=10 ** 12, method="tau", record_multievents=False)
    print(fill.__name__, "done", flush=True)
