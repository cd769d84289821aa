"""Diagnostic: phases of the on-device tau step loop (vgx_taus_kernel) on the small models of the tau_small bench leg.
make -C vgsim_amd/csrc prof; VGX_LIBRARY=vgsim_amd/libvgx_prof.so python tools/profile_taus.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers
from vgsim_amd import Simulator, _capi
lib = C.CDLL(_capi.LIB_PATH)
out = (C.c_ulonglong * 12)()
for sites, pops in ((2, 3),):
    with helpers.quiet():
        s = Simulator(number_of_sites=sites, populations_number=pops, seed=7)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
    s.set_total_migration_probability(0.002); s.set_population_size(10 ** 6)
    with helpers.quiet():
        s.simulate(2000, sample_size=10 ** 12)
        s.simulate(300, sample_size=10 ** 12, method="tau", record_multievents=False)
        lib.vgx_taus_get_profile(out, 1)
        s.simulate(1000, sample_size=10 ** 12, method="tau", record_multievents=False)
    lib.vgx_taus_get_profile(out, 0)
    v = np.array(list(out), dtype=np.float64)
    steps, tries = max(v[6], 1), max(v[7], 1)
    print("sites %d pops %d: %.0f steps/s device; %d steps, %d tries; cycles per step %.0f" % (
        sites, pops, 1000 / (s.simulation._engine.last_kernel_ms * 1e-3), steps, tries, v[:6].sum() / steps))
    for n, x in zip(["loop condition + densities", "drift + ChooseTau", "zeroing (per try)", "draws (per try)", "bounds check (per try)", "apply + totals + record"], v[:6]):
        print("  %-28s %5.1f %%  %8.0f cycles per step" % (n, 100 * x / v[:6].sum(), x / steps))
    print("  inside the draws (wavefront 0, all rounds): rates %.0f, sampler %.0f, bookkeeping %.0f cycles per step" % (v[8] / steps, v[9] / steps, v[10] / steps))
