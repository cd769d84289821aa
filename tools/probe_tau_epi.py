"""Development aid: the large-epidemic case of tools/probe_tau_small.py alone (4096 haplotypes x 8 populations, 10^7 hosts
each, 2300 tau steps from an index-case warm-up), for rocprofv3; `large`: 65536 haplotypes x 16 populations, 2000 steps.  With
the diagnostic build (VGX_LIBRARY=vgsim_amd/libvgx_prof.so) prints the phase stamps of the draw_big and events kernels."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import helpers
from vgsim_amd import Simulator
LARGE = len(sys.argv) > 1 and sys.argv[1] == "large"   # 65536 haplotypes x 16 populations, 1500 + 500 steps
with helpers.quiet():
    s = Simulator(number_of_sites=8 if LARGE else 6, populations_number=16 if LARGE else 8, seed=7)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
s.set_total_migration_probability(0.02); s.set_population_size(10 ** 7)
with helpers.quiet():
    s.simulate(2000, sample_size=10 ** 12)
    s.simulate(1500 if LARGE else 300, sample_size=10 ** 12, method="tau", record_multievents=False)
    NT = 500 if LARGE else 2000
    t0 = time.time(); s.simulate(NT, sample_size=10 ** 12, method="tau", record_multievents=False); t1 = time.time()
m = s.simulation
print("steps/s", NT / (t1 - t0), "infected", m.globalInfectious, "occupied", int((m.infectious > 0).sum()), "max", int(m.infectious.max()))
if os.environ.get("VGX_LIBRARY"):   # diagnostic build: phases of vgx_tau_draw_big_kernel (lane 0 of every wavefront)
    import ctypes as C
    from vgsim_amd import _capi
    out = (C.c_ulonglong * 8)()
    C.CDLL(_capi.LIB_PATH).vgx_tau_get_big_profile(out)
    v = list(out); tot = float(sum(v[:7]))
    for nme, x in zip(["setup loads", "channel means", "draws", "books", "sums + leader", "flush", "end"], v[:7]):
        print("  %-14s %5.1f %%  %8.0f cycles per iteration" % (nme, 100 * x / tot, x / max(v[7], 1)))
    out = (C.c_ulonglong * 16)()
    C.CDLL(_capi.LIB_PATH).vgx_tau_get_profile(out, 0)
    v = [float(x) for x in out]
    names = ["prologue", "bursts (second half)", "after the draws", "rescue tests", "staged list -> global", "epilogue", None,
             "  rates + number of events", "  split", "  mutants", "  migrants", "  tallies", None, None, "bursts (first half)"]
    tot = sum(v[:6]) + sum(v[7:12]) + v[14]
    print("events kernel: %d wavefronts, %.1f rounds each, %.0f cycles each" % (v[13], v[12] / max(v[13], 1), tot / max(v[13], 1)))
    for nme, x in zip(names, v[:15]):
        if nme:
            print("  %-28s %5.1f %%  %8.0f cycles per wavefront" % (nme, 100 * x / tot, x / max(v[13], 1)))
