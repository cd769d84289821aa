"""Development aid: where the wavefronts of vgx_tau_events_kernel spend their cycles on BASELINE config 4 (diagnostic build,
`make -C vgsim_amd/csrc prof`; run with VGX_LIBRARY=vgsim_amd/libvgx_prof.so).  Read the SHARES (stamps cost cycles)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from vgsim_amd import _capi
lib = C.CDLL(_capi.LIB_PATH)
out = (C.c_ulonglong * 16)()
lib.vgx_tau_get_profile(out, 1)
r = bench.tau_leg(0)
lib.vgx_tau_get_profile(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ["prologue", "wait for the count", "after the draws (lists, checks)", "rescue tests", "staged list -> global", "epilogue", None,
         "  rates + number of events", "  split into kinds", "  mutants", "  migrants", "  tallies"]
tot = v[:6].sum() + v[7:12].sum() + v[14]
v[6] = 0
print("tau leg: %.3f ms/step; wavefronts %d, rounds %d (%.2f per wavefront), %.0f cycles per wavefront"
      % (r["ms_per_step"], v[13], v[12], v[12] / max(v[13], 1), tot / max(v[13], 1)))
for n, x in zip(names, v[:12]):
    if n:
        print("  %-34s %5.1f %%   %9.0f cycles per wavefront   %9.0f per round" % (n, 100 * x / tot, x / max(v[13], 1), x / max(v[12], 1)))
print("  %-34s %5.1f %%   (first burst: queue entries + ok; the row 'wait for the count' is the second burst + LDS writes)" % ("first burst", 100 * v[14] / tot))
