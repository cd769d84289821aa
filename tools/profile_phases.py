"""Development aid: per-phase cycle shares of the direct kernel (diagnostic build, `make -C vgsim_amd/csrc prof`).
Run with VGX_LIBRARY=vgsim_amd/libvgx_prof.so.  Read the SHARES, not the totals (stamps serialise the loop)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from probe import c3_model
from vgsim_amd import _capi

NAMES = ["0 loop top/tail", "1 RNG+log+time", "2 pop select", "3 tile load+tE", "4 row_select", "5 event apply",
         "6 list ops", "7 add_event", "8 birth_update", "9 tE_fill", "10 row_sum", "11 immune+popRate",
         "12 refresh_cum", "13 full part", "14 refresh_mig", "15 lockdown/round1"]


def run(sites, P, R, N):
    s = c3_model(sites, P)
    m = s.simulation
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=R)
    m.events.CreateEvents(N)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.arange(2020, 2020 + R))
    o = _capi.VgxRunOpts(); o.record_events = 0
    eng._check(eng.lib.vgx_simulate_direct(eng.handle, N, 10 ** 12, -1.0, 200, C.byref(o)))
    ev = eng.counters(0).ev_ptr
    out = np.zeros(16, dtype=np.int64)
    eng._check(eng.lib.vgx_get_profile(eng.handle, 0, _capi._p(out)))
    tot = out.sum()
    print("sites=%d P=%d R=%d N=%d: events(rep0)=%d kernel=%.1f ms, %.0f cycles/event" % (sites, P, R, N, ev, eng.last_kernel_ms, tot / max(ev, 1)))
    for n, v in zip(NAMES, out):
        print("   %-20s %8.0f cyc/event  %5.1f%%" % (n, v / max(ev, 1), 100.0 * v / max(tot, 1)))
    eng.close()


if __name__ == "__main__":
    run(0, 1, 1, 50000)
    run(8, 64, 1, 20000)
