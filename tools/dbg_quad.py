import sys, struct
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np, ctypes as C
import helpers, models
from vgsim_amd import Simulator, _capi
with helpers.quiet():
    sim, phases = models.build(Simulator, "g5")
    phases[0][0](sim)
try:
    with helpers.quiet():
        sim.simulate(100000, kernel="quad")
    print("no error")
except Exception as ex:
    print(ex)
    eng = sim.simulation._engine
    out = np.zeros(16, dtype=np.int64)
    eng.lib.vgx_get_profile(eng.handle, 0, out.ctypes.data_as(C.POINTER(C.c_int64)))
    names = "n spi tpi ti_spi rr rm total before maxn2 gI ti_tpi ln0 ev_ptr loops totalMig choose".split()
    for k,v in zip(names,out):
        if k in ("rr","rm","totalMig","choose"): v = struct.unpack("d", struct.pack("q", int(v)))[0]
        print(k, v)
