"""Development aid: how the occupied compartments of bench.tau_warm_start's state spread over the regions of the occupancy lists."""
import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vgsim_amd import Simulator
with contextlib.redirect_stdout(io.StringIO()):
    s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.4)
s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
m = s.simulation
for k in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(3000, sample_size=10 ** 15, method="tau", record_multievents=False)
I = m.infectious
pn, h = np.nonzero(I)
reg = (h >> 16) * 8 + ((h >> 8) & 7)
key = pn.astype(np.int64) * 128 + reg
n = np.bincount(key, minlength=256 * 128)
print("occupied", len(h), "regions", n.size, "max n", n.max(), "regions > 1024:", int((n > 1024).sum()), "> 256:", int((n > 256).sum()), "empty:", int((n == 0).sum()))
tiles = n.reshape(256, 16, 8)
print("tiles with an overflowing region:", int((tiles > 1024).any(axis=2).sum()), "of", 256 * 16)
print("entries in overflowing regions:", int(n[n > 1024].sum()), " in regions > 256:", int(n[n > 256].sum()))
print("percentiles of n:", [int(np.percentile(n, q)) for q in (50, 90, 99, 99.9)])
cnt = I[pn, h]
print("counts: max", cnt.max(), " >=255:", int((cnt >= 255).sum()), " >= 1000:", int((cnt >= 1000).sum()), ">= 1e4:", int((cnt >= 10000).sum()))
