import contextlib, io, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from vgsim_amd import Simulator
with contextlib.redirect_stdout(io.StringIO()):
    s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.4)
s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
m = s.simulation
done = 0
while done < 6000:
    with contextlib.redirect_stdout(io.StringIO()):
        s.simulate(3000, sample_size=10 ** 15, method="tau", record_multievents=False)
    done += 3000
I = m.infectious
print("occupied", int((I != 0).sum()), "of", I.size)
for c in (16, 64, 256, 1024, 4096, 65536):
    nz = (I.reshape(I.shape[0], -1, c) != 0).any(axis=2)
    print("chunks of %6d cells: %.2f %% non-empty" % (c, 100.0 * nz.mean()))
col = (I != 0).any(axis=0)
print("haplotypes occupied somewhere: %d of %d" % (col.sum(), col.size))
print("max count", int(I.max()), "cells >= 67:", int((I >= 67).sum()))
