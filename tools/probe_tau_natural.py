"""Development probe: tau-leaping on a natural (index-case) epidemic that has grown large: few compartments, many hosts each."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgsim_amd import Simulator


def run(sites, P, steps, mut=0.01, N=10 ** 8):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=P, seed=5)
        s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(mut)
        if P > 1:
            s.set_migration_probability(0.01)
        s.set_population_size(N)
        s.simulate(20000, sample_size=10 ** 9)
        m = s.simulation
        for k in range(4):
            t0 = time.time()
            s.simulate(steps, sample_size=10 ** 12, method="tau")
            wall = time.time() - t0
            eng = m._engine
            print("sites=%d P=%d: infected %d (max compartment %d), %d steps: kernels %.1f ms (%.3f ms/step), wall %.2f s, drawn %d, t=%.3f" % (
                sites, P, m.globalInfectious, m.infectious.max(), steps, eng.last_kernel_ms, eng.last_kernel_ms / steps, wall,
                eng.last_events_drawn, m.currentTime), file=sys.stderr, flush=True)


if __name__ == "__main__":
    run(2, 3, 100)
    run(6, 8, 100)
