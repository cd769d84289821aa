"""Replicate ensembles on the GPU: every replicate must be the trajectory a single seeded Simulator produces
(bit for bit), and the on-device summary trajectories must equal a replay of that replicate's event log."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu


def build(name):
    from vgsim_amd import Simulator
    with helpers.quiet():
        sim, phases = models.build(Simulator, name)
    phases[0][0](sim)
    return sim


@pytest.mark.parametrize("name,n_events", [("g9_short", 3000), ("p70", 4000), ("c3_s5_p16", 3000)])
def test_replicates_equal_single_runs(oracle_mod, name, n_events):
    from vgsim_amd.ensemble import Ensemble
    R = 6
    sim = build(name)
    base = sim.simulation.user_seed
    seeds = np.array([base, base + 1, base + 7, base + 100, 5, 123456789], dtype=np.int64)
    ens = Ensemble(sim, R, seeds=seeds)
    T = 33
    res = ens.simulate(n_events, sample_size=10 ** 9, record_events=True, traj_points=T, traj_window=(0.0, 8.0))
    traj = ens.trajectories()
    assert traj.shape == (R, T, sim.simulation.popNum, 2)
    for r in range(R):
        from vgsim_amd import Simulator
        ctor, _ = models.CASES[name]
        with helpers.quiet():
            one = Simulator(**dict(ctor, seed=int(seeds[r])))
        models.CASES[name][1][0][0](one)
        m = one.simulation
        assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200, log_mode=oracle_mod.LOG_PORTABLE) == 0
        chain = ens.replicate_events(r)
        assert res.events[r] == m.events.ptr
        assert np.array_equal(chain, m.events.as_array()[:, :m.events.ptr]), "replicate %d" % r
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible)
        assert st.currentTime == m.currentTime and st.good_attempt == m.good_attempt
        # replay the log: totals per population just before each grid time
        P, S = m.popNum, m.susNum
        tot_i = m.initial_infectious.sum(axis=1).astype(float)
        tot_s = m.initial_susceptible.sum(axis=1).astype(float)
        grid = np.linspace(0.0, 8.0, T)
        want = np.zeros((T, P, 2))
        j = 0
        for e in range(m.events.ptr):
            t = m.events.times[e]
            while j < T and grid[j] < t:
                want[j, :, 0], want[j, :, 1] = tot_i, tot_s
                j += 1
            ty, pop, npop = m.events.types[e], m.events.populations[e], m.events.newPopulations[e]
            if ty == 0:
                tot_i[pop] += 1; tot_s[pop] -= 1
            elif ty in (1, 2):
                tot_i[pop] -= 1; tot_s[pop] += 1
            elif ty == 5:
                tot_i[npop] += 1; tot_s[npop] -= 1
        while j < T:
            want[j, :, 0], want[j, :, 1] = tot_i, tot_s
            j += 1
        assert np.array_equal(traj[r], want), "trajectory of replicate %d" % r
    ens.close()


def test_ensemble_without_log_counts_the_same(oracle_mod):
    from vgsim_amd.ensemble import Ensemble
    sim = build("stress_h64")
    ens = Ensemble(sim, 4)
    a = ens.simulate(5000, sample_size=10 ** 9, record_events=True)
    b = ens.simulate(5000, sample_size=10 ** 9, record_events=False)
    assert np.array_equal(a.events, b.events) and np.array_equal(a.loop_iterations, b.loop_iterations)
    ens.close()
