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
        assert oracle_mod.run_direct(m, n_events, 10 ** 9, -1, 200) == 0
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


def test_rccl_gather_single_rank(tmp_path):
    """The collective of the ensemble layer on the real backend (nccl = RCCL), world_size 1 on this GPU: synchronous
    and asynchronous gather into a preallocated result, straight from device memory."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, io, contextlib
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import models
        from vgsim_amd import Simulator
        from vgsim_amd.ensemble import Ensemble
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        with contextlib.redirect_stdout(io.StringIO()):
            sim, ph = models.build(Simulator, "c3_s5_p16")
            ph[0][0](sim)
        ens = Ensemble(sim, 16)
        ens.simulate(2000, sample_size=10 ** 9, traj_points=17, traj_window=(0.0, 5.0))
        want = ens.trajectories()
        got = ens.gather_trajectories(dst=0)
        assert got.is_cuda and got.shape == (1,) + want.shape and np.array_equal(got[0].cpu().numpy(), want)
        out = torch.empty((1,) + want.shape, dtype=torch.float64, device="cuda")
        pend = ens.gather_trajectories(dst=0, out=out, async_op=True)
        ens.simulate(2000, sample_size=10 ** 9, traj_points=17, traj_window=(0.0, 5.0), seeds=np.arange(900, 916))
        res = pend.wait()
        torch.cuda.synchronize()
        assert res is out and np.array_equal(res[0].cpu().numpy(), want)      # the first call's trajectories
        assert not np.array_equal(ens.trajectories(), want)
        want2 = ens.trajectories()
        got32 = ens.gather_trajectories(dst=0, wire_dtype=torch.int32)      # the wire format bench.py uses
        assert got32.is_cuda and got32.dtype == torch.int32 and np.array_equal(got32[0].cpu().numpy(), want2)
        dist.destroy_process_group()
        print("RCCL_OK")
    """) % (root, root))
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"RCCL_OK" in p.stdout, p.stdout.decode()[-3000:]


@pytest.mark.parametrize("kernel", ["auto", "wave", "quadg"])
def test_continued_calls_keep_the_host_clock(oracle_mod, kernel):
    """Time-sliced runs through the C ABI: a second vgx_simulate_direct on the device-resident state (no vgx_set_state in between)
    continues every replicate where IT stopped — its own events.ptr, and its own clock: the event times of the second slice are the
    reference's libm sums continued from the first slice's last event, bit for bit what the oracle gives when one model runs both
    slices (the host clock of the first slice is rebuilt before its logs are overwritten; vgx_api.hip direct_core)."""
    import ctypes as C
    from vgsim_amd import Simulator, _capi
    ctor, phases = models.CASES["g5_short"]
    seeds = np.array([5, 6, 2020], dtype=np.int64)
    R, cap, t1, t2 = len(seeds), 30000, 2.5, 5.0
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    m = sim.simulation
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=R)
    m.events.CreateEvents(cap)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(seeds)
    o = _capi.VgxRunOpts(); o.record_events = 1
    o.kernel = {"auto": 0, "wave": 1, "quadg": 4}[kernel]
    eng._check(eng.lib.vgx_simulate_direct(eng.handle, cap, 10 ** 9, t1, 200, C.byref(o)))
    first_ptr = [eng.counters(r).ev_ptr for r in range(R)]
    eng._check(eng.lib.vgx_simulate_direct(eng.handle, cap, 10 ** 9, t2, 200, C.byref(o)))
    for r in range(R):
        with helpers.quiet():
            one = Simulator(**dict(ctor, seed=int(seeds[r])))
        phases[0][0](one)
        om = one.simulation
        assert oracle_mod.run_direct(om, cap, 10 ** 9, t1, 200) == 0
        p1 = om.events.ptr
        assert oracle_mod.run_direct(om, cap, 10 ** 9, t2, 200) == 0
        c = eng.counters(r)
        assert first_ptr[r] == p1 and c.ev_first_new == p1 and c.ev_ptr == om.events.ptr, (r, first_ptr[r], p1, c.ev_ptr, om.events.ptr)
        n = c.ev_ptr - p1
        assert n > 100
        times = np.zeros(n); cols = [np.zeros(n, dtype=np.int64) for _ in range(5)]
        eng._check(eng.lib.vgx_get_events(eng.handle, r, p1, n, _capi._p(times), *[_capi._p(x) for x in cols]))
        ref = om.events.as_array()[:, p1:om.events.ptr]
        assert np.array_equal(times, ref[0]), "replicate %d: times of the second slice" % r
        for k in range(5):
            assert np.array_equal(cols[k], ref[k + 1].astype(np.int64))
    eng.close()


@pytest.mark.parametrize("kernel", ["auto", "solo", "wave"])
def test_continued_call_after_a_slice_without_a_clock(oracle_mod, kernel):
    """First slice: event log, NO time limit (the latency kernel then runs without its device clock); second slice on the
    device-resident state WITH a time limit: it must start from the first slice's final time (the host clock's), stop where the
    oracle stops and carry the oracle's times."""
    import ctypes as C
    from vgsim_amd import Simulator, _capi
    ctor, phases = models.CASES["g5_short"]
    seeds = np.array([5, 6, 2020], dtype=np.int64)
    R, cap, s1, t2 = len(seeds), 500000, 1, 7.5
    with helpers.quiet():
        sim = Simulator(**ctor)
    phases[0][0](sim)
    m = sim.simulation
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=R)
    m.events.CreateEvents(cap)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(seeds)
    o = _capi.VgxRunOpts(); o.record_events = 1
    o.kernel = {"auto": 0, "wave": 1, "solo": 5}[kernel]
    eng._check(eng.lib.vgx_simulate_direct(eng.handle, cap, s1, -1.0, 200, C.byref(o)))
    if kernel != "wave":
        assert eng.lib.vgx_last_direct_kernel(eng.handle) == 5
    first_ptr = [eng.counters(r).ev_ptr for r in range(R)]
    eng._check(eng.lib.vgx_simulate_direct(eng.handle, cap, 10 ** 9, t2, 200, C.byref(o)))
    for r in range(R):
        with helpers.quiet():
            one = Simulator(**dict(ctor, seed=int(seeds[r])))
        phases[0][0](one)
        om = one.simulation
        assert oracle_mod.run_direct(om, cap, s1, -1, 200) == 0
        p1 = om.events.ptr
        assert oracle_mod.run_direct(om, cap, 10 ** 9, t2, 200) == 0
        c = eng.counters(r)
        assert first_ptr[r] == p1 and c.ev_first_new == p1 and c.ev_ptr == om.events.ptr, (r, first_ptr[r], p1, c.ev_ptr, om.events.ptr)
        n = c.ev_ptr - p1
        assert n > 100
        times = np.zeros(n); cols = [np.zeros(n, dtype=np.int64) for _ in range(5)]
        eng._check(eng.lib.vgx_get_events(eng.handle, r, p1, n, _capi._p(times), *[_capi._p(x) for x in cols]))
        ref = om.events.as_array()[:, p1:om.events.ptr]
        assert np.array_equal(times, ref[0]), "replicate %d: times of the second slice" % r
        for k in range(5):
            assert np.array_equal(cols[k], ref[k + 1].astype(np.int64))
    eng.close()
