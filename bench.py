#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on its config-3 workload: simulated events/sec, direct Gillespie,
65 536 haplotypes (8 sites) x 64 populations, replicate ensemble, on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: `--replicates` independent seeded trajectories per GPU,
`--events` recorded events each, from the single index case (the only initial condition the reference API can
express), parameters of SURVEY.md §8(d) config 3.  Weak scaling: per-GPU work is fixed, replicates are sharded
one wavefront each, no data-path collective; the only collective is one RCCL gather of the summary
trajectories per step.  The JSON line also carries the roofline of the dominant kernel (device time from HIP
events on the engine's stream) and the CPU baseline (the oracle = op-for-op port of the reference's dense
algorithm, timed on this host, rank 0, N=1 only).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SITES, POPS, SUS = 8, 64, 1           # BASELINE.json configs[2]: 4^8 = 65 536 haplotypes x 64 populations
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def pmc_traffic(leg, config):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_direct_c3.json: separate rocprofv3 --pmc runs of
    this same command, corrected as MI355X_MICROARCH.md prescribes), or None when the configuration differs."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_direct_c3.json")))[leg]
        if all(d["config"].get(k) == v for k, v in config.items()):
            return d["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def libm_probe_match():
    """Whether this host's log() reproduces the 4096 (u, log u) pairs recorded on the fixture host (tests/golden/libm_probe.json):
    where it does, the engine's event TIMES are bit-identical to the reference goldens (the integer rows are everywhere)."""
    import math
    try:
        probe = json.load(open(os.path.join(ROOT, "tests", "golden", "libm_probe.json")))
        return all(math.log(float.fromhex(u)) == float.fromhex(v) for u, v in probe)
    except Exception:
        return None


def make_simulator(seed):
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=SITES, populations_number=POPS, number_of_susceptible_groups=SUS, seed=seed)
    s.set_transmission_rate(2.5)
    s.set_recovery_rate(0.9)
    s.set_sampling_rate(0.1)
    s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01)
    s.set_population_size(10 ** 7)
    return s


def cpu_sample(seed, seconds_target):
    """One bounded sample of the config-3 workload on the oracle (dense = the reference's own O(H*S*P)-per-event
    algorithm), one core: (events, seconds, start-up events)."""
    from oracle import oracle
    oracle.build()
    sim = make_simulator(seed)
    m = sim.simulation
    t0 = time.time()
    oracle.run_direct(m, 101, 10 ** 12, -1, 200)      # includes PrepareParameters/UpdateAllRates and Restarts
    t1 = time.time()
    n1 = m.events.ptr
    per_event = max((t1 - t0) / max(n1, 1), 1e-6)
    n2 = int(max(50, min(5000, seconds_target / per_event)))
    t2 = time.time()
    oracle.run_direct(m, n2, 10 ** 12, -1, 200)       # continues the same trajectory
    t3 = time.time()
    return m.events.ptr - n1, t3 - t2, n1


def cpu_baseline(seconds_target=45.0):
    """The oracle (oracle/vgx_oracle.c) on the same workload: one host core (the headline baseline) and, beside it,
    one independent seeded process per available core (a CPU user's way to run an ensemble)."""
    import subprocess
    from oracle import oracle
    oracle.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    # the per-core processes start first (plain interpreters: nothing of this process' GPU state is inherited)
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "d, t, n = bench.cpu_sample(int(sys.argv[1]), %f); print(d, t)" % (ROOT, seconds_target * 0.6))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(3000 + k)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
             for k in range(cores)] if cores > 1 else []
    outs = [pr.communicate()[0].decode().split() for pr in procs]
    rates = [float(o[0]) / max(float(o[1]), 1e-9) for o in outs if len(o) == 2]
    done, secs, n1 = cpu_sample(2020, seconds_target)
    out = {"value": done / max(secs, 1e-9), "unit": "events/s", "cores": 1, "kind": "port",
           "sample": "%d events of one config-3 trajectory (seed 2020) after a %d-event start, oracle in the "
                     "reference's dense mode, %.1f s" % (done, n1, secs)}
    if rates:
        out["all_cores"] = {"value": sum(rates), "unit": "events/s", "cores": len(rates),
                            "sample": "one independent process per available core, same workload, seeds 3000.."}
    # one trajectory of config 2 (H = P = 1: the regime where a CPU core is at its best)
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        c2 = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
    c2.set_transmission_rate(4.0); c2.set_recovery_rate(1.5); c2.set_sampling_rate(0.3)
    t6 = time.time()
    oracle.run_direct(c2.simulation, 1000000, 10 ** 12, -1, 200)
    t7 = time.time()
    out["config2"] = {"value": c2.simulation.events.ptr / max(t7 - t6, 1e-9), "unit": "events/s", "cores": 1,
                      "events": int(c2.simulation.events.ptr)}
    # backward pass (GetGenealogy) of the oracle on a 300 000-event chain of the same model (forward run in the oracle's
    # occupied-haplotypes-only mode, which is bit-identical to the dense one)
    sim2 = make_simulator(2020)
    m2 = sim2.simulation
    oracle.run_direct(m2, 300000, 10 ** 12, -1, 200, sparse=True)
    if m2.sCounter >= 2:
        t4 = time.time()
        oracle.run_genealogy(m2, 7)
        t5 = time.time()
        out["genealogy"] = {"value": m2.events.ptr / max(t5 - t4, 1e-9), "unit": "events/s", "events": int(m2.events.ptr),
                            "samples": int(m2.sCounter), "seconds": t5 - t4}
    return out


def spread_leg(device, mode="exact", replicates=1024, events=20000, occupied=4096):
    """Config 3 at *spread* occupancy: every population starts with `occupied` distinct
    haplotypes (1-3 infected each, written straight into the model's arrays — the reference's own
    set_infectious is broken, pyx:1596).  Here the per-event work is the stream over the chosen population's
    occupancy list, so the kernel is priced against HBM with the same byte formula as the headline leg."""
    import numpy as np
    from vgsim_amd.ensemble import Ensemble
    sim = make_simulator(2020)
    m = sim.simulation
    H = m.hapNum
    rng = np.random.default_rng(2020)
    for pn in range(POPS):
        haps = rng.choice(H, size=occupied, replace=False)
        m.infectious[pn, haps] = rng.integers(1, 4, size=occupied)
        m.susceptible[pn, 0] -= int(m.infectious[pn].sum())
    ens = Ensemble(sim, replicates, device=device)
    seeds = 5000 + np.arange(replicates, dtype=np.int64)
    best = None
    for it in range(2):   # first launch is the warm-up
        res = ens.simulate(events, sample_size=10 ** 12, record_events=True, traj_points=0, seeds=seeds + it * replicates,
                           mode=mode)
        best = res
    st = ens.replicate_state(0)
    nocc = float((st.infectious != 0).sum(axis=1).mean())
    ms = best.kernel_ms
    ev = best.total_events
    # exact, one rate class, long-list row kernel: the ONE-BYTE counts of the whole list for the rate refresh, which leaves the
    # running sum at the end of every tile (8 B per 64 entries written, read back by the next choice) + one tile of 4-byte counts
    # for the choice.  fast: tile sums + one tile (4-byte counts and haplotypes)
    exact_bytes = 1.0 * nocc + 16.0 * (nocc / 64.0) + 4.0 * 64
    bpe = (exact_bytes if mode == "exact" else 8.0 * (nocc / 64.0) + 8.0 * 64) + 8.0 * POPS + 28.0 + 8.0
    traffic = pmc_traffic("spread_occupancy" if mode == "exact" else "spread_occupancy_fast",
                          {"replicates_per_gpu": replicates, "events_per_replicate": events, "occupied": occupied, "mode": mode})
    out = {"workload": "BASELINE config 3, spread occupancy: %d occupied haplotypes per population at start "
                       "(mean %.0f at the end), %d replicates x %d events, %s mode" % (occupied, nocc, replicates, events, mode),
           "value": ev / (ms * 1e-3), "unit": "events/s (device time)", "kernel_ms_per_launch": ms,
           "roofline": {"bound": "hbm", "achieved": ev * bpe / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ev * bpe / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                        "bytes_per_event": bpe, "mean_occupancy_list_len": nocc}}
    if mode == "exact":
        # per event: the refreshed population's whole list (pyx:519-528), on average half of the selected population's
        # list (fast_choose.pxi:22-25), the popRate and migPopRate totals (pyx:537-546)
        rows = 4     # vgx_quad.hip: every chain instruction serves four replicates
        # the choice starts at the tile the cached running sums point to (64 steps instead of half the list)
        steps = 1.0 * nocc + 64.0 + 2.0 * POPS
        out["chain_bound"] = {"bound": "dependent f64 additions in the reference's order (one v_fmac_f64 per term)",
                              "steps_per_event": steps, "achieved": ev * steps / (ms * 1e-3), "peak": CHAIN_STEPS_PER_S * rows,
                              "unit": "chain steps/s", "frac": ev * steps / (ms * 1e-3) / (CHAIN_STEPS_PER_S * rows),
                              "chains_per_instruction": rows}
    ens.close()
    return out


def make_general_c3(seed=2020):
    """BASELINE config 3's shape as a GENERAL model (what testing/getting_reference.py:14-19 does at scale): two susceptibility groups
    (recovered hosts are half as susceptible and lose their immunity at rate 0.02), one fitter haplotype (transmission 3.0 instead of
    2.5: a second rate class), an NPI on every deme (contact density 0.5 above 1 % infected, off below 0.2 %)."""
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=SITES, populations_number=POPS, number_of_susceptible_groups=2, seed=seed)
    s.set_transmission_rate(2.5)
    s.set_transmission_rate(3.0, haplotype=5)
    s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_susceptibility_type(1)
    s.set_susceptibility(0.5, susceptibility_type=1)
    s.set_immunity_transition(0.02, source=1, target=0)
    s.set_total_migration_probability(0.01)
    s.set_population_size(10 ** 7)
    s.set_npi([0.5, 0.01, 0.002])
    return s


def general_c3_leg(device, replicates=16384, events=50000):
    """config3_general: the headline shape with several susceptibility groups, rate classes and NPIs — the general row kernel
    (vgx_quadg.hip) on lists of config-3 length; exact mode, index-case start (natural occupancy)."""
    import numpy as np
    from vgsim_amd.ensemble import Ensemble
    ens = Ensemble(make_general_c3(), replicates, device=device)
    res = None
    for it in range(2):
        res = ens.simulate(events, sample_size=10 ** 12, record_events=True,
                           seeds=2020 + it * replicates + np.arange(replicates, dtype=np.int64))
    st = ens.replicate_state(0)
    occ = (st.infectious != 0).sum(axis=1).astype(float)
    w = st.infectious.sum(axis=1).astype(float)
    out = {"workload": "BASELINE config 3's shape as a general model: 65536 haplotypes x 64 populations x 2 susceptibility groups, one fitter "
                       "haplotype (2 rate classes), immunity loss 0.02, NPI [0.5, 0.01, 0.002] on every deme; %d replicates x %d events, exact mode, "
                       "index-case start" % (replicates, events),
           "value": res.total_events / (res.kernel_ms * 1e-3), "unit": "events/s (device time)", "kernel": ens.engine.last_kernel,
           "kernel_ms_per_launch": res.kernel_ms, "event_weighted_list_len": float((occ * w).sum() / max(w.sum(), 1.0)),
           "mean_occupancy_list_len": float(occ.mean()), "immunity_transitions_share": None}
    ens.close()
    return out


def c4_direct_leg(device, replicates=1024, events=20000):
    """Direct Gillespie at BASELINE config 4's shape (2^20 haplotypes x 256 populations, migration), index-case start: the
    general instantiation of the wave kernel (populations in tiles of 64, 37 KB of LDS tables per wavefront)."""
    import numpy as np
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=10, populations_number=256, seed=2020)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    ens = Ensemble(s, replicates, device=device)
    res = None
    for it in range(2):
        res = ens.simulate(events, sample_size=10 ** 12, record_events=True,
                           seeds=2020 + it * replicates + np.arange(replicates, dtype=np.int64))
    out = {"workload": "direct Gillespie at config 4's shape: 1048576 haplotypes x 256 populations, %d replicates x %d events, "
                       "exact mode, index-case start" % (replicates, events),
           "value": res.total_events / (res.kernel_ms * 1e-3), "unit": "events/s (device time)", "kernel_ms_per_launch": res.kernel_ms,
           "kernel": ens.engine.last_kernel}
    ens.close()
    # ... and ONE trajectory of that shape (256 populations: beyond the 64 of the list-resident latency kernel, so a lone wavefront of
    # the general wave kernel runs it)
    one = Ensemble(s, 1, device=device)
    for it in range(2):
        r1 = one.simulate(events // 2, sample_size=10 ** 12, record_events=True, seeds=np.array([2020 + it], dtype=np.int64))
    out["single_trajectory"] = {"events_per_s": r1.total_events / (r1.kernel_ms * 1e-3), "kernel": one.engine.last_kernel, "events": int(r1.total_events)}
    one.close()
    return out


# ---- the reference's only published benchmark for this path: data/Table 3 ----------------------------------
# seconds for 10^8 direct-Gillespie iterations, ONE trajectory, hardware not stated (data/Table 3/table.txt:9-17)
TABLE3_PUBLISHED_S = {(0.001, 2): 28.7, (0.001, 5): 30.0, (0.001, 10): 31.9, (0.001, 20): 35.1, (0.001, 50): 47.2, (0.001, 100): 69.3,
                      (0.1, 2): 30.3, (0.1, 5): 31.9, (0.1, 10): 33.8, (0.1, 20): 37.0, (0.1, 50): 50.3, (0.1, 100): 73.0}


def make_table3(K, M, seed=2023):
    """The model of data/Table 3/Table 3.py:5-22 through the current public setters (the script's `set_lockdown` and
    `set_migration_probability(total_probability=)` are today's `set_npi` and `set_total_migration_probability`):
    16 haplotypes (2 sites), 3 susceptibility groups, K demes of 2e9/K hosts, four rate classes, NPI on every deme."""
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=2, populations_number=K, number_of_susceptible_groups=3, seed=seed)
    s.set_transmission_rate(2.5)
    s.set_transmission_rate(4.0, haplotype='GG')
    s.set_recovery_rate(0.9)
    s.set_sampling_rate(0.1)
    s.set_total_migration_probability(M)
    s.set_population_size(int(2 * 10 ** 9 / K))
    s.set_susceptibility(1.0, susceptibility_type=0)
    s.set_susceptibility(0.0, susceptibility_type=1)
    s.set_susceptibility(0.5, susceptibility_type=1, haplotype='C*')
    s.set_susceptibility(1.0, susceptibility_type=1, haplotype='G*')
    s.set_susceptibility(0.0, susceptibility_type=2)
    s.set_susceptibility_type(1)
    s.set_susceptibility_type(2, haplotype='C*')
    s.set_susceptibility_type(2, haplotype='G*')
    s.set_immunity_transition(0.01, target=0)
    s.set_npi([0.1, 0.02, 0.01])
    return s


def table3_cpu(K, M, iterations=3000000):
    """The oracle (op-for-op port of the reference's algorithm) on the Table-3 model, one host core: events/s."""
    from oracle import oracle
    oracle.build()
    m = make_table3(K, M).simulation
    t0 = time.time()
    oracle.run_direct(m, iterations, 10 ** 12, -1, 200)
    t1 = time.time()
    return {"value": m.events.ptr / max(t1 - t0, 1e-9), "unit": "events/s", "cores": 1, "kind": "port",
            "events": int(m.events.ptr), "s_per_1e8_iterations": 1e8 * (t1 - t0) / max(m.events.ptr + m.migNonPlus, 1)}


def table3_leg(device, cpu=True, replicates=16384, events=50000, single_events=300000, cells=None):
    """data/Table 3 (the reference's published timing): K in {2, 10, 100} demes x cumulative migration M in
    {0.001, 0.1}.  Per cell: ONE trajectory (the published quantity: seconds per 10^8 iterations, device time of the
    persistent kernel), an ensemble of `replicates` seeded trajectories (aggregate events/s, bit-exact mode), and the oracle on
    one host core."""
    import numpy as np
    from vgsim_amd.ensemble import Ensemble
    out = {"workload": "data/Table 3/Table 3.py:5-22: sites=2 (16 haplotypes), 3 susceptibility groups, K demes of 2e9/K hosts, "
                       "b=2.5 (GG: 4.0), d=0.9, s=0.1, 4 rate classes, immunity loss 0.01, NPI [0.1, 0.02, 0.01] on every deme, "
                       "seed 2023; exact mode",
           "published": "data/Table 3/table.txt:9-17: seconds per 1e8 iterations of one trajectory, hardware not stated",
           "cells": {}}
    for M in (0.001, 0.1):
        for K in (2, 10, 100):
            if cells is not None and (K, M) not in cells:
                continue
            cell = {"published_s_per_1e8": TABLE3_PUBLISHED_S[(M, K)], "published_events_per_s": 1e8 / TABLE3_PUBLISHED_S[(M, K)]}
            sim = make_table3(K, M)
            ens = Ensemble(sim, 1, device=device)
            res = None
            for it in range(2):
                res = ens.simulate(single_events, sample_size=10 ** 12, record_events=True, seeds=np.array([2023 + it], dtype=np.int64))
            it1 = float(res.loop_iterations.sum())
            cell["single_trajectory"] = {"events_per_s": res.total_events / (res.kernel_ms * 1e-3),
                                         "iterations_per_s": it1 / (res.kernel_ms * 1e-3),
                                         "s_per_1e8_iterations": 1e8 * res.kernel_ms * 1e-3 / max(it1, 1.0),
                                         "vs_baseline": (it1 / (res.kernel_ms * 1e-3)) / (1e8 / TABLE3_PUBLISHED_S[(M, K)])}
            ens.close()
            # ... and what the published figure is: the wall clock of Simulator.simulate(N) (data/Table 3/Table 3.py:23-25), end to end
            # through the facade: parameters and state to the device, the kernel, the event log back incl. the host clock (libm
            # times), the counters, Stats.  The engine exists already (a first short call), as the reference's object does.
            one = make_table3(K, M)
            with contextlib.redirect_stdout(io.StringIO()):
                one.simulate(1000, sample_size=10 ** 12)
                m1 = one.simulation
                ev0, it0 = m1.events.ptr, m1.events.ptr + m1.migNonPlus
                t0 = time.perf_counter()
                one.simulate(single_events, sample_size=10 ** 12)
                tw = time.perf_counter() - t0
            evw, itw = m1.events.ptr - ev0, m1.events.ptr + m1.migNonPlus - it0
            cell["single_trajectory"]["wall"] = {
                "what": "Simulator.simulate(%d) end to end (set_params, set_state, kernel, event log + host clock back, Stats); second call on the object" % single_events,
                "seconds": tw, "kernel_ms": m1._engine.last_kernel_ms, "events_per_s": evw / tw, "iterations_per_s": itw / tw,
                "s_per_1e8_iterations": 1e8 * tw / max(itw, 1), "vs_baseline": (itw / tw) / (1e8 / TABLE3_PUBLISHED_S[(M, K)]),
                "kernel_share_of_wall": m1._engine.last_kernel_ms * 1e-3 / tw}
            R = replicates if K <= 16 else max(replicates // 4, 1024)
            ens = Ensemble(sim, R, device=device)
            for it in range(2):
                res = ens.simulate(events, sample_size=10 ** 12, record_events=True,
                                   seeds=2023 + it * R + np.arange(R, dtype=np.int64))
            its = float(res.loop_iterations.sum())
            cell["ensemble"] = {"replicates": R, "events_per_replicate": events, "events_per_s": res.total_events / (res.kernel_ms * 1e-3),
                                "iterations_per_s": its / (res.kernel_ms * 1e-3), "kernel_ms_per_launch": res.kernel_ms,
                                "rejected_migration_share": 1.0 - res.total_events / max(its, 1.0),
                                "vs_baseline": (its / (res.kernel_ms * 1e-3)) / (1e8 / TABLE3_PUBLISHED_S[(M, K)])}
            # the counter-based stream (mode='fast_philox', north_star's "Philox-style counter-based RNG"): the same kernels' exact arithmetic
            # on that stream (every replicate = the oracle fed with it: tests/test_hip_quadg.py::test_counter_based_stream_on_the_general_row_kernel)
            for it in range(2):
                res = ens.simulate(events, sample_size=10 ** 12, record_events=True, mode="fast_philox",
                                   seeds=2023 + it * R + np.arange(R, dtype=np.int64))
            cell["ensemble_philox"] = {"replicates": R, "events_per_s": res.total_events / (res.kernel_ms * 1e-3), "kernel": ens.engine.last_kernel}
            ens.close()
            if cpu:
                cell["cpu_baseline"] = table3_cpu(K, M)
            out["cells"]["K=%d,M=%g" % (K, M)] = cell
    k10 = out["cells"].get("K=10,M=0.001")
    if k10 is None:
        return out
    out["value"] = k10["ensemble"]["events_per_s"]
    out["unit"] = "events/s (device time, K=10 M=0.001 ensemble)"
    out["vs_baseline"] = k10["ensemble"]["vs_baseline"]
    return out


# Sequential f64 additions in the reference's order run as a chain of dependent v_fmac_f64 (DPP) steps, one term per
# step; measured issue rate of that chain: 1.9 ns per step and SIMD at >= 4 waves/SIMD (profiles/r01_microbench_chain.txt),
# 1024 SIMDs.  This is the bound of the exact mode on long occupancy lists, reported beside the HBM one.
CHAIN_STEPS_PER_S = 1024 / 1.9e-9


def genealogy_leg(device, events=300000):
    """the backward pass (GetGenealogy, pyx:743-1000) over one config-3 chain produced on the
    device, in libvgx's host code (one core).  The oracle's literal restatement is timed in the cpu_baseline leg."""
    from vgsim_amd import _capi
    sim = make_simulator(2020)
    with contextlib.redirect_stdout(io.StringIO()):
        sim.simulate(events, sample_size=10 ** 12)
    m = sim.simulation
    t0 = time.perf_counter()
    out = _capi.get_genealogy(m, 7)
    t1 = time.perf_counter()
    return {"workload": "backward pass over one config-3 chain (host code)", "events": int(m.events.ptr), "samples": int(m.sCounter),
            "tree_nodes": int(out["nodes_used"]), "value": m.events.ptr / (t1 - t0), "unit": "events/s (one host core)",
            "seconds": t1 - t0}


def single_leg(device):
    """One trajectory at a time (the classic API): latency of the sequential event loop on one wavefront."""
    import numpy as np
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    out = {"workload": "a single replicate (one wavefront): events/s of device time"}
    with contextlib.redirect_stdout(io.StringIO()):
        c2 = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
    c2.set_transmission_rate(4.0); c2.set_recovery_rate(1.5); c2.set_sampling_rate(0.3)
    out["kernel"] = {}
    for name, sim, n in (("config3", make_simulator(2020), 100000), ("config2", c2, 200000), ("config3_general", make_general_c3(), 50000)):
        ens = Ensemble(sim, 1, device=device)
        res = None
        for it in range(2):
            res = ens.simulate(n, sample_size=10 ** 12, record_events=True, seeds=np.array([2020 + it], dtype=np.int64))
        out[name] = res.total_events / (res.kernel_ms * 1e-3)
        out["kernel"][name] = ens.engine.last_kernel
        ens.close()
    return out


def rowscan_leg(device, rows=4096):
    """K3: the dense propensity row pass of the reference's layout (UpdateRates infect branch + fastChoose over hapPopRate,
    pyx:518-528, fast_choose.pxi:18-31) streamed for `rows` (replicate, population) rows of config 3 (H = 65 536, S = 1)
    resident in HBM.  Algorithmic bytes per row visit: SURVEY.md 8(d)'s H*(84+16S) + 16P + 48."""
    import numpy as np
    from vgsim_amd import _capi
    H, S = 4 ** SITES, SUS
    rng = np.random.default_rng(3)
    inf = (rng.integers(1, 4, size=(1, H)) * (rng.random((1, H)) < 0.0625)).astype(np.int64)     # 4096 occupied haplotypes
    r123 = np.broadcast_to(np.array([0.9, 0.1, 0.08]), (1, H, 3)).copy()
    out = _capi.propensity_scan(inf, r123, np.arange(H), np.full(H, 2.5), np.ones((H, S)), np.full((1, S), 1e7 - 8000.0),
                                np.array([9.9e-8]), rng.random(rows), bench_rows=rows, repeats=5)
    survey_row = H * (84 + 16 * S) + 16 * POPS + 48
    ms = out["ms_update"] + out["ms_choose"]
    # what has to stream from HBM per row visit: the row's own arrays (infectious 8, eventHapPopRate[1:4] 24 read; the four rate
    # arrays 24 + 8S written; on average half of hapPopRate read again by the choice).  The per-haplotype parameter arrays of
    # SURVEY's count (numToHap, bRate, susceptibility: 16 + 8S B/hn) are shared by all rows and stay in L2 / Infinity Cache.
    per_row = H * (8 + 24 + 24 + 8 * S) + 4 * H
    exact = {"row_arrays_read": H * 32, "row_arrays_written": H * (24 + 8 * S), "choice_mean": 4 * H,
             "shared_parameter_arrays_cached": H * (16 + 8 * S)}
    traffic = None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_rowscan.json")))
        if d["config"] == {"rows": rows, "H": H, "S": S}:
            traffic = d["hbm_bytes_per_pass"]
    except Exception:
        pass
    return {"workload": "K3 dense propensity row pass: %d rows x %d haplotypes x %d group(s), row update + fastChoose" % (rows, H, S),
            "rows_per_s": rows / (ms * 1e-3), "unit": "row visits/s (device time)", "ms_update": out["ms_update"], "ms_choose": out["ms_choose"],
            "roofline": {"bound": "hbm", "achieved": rows * per_row / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": rows * per_row / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "vgx_rowscan_update_kernel + vgx_rowscan_choose_kernel",
                         "bytes_per_row_visit": per_row, "bytes_by_array": exact,
                         "survey_formula": {"bytes_per_row_visit": survey_row, "GBs": rows * survey_row / (ms * 1e-3) / 1e9,
                                            "note": "SURVEY.md 8(d) counts the shared parameter arrays per row visit; they are "
                                                    "cache-resident here, so this figure can exceed the HBM peak"},
                         # what this box's memory system gives plain streams of the same size (the row pass reads and writes 1 : 1)
                         "this_gpu_streams": stream_rates(device)}}


def stream_rates(device):
    """GB/s of three plain torch streams over 2 GiB arrays on this GPU (HIP events, 10 repeats): copy (read + write 1 : 1), sum (read
    only), a += b (2 reads, 1 write) — the practical ceilings beside the 8 TB/s pin rate."""
    import torch
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    b = torch.ones(n, dtype=torch.float64, device="cuda")

    def t(f, reps=10):
        f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    out = {"copy_GBs": 2 * n * 8 / t(lambda: a.copy_(b)) / 1e6, "read_only_sum_GBs": n * 8 / t(lambda: b.sum()) / 1e6,
           "axpy_GBs": 3 * n * 8 / t(lambda: torch.add(a, b, out=a)) / 1e6}
    del a, b
    torch.cuda.empty_cache()
    return out


def fast_leg(device, replicates, events, traj_points):
    """The headline workload (natural occupancy) in FAST mode: the row-per-replicate kernel vgx_quadf.hip (four replicates per
    wavefront, three wavefronts per SIMD: 24 576 replicates are two full rounds of the chip), device time of one launch after a
    warm-up; then the same with the counter-based stream (mode 2: Philox4x32-10 per lane), on the same kernel since round 3."""
    import numpy as np
    from vgsim_amd.ensemble import Ensemble
    requested = replicates
    replicates = 24576 if replicates >= 16384 else replicates
    ens = Ensemble(make_simulator(2020), replicates, device=device)
    res = None
    for it in range(2):
        res = ens.simulate(events, sample_size=10 ** 12, record_events=True, traj_points=traj_points,
                           traj_window=(0.0, 12.0), seeds=2020 + it * replicates + np.arange(replicates, dtype=np.int64),
                           mode="fast")
    out = {"workload": "headline workload in FAST mode (order-free sums, same PCG64 stream), %d replicates x %d events" % (replicates, events),
           "value": res.total_events / (res.kernel_ms * 1e-3), "unit": "events/s (device time)",
           "kernel": "vgx_quadf_kernel", "kernel_ms_per_launch": res.kernel_ms,
           "replicates": replicates, "requested_replicates": requested}
    # the same with the counter-based random stream (vgx_run_opts.mode = 2: Philox4x32-10, every draw formed on its own)
    for it in range(2):
        res = ens.simulate(events, sample_size=10 ** 12, record_events=True, traj_points=traj_points,
                           traj_window=(0.0, 12.0), seeds=2020 + it * replicates + np.arange(replicates, dtype=np.int64),
                           mode="fast_philox")
    out["philox_stream"] = {"value": res.total_events / (res.kernel_ms * 1e-3), "unit": "events/s (device time)", "kernel": "vgx_quadf_kernel",
                            "replicates": replicates, "kernel_ms_per_launch": res.kernel_ms}
    ens.close()
    return out


def c2_leg(device):
    """BASELINE config 2 (H=1, P=1, S=1, N=1e6): latency-bound — one dependent chain per event; reported as
    events/s per replicate and replicates in flight, for the wavefront-per-replicate kernel, for the kernel the
    engine picks by itself at the headline's 16 384 replicates (the one-class row kernel) and for the lane-per-replicate kernel
    (vgx_lanes.hip) that it picks for minimal models in very large ensembles."""
    import numpy as np
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020)
    s.set_transmission_rate(4.0); s.set_recovery_rate(1.5); s.set_sampling_rate(0.3)
    out = {"workload": "BASELINE config 2: 1 haplotype x 1 population, b=4.0 d=1.5 s=0.3, N=1e6"}
    for key, kernel, replicates, events in (("wave", "wave", 16384, 100000), ("ensemble_16384", "auto", 16384, 100000),
                                            ("lane", "auto", 262144, 10000)):
        ens = Ensemble(s, replicates, device=device)
        res = None
        for it in range(2):
            res = ens.simulate(events, sample_size=10 ** 12, record_events=True, kernel=kernel,
                               seeds=2020 + it * replicates + np.arange(replicates, dtype=np.int64))
        ms = res.kernel_ms
        out[key] = {"replicates_in_flight": replicates, "events_per_replicate": events,
                    "value": res.total_events / (ms * 1e-3), "unit": "events/s (device time)",
                    "events_per_s_per_replicate": res.total_events / (ms * 1e-3) / replicates, "kernel_ms_per_launch": ms,
                    "kernel": ens.engine.last_kernel}
        ens.close()
    out["value"] = max(out[k]["value"] for k in ("wave", "ensemble_16384", "lane"))
    out["unit"] = "events/s (device time)"
    return out


def tau_cpu_time(model, seconds_target=4.0, first=20):
    """The oracle's SimulatePopulation_tau (op-for-op port of pyx:2293-2593: every channel of the reference drawn in every step) on a
    host model, one core: steps per second and the cost of one channel-step.  Test infrastructure used as the CPU baseline only."""
    from oracle import oracle
    oracle.build()
    prop = oracle.prop_num(model)
    n, done, spent = first, 0, 0.0
    while True:
        t0 = time.perf_counter()
        rc = oracle.run_tau(model, n, 10 ** 15, -1, 200)
        dt = time.perf_counter() - t0
        assert rc == 0
        done += n; spent += dt
        if spent >= seconds_target or done >= 20000:
            break
        n = int(max(1, min(4 * n, (seconds_target - spent) / max(dt / n, 1e-9))))
    return {"value": done / spent, "unit": "steps/s", "cores": 1, "kind": "port", "steps": done, "seconds": spent,
            "channels_per_step": prop, "ns_per_channel_step": 1e9 * spent / (done * float(prop)),
            "sample": "%d tau steps of this model on one host core (every one of the reference's %d channels drawn per step)" % (done, prop)}


def tau_cpu_dense(sites, P, per_cell=3, seconds_target=4.0):
    """config 4's recipe at a shape the reference's dense channel arrays still fit: uniform fill, total migration 0.01."""
    from vgsim_amd import Simulator
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=P, seed=2020)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    m = s.simulation
    m.infectious[:] = per_cell
    m.susceptible[:, 0] -= per_cell * m.hapNum
    m.first_simulation = False
    out = tau_cpu_time(m, seconds_target, first=2)
    out["shape"] = "%d haplotypes x %d populations, %d infected per compartment" % (m.hapNum, P, per_cell)
    return out


def tau_warm_start(device, warm_steps=6000, timed_steps=100, seed=2020):
    """config 4 started the way SURVEY.md 8(d) prescribes instead of by a uniform fill: one index case, a high-mutation warm-up
    (mutation rate 0.4 per site, tau-leaping, until on average >= 4096 haplotypes per population are occupied), mutation rate back
    to 0.01, then the timed steps.  Natural occupancy: well under 1 % of the 2^28 compartments hold anyone, and the step kernels
    still stream the dense arrays."""
    import ctypes as C
    import numpy as np
    from vgsim_amd import Simulator, _capi
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=10, populations_number=256, seed=seed)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.4)
    s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    m = s.simulation
    t0 = time.perf_counter()
    done = 0
    while done < warm_steps:
        n = min(3000, warm_steps - done)
        with contextlib.redirect_stdout(io.StringIO()):
            s.simulate(n, sample_size=10 ** 15, method="tau", record_multievents=False)
        done += n
    warm_s = time.perf_counter() - t0
    occupied = int((m.infectious != 0).sum())
    s.set_mutation_rate(0.01)
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1, device=device)
    m.events.CreateEvents(timed_steps)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([seed], dtype=np.int64)); eng.stage_tau()
    o = _capi.VgxRunOpts(); o.record_events = 0
    t_wall = time.perf_counter()
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, timed_steps, 10 ** 15, -1.0, 1, C.byref(o)))
    t_wall = time.perf_counter() - t_wall
    c = eng.counters(0)
    n = max(int(c.loop_iterations), 1)
    ms = eng.last_kernel_ms
    out = {"start": "SURVEY.md 8(d): index case, %d tau steps at mutation rate 0.4 per site (%.1f s of wall time through the Simulator API), then "
                    "mutation rate 0.01" % (warm_steps, warm_s),
           "epidemic_time_at_start": float(m.currentTime), "infected_at_start": int(m.globalInfectious), "occupied_compartments": occupied,
           "occupied_haplotypes_per_population": occupied / float(m.popNum), "occupied_share": occupied / float(m.infectious.size),
           "steps": n, "ms_per_step": ms / n, "wall_ms_per_step": 1e3 * t_wall / n, "events_drawn": int(c.reserved[0]),
           "value": c.reserved[0] / (ms * 1e-3), "unit": "events/s (device time)",
           "roofline_frac": 16.0 * m.popNum * m.hapNum * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "note": "the drift pass and the column sums stream the dense [P][H] byte arrays whatever the occupancy; a try's scan and front pass "
                   "run over the lists of occupied compartments the drift pass writes (DESIGN.md 4.3d); 1/400 of the uniform fill's events"}
    eng.close()
    return out


def tau_leg(device, steps=20, per_cell=3, seed=2020, cpu=True, warm=True):
    """Tau-leaping on BASELINE config 4 (2^20 haplotypes x 256 populations, migration), dense ("spread")
    occupancy written straight into the model's arrays; the reference cannot even construct this shape
    (SURVEY.md §0.8).  Reports events drawn per second of device time and the step's HBM roofline against the
    fused-minimum traffic 16*P*H bytes/step."""
    import ctypes as C
    import numpy as np
    from vgsim_amd import Simulator, _capi
    sites, P = 10, 256
    with contextlib.redirect_stdout(io.StringIO()):
        s = Simulator(number_of_sites=sites, populations_number=P, seed=2020)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    m = s.simulation
    H = m.hapNum
    m.infectious[:] = per_cell
    m.susceptible[:, 0] -= per_cell * H
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1, device=device)
    m.events.CreateEvents(steps)
    m.events.ptr = 1            # not the first call of the model: capacity = ptr + iterations (events.pxi:61-68)
    m.events.CreateEvents(steps)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([seed], dtype=np.int64))
    o = _capi.VgxRunOpts(); o.record_events = 0
    t_stage = time.perf_counter()
    eng.stage_tau()                             # hand-over: the 2^28 compartments converted and resident in HBM before the timed call
    t_stage = time.perf_counter() - t_stage
    t_wall = time.perf_counter()
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, steps, 10 ** 15, -1.0, 1, C.byref(o)))
    t_wall = time.perf_counter() - t_wall       # the whole C-ABI call: its preparation, every step's launches and synchronisations
    c = eng.counters(0)
    ms = eng.last_kernel_ms
    n = max(int(c.loop_iterations), 1)
    drawn = int(c.reserved[0])
    # the same start state again in ONE call of 200 steps: the call's host-side preparation (snapshot, conversion and upload of
    # the 2^28 compartments) is paid once per call, so its share of the wall time shrinks with the length of the call
    long_steps = 200
    m.events.CreateEvents(long_steps)
    eng.set_state(m)
    eng.stage_tau()
    t_long = time.perf_counter()
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, long_steps, 10 ** 15, -1.0, 1, C.byref(o)))
    t_long = time.perf_counter() - t_long
    c2 = eng.counters(0)
    n2 = max(int(c2.loop_iterations), 1)
    long_call = {"steps": n2, "device_ms_per_step": eng.last_kernel_ms / n2, "wall_ms_per_step": 1e3 * t_long / n2,
                 "events_drawn": int(c2.reserved[0]), "value_wall": c2.reserved[0] / t_long, "unit": "events/s (wall time of the call)"}
    fused = 16.0 * P * H * n
    traffic = None
    try:   # HBM bytes per step from the committed PMC passes of this same leg (profiles/pmc_tau_c4.json)
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_tau_c4.json")))
        if d["config"] == {"steps": steps, "per_cell": per_cell}:
            traffic = d["hbm_bytes_per_step"]
    except Exception:
        pass
    out = {"workload": "BASELINE config 4: 1048576 haplotypes (10 sites) x 256 populations, total migration 0.01, "
                       "dense occupancy (%d infected per compartment: a uniform fill written into the model's arrays, not the "
                       "mutation warm-up of SURVEY.md 8(d)), Poisson tau-leaping" % per_cell,
           "steps": n, "ms_per_step": ms / n, "events_drawn": drawn,
           "value": drawn / (ms * 1e-3), "unit": "events/s (device time)",
           "wall": {"ms_per_step": 1e3 * t_wall / n, "value": drawn / t_wall, "unit": "events/s (wall time of the vgx_simulate_tau "
                    "call, start state resident in HBM: staged by vgx_stage_tau before the call)",
                    "staging_ms": 1e3 * t_stage, "ms_per_step_with_staging": 1e3 * (t_wall + t_stage) / n,
                    "note": "staging = snapshot, int64 -> int32 conversion and PCIe upload of the 2^28 compartments (host buffers -> HBM)"},
           "call_of_200_steps": long_call,
           "cpu_baseline": ({"note": "the reference cannot construct config 4 (its channel arrays would need > 1 TiB, SURVEY.md 0.8): the "
                                     "oracle's tau at the shapes SURVEY.md 8(d)(ii) names, same recipe, one core of this host; the engine's "
                                     "config-4 step covers %d compartment-channels" % (P * H * (2 + 3 * sites + 1 + (P - 1))),
                             "4096x8": tau_cpu_dense(6, 8), "256x8": tau_cpu_dense(4, 8)} if cpu else None),
           "roofline": {"bound": "hbm", "achieved": fused / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": fused / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                        "note": "all step kernels together (prep, column sums and the drift pass on the one-byte counts, sieve, {scan, events, "
                                "arrivals, verdict, decide} x tries, apply of the accepted try's list of moves + the one-byte copy); algorithmic "
                                "bytes = 16*P*H per step (read+write infectious once, one try); traffic = PMC bytes per step (about three "
                                "tries of the reference's halving loop per step)"}}
    eng.close()
    if warm:
        try:
            out["warmup_start"] = tau_warm_start(device, seed=seed)
        except Exception as ex:
            out["warmup_start"] = {"error": repr(ex)}
    return out


def config5_partition(world, rank, total=256, first_seed=2020):
    """BASELINE config 5's replicates on `world` ranks: 256 / world consecutive seeds per rank, from 2020 on."""
    import numpy as np
    R = max(total // world, 1)
    return R, first_seed + rank * R + np.arange(R, dtype=np.int64)


def config5_leg(device, world=1, rank=0, events=100000, traj_points=1001, ens=None):
    """BASELINE config 5 as written: 256 independent seeded replicates of config 3 (seeds 2020..2275) sharded over the GPUs of the
    node (256 / world per GPU; all 256 on the one GPU at N = 1), one RCCL gather of the f64 epidemic trajectories
    [replicates, 1001, 64, 2] to rank 0 (32.8 MB per GPU at 8 GPUs).  Runs on every rank."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from vgsim_amd.ensemble import Ensemble
    R, seeds = config5_partition(world, rank)
    cuda = ens is None          # (tests hand in an engine stand-in and run the leg's own partition / gather / reduction on CPU ranks)
    if ens is None:
        ens = Ensemble(make_simulator(2020), R, device=device)
    dev = "cuda" if cuda else "cpu"
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    res = None
    for it in range(2):      # the first launch is the warm-up
        sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        err = None
        try:
            res = ens.simulate(events, sample_size=10 ** 12, record_events=True, traj_points=traj_points, traj_window=(0.0, 12.0), seeds=seeds)
        except Exception as ex:   # every rank must learn of it BEFORE the gather: a rank alone in a collective hangs the others
            err = ex
        t_sim = time.perf_counter() - t0
        if world > 1:
            okv = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(okv, op=dist.ReduceOp.MIN)
            if okv.item() < 1.0:
                raise RuntimeError("config5: a rank failed in simulate (%r on this rank)" % (err,))
        elif err is not None:
            raise err
        t1 = time.perf_counter()
        out = ens.gather_trajectories(dst=0, device=dev)     # (one rank: the result stays on the GPU, where an RCCL gather leaves it on rank 0)
        sync()
        t_gather = time.perf_counter() - t1
    ev = float(res.total_events)
    tot, tmax = ev, t_sim + t_gather
    if world > 1:
        v = torch.tensor([ev], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        m = torch.tensor([t_sim + t_gather, t_gather], dtype=torch.float64, device=dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        tot, tmax, t_gather = float(v.item()), float(m[0].item()), float(m[1].item())
    shape = list(out.shape) if out is not None else None
    o = {"workload": "BASELINE config 5: 256 seeded replicates of config 3 (seeds 2020..2275), %d per GPU on %d GPU(s), %d events each, "
                     "f64 trajectories [%d, %d, %d, 2] gathered to rank 0" % (R, world, events, R, traj_points, POPS),
         "value": tot / tmax, "unit": "events/s (wall: simulate + gather, max over ranks)", "replicates_per_gpu": R,
         "kernel_ms_per_launch": res.kernel_ms, "simulate_s": t_sim, "gather_ms": 1e3 * t_gather,
         "gather_bytes_per_gpu": R * traj_points * POPS * 2 * 8, "gathered_shape_on_rank0": shape,
         "collective": "torch.distributed.gather (RCCL)" if world > 1 else "none (one rank: device-to-device copy into the result tensor)",
         "result_on": str(out.device) if out is not None else None,
         "kernel": getattr(getattr(ens, "engine", None), "last_kernel", None),
         "note": "256 replicates are 256 wavefronts: ONE GPU already runs them all concurrently (one per CU), so sharding them 32 per GPU "
                 "cannot shorten any of them: this leg is latency-bound per trajectory and flat in the GPU count; the weak-scaling "
                 "headline (16 384 replicates per GPU) is the curve that scales"}
    ens.close()
    return o


def tau_small_leg(device, cpu=True):
    """Tau-leaping on SMALL models (the regime users run for large epidemics with few haplotypes): the on-device step loop of
    vgx_taus.hip — one workgroup per replicate, no host round trip per step.  Steps per second of one trajectory, and of an ensemble."""
    import numpy as np
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    out = {"workload": "tau-leaping, small models after a 2000-event direct warm-up: b=2.5 d=0.9 s=0.1 m=0.05/site, total migration 0.002; "
                       "1000 steps per replicate, device time"}
    for name, sites, pops, size, reps in (("16x3", 2, 3, 10 ** 6, 2048), ("256x5", 4, 5, 10 ** 6, 512)):
        with contextlib.redirect_stdout(io.StringIO()):
            s = Simulator(number_of_sites=sites, populations_number=pops, seed=7)
        s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
        s.set_total_migration_probability(0.002); s.set_population_size(size)
        with contextlib.redirect_stdout(io.StringIO()):
            s.simulate(2000, sample_size=10 ** 12)
        cell = {}
        for key, R in (("single", 1), ("ensemble", reps)):
            ens = Ensemble(s, R, device=device)
            res = None
            for it in range(2):
                # (a replicate can run into upstream's own dead end — a compartment left below zero because the bounds check books
                # migrants on their source, pyx:2473 — which this engine reports as an error after 200 halvings)
                res = ens.simulate_tau(1000, sample_size=10 ** 15, seeds=7 + it * R + np.arange(R, dtype=np.int64))
            steps = float(res.loop_iterations.sum())
            cell[key] = {"replicates": R, "steps_per_s": steps / (res.kernel_ms * 1e-3), "events_per_s": float(res.events_drawn.sum()) / (res.kernel_ms * 1e-3),
                         "kernel_ms": res.kernel_ms}
            ens.close()
        if cpu:   # the oracle from the same kind of start state (its own 2000-event direct warm-up), one host core
            from oracle import oracle
            oracle.build()
            with contextlib.redirect_stdout(io.StringIO()):
                o = Simulator(number_of_sites=sites, populations_number=pops, seed=7)
            o.set_transmission_rate(2.5); o.set_recovery_rate(0.9); o.set_sampling_rate(0.1); o.set_mutation_rate(0.05)
            o.set_total_migration_probability(0.002); o.set_population_size(size)
            oracle.run_direct(o.simulation, 2000, 10 ** 12, -1, 200)
            cell["cpu_baseline"] = tau_cpu_time(o.simulation, 3.0, first=50)
        out[name] = cell
    return out


class BenchLoop:
    """The timed loop of one rank: `step(i)` runs one batch of replicates and hands that step's summary trajectories
    to an asynchronous gather to rank 0 (it overlaps the next step's kernel), `drain()` waits for the last gather, `reduce(elapsed, events)` gives the whole job's MAX time and SUM of events.  The engine is
    anything with `simulate(...)` / `gather_trajectories(...)` of `vgsim_amd.ensemble.Ensemble`: the world_size-2 gloo test
    drives this same class with a stand-in engine on CPU tensors (tests/test_bench_ranks_gloo.py)."""

    def __init__(self, ens, replicates, events, traj_points, world=1, rank=0, device="cuda", pops=POPS):
        self.ens, self.R, self.N, self.T = ens, replicates, events, traj_points
        self.world, self.rank, self.device, self.pops = world, rank, device, pops
        self.gather_out = None      # [world, R, T, P, 2] on rank 0, allocated once
        self.pending = None         # the gather of the previous step

    def seeds(self, i):
        """Distinct seeds per (step, rank, replicate): disjoint across ranks, independent of the GPU count."""
        import numpy as np
        return 2020 + (i * self.world + self.rank) * self.R + np.arange(self.R, dtype=np.int64)

    def step(self, i):
        res = self.ens.simulate(self.N, sample_size=10 ** 12, record_events=True, traj_points=self.T,
                                traj_window=(0.0, 12.0), seeds=self.seeds(i))
        if self.world > 1:
            import torch
            if self.pending is not None:
                self.pending.wait()
            if self.rank == 0 and self.gather_out is None:
                self.gather_out = torch.empty((self.world, self.R, self.T, self.pops, 2), dtype=torch.int32, device=self.device)
            # compartment totals are whole numbers below the population size (1e7): 32-bit on the wire, half the bytes per link
            self.pending = self.ens.gather_trajectories(dst=0, out=self.gather_out, async_op=True, wire_dtype=torch.int32)
        return res

    def drain(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

    def reduce(self, elapsed, events):
        """(max over ranks of the elapsed time, sum over ranks of the events)."""
        if self.world == 1:
            return float(elapsed), float(events)
        import torch
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e = torch.tensor([float(events)], dtype=torch.float64, device=self.device)
        dist.all_reduce(e, op=dist.ReduceOp.SUM)
        return float(t.item()), float(e.item())

    def per_rank(self, value):
        """`value` of every rank, in rank order (the JSON line lists each rank's events)."""
        if self.world == 1:
            return [float(value)]
        import torch
        import torch.distributed as dist
        mine = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        return [float(o.item()) for o in out]


def memory_plan(R, N, T, world, rank, free_bytes, total_bytes):
    """Device memory the headline leg of this rank needs (bytes, by item) — printed on stderr, and the run refused before anything is
    allocated when it cannot fit: occupancy lists (capped at 32 GiB, vgx_api.hip init_device_state), the event log resident in HBM
    (32 B per event), the f64 summary trajectories, their int32 wire copy and, on rank 0, the gathered result of all ranks."""
    P = POPS
    plan = {"occupancy_lists": min(32 * 2 ** 30, int(0.45 * free_bytes)), "event_log": R * N * 32, "trajectories_f64": R * T * P * 2 * 8,
            "wire_int32": (R * T * P * 2 * 4) if world > 1 else 0, "gathered_on_rank0": (world * R * T * P * 2 * 4) if (world > 1 and rank == 0) else 0,
            "population_blocks_and_scalars": R * (P * 9 * 8 + P * P * 8 + 512)}
    need = sum(plan.values())
    sys.stderr.write("bench.py rank %d/%d memory plan: %s = %.1f GB of %.1f GB free (%.1f GB device)\n" % (
        rank, world, ", ".join("%s %.1f GB" % (k, v / 1e9) for k, v in plan.items() if v), need / 1e9, free_bytes / 1e9, total_bytes / 1e9))
    if need > 0.92 * free_bytes:
        raise SystemExit("bench.py: rank %d needs %.1f GB of device memory for %d replicates x %d events (%s) but %.1f GB are free: "
                         "lower --replicates / --events / --traj-points" % (rank, need / 1e9, R, N, ", ".join(
                             "%s %.1f GB" % (k, v / 1e9) for k, v in plan.items() if v), free_bytes / 1e9))
    return plan


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (torch.distributed.run, one per GPU) as a CHILD
    process and exit with its code.  Nothing in this parent has touched HIP or imported torch, and nothing is re-exec'ed."""
    import subprocess
    # --standalone: the launcher picks a free rendezvous port itself (no bind-then-close race with other jobs on the node)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node=%d" % n, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank launch failed (exit code %d): %s\n" % (n, rc, " ".join(cmd)))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--replicates", type=int, default=16384, help="replicates per GPU and step (four per wavefront)")
    ap.add_argument("--events", type=int, default=100000, help="recorded events per replicate and step")
    ap.add_argument("--traj-points", type=int, default=1001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tau", action="store_true", help="skip the tau-leap (config 4) leg")
    ap.add_argument("--no-tau-warmup-start", action="store_true", help="profiling: leave out tau_leap.warmup_start (6000 tau steps of warm-up: a quarter of a million launches in a trace)")
    ap.add_argument("--no-extra", action="store_true", help="skip the FAST-mode, spread-occupancy and config-2 legs")
    ap.add_argument("--only", default="", help="development/profiling: run only this extra leg (fast_mode, spread_occupancy, "
                                               "spread_occupancy_fast, config2, genealogy, single_trajectory, direct_config4_shape, config3_general, table3, tau_small, config5, propensity_scan, tau_leap) and print its JSON")
    ap.add_argument("--table3-cells", default="", help="profiling: restrict the table3 leg to these cells, e.g. 10:0.001,100:0.1")
    a = ap.parse_args()
    cells3 = {(int(c.split(":")[0]), float(c.split(":")[1])) for c in a.table3_cells.split(",") if c} or None

    # ---- ranks: one process per GPU.  Under a launcher (torch.distributed.run sets WORLD_SIZE) this process is one rank;
    # without one, `--gpus N > 1` starts N fresh ranks as a child process BEFORE anything here imports torch or touches HIP.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(env_world or "1")
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; run `python bench.py --gpus %d` (it starts the ranks itself) or "
                 "`python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ...`"
                 % (a.gpus, world, a.gpus, a.gpus, a.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but this node shows %d device(s)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from vgsim_amd.ensemble import Ensemble
    R, N = a.replicates, a.events
    H = 4 ** SITES
    if not a.only:
        free_b, total_b = torch.cuda.mem_get_info(local)
        memory_plan(R, N, a.traj_points, world, rank, free_b, total_b)
    extra_legs = (("fast_mode", lambda d: fast_leg(d, R, N, a.traj_points)),
                  ("spread_occupancy", lambda d: spread_leg(d, "exact", replicates=8192, events=10000)),
                  ("spread_occupancy_fast", lambda d: spread_leg(d, "fast", replicates=12288, events=10000)),
                  ("config2", c2_leg), ("genealogy", genealogy_leg), ("single_trajectory", single_leg),
                  ("direct_config4_shape", c4_direct_leg), ("config3_general", general_c3_leg), ("table3", lambda d: table3_leg(d, cpu=not a.no_cpu_baseline, cells=cells3)),
                  ("tau_small", lambda d: tau_small_leg(d, cpu=not a.no_cpu_baseline)), ("propensity_scan", rowscan_leg),
                  ("tau_leap", lambda d: tau_leg(d, cpu=not a.no_cpu_baseline, warm=not a.no_tau_warmup_start)))
    if a.only:
        legs = dict(extra_legs)
        legs["config5"] = lambda d: config5_leg(d, world=world, rank=rank)
        print(json.dumps({a.only: legs[a.only](local)}), flush=True)
        return
    sim = make_simulator(2020)
    ens = Ensemble(sim, a.replicates, device=local)
    loop = BenchLoop(ens, R, N, a.traj_points, world=world, rank=rank, device="cuda")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        loop.step(i)
    loop.drain()
    sync()
    t0 = time.perf_counter()
    events = 0
    kernel_ms = 0.0
    for i in range(a.steps):
        res = loop.step(a.warmup + i)
        events += res.total_events
        kernel_ms += res.kernel_ms
    loop.drain()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed, total_events = loop.reduce(elapsed, events)
    events_per_rank = loop.per_rank(events)

    if rank == 0:
        value = total_events / elapsed
        # ---- roofline of the dominant kernel (vgx_direct_kernel), this rank ----
        # Algorithmic bytes per recorded event of THIS engine's layout: the chosen population's
        # occupancy-list stream (16 B/entry, read once and kept in registers), the migration row (8P), the
        # event record (32 B) and the count write-back (8 B).  Mean list length measured from final state.
        # The list an event works on is the list of ITS population, and events fall where the infected are: the index-case
        # population holds most of them AND the longest list (the mean over the populations understates the bytes an event
        # touches several-fold).  Event-weighted list length from the final states of a sample of replicates: population p gets
        # weight totalInfectious[p] (the share of the infection-driven events it draws).
        nocc_mean, nocc_w, wsum = 0.0, 0.0, 0.0
        sample = list(range(0, R, max(R // 16, 1)))[:16]
        for r in sample:
            st = ens.replicate_state(r)
            occ = (st.infectious != 0).sum(axis=1).astype(float)
            w = st.infectious.sum(axis=1).astype(float)
            nocc_mean += occ.mean() / len(sample)
            nocc_w += float((occ * w).sum())
            wsum += float(w.sum())
        nocc_w = nocc_w / max(wsum, 1.0)
        # short-list form of the row kernel: 16 B per entry (haplotype, class, count) of the event's list read once; lists beyond a
        # tile stream the 4-byte counts (refresh: the whole list; choice: on average half of it) + haplotypes of the hit tile
        list_bytes = 16.0 * nocc_w if nocc_w <= 64 else 6.0 * nocc_w + 4.0 * 64
        bytes_per_event = max(list_bytes, 16.0) + 8.0 * POPS + 32.0 + 8.0
        ev_per_launch = events / max(a.steps, 1)
        launch_s = (kernel_ms / max(a.steps, 1)) * 1e-3
        achieved = ev_per_launch * bytes_per_event / launch_s / 1e9
        dense_bytes = H * (84 + 16 * SUS) + 16 * POPS + 48   # the reference's dense layout
        traffic = pmc_traffic("headline", {"replicates_per_gpu": R, "events_per_replicate": N, "trajectory_points": a.traj_points})
        li = float(res.loop_iterations.sum()) / max(float(res.events.sum()), 1.0)
        line = {
            # BASELINE.json's metric string; `value` is its direct-Gillespie leg on config 3, the tau-leap leg (config 4) is the
            # `tau_leap` object of the same line, the Cython-equivalent CPU path `cpu_baseline`
            "metric": "simulated events/sec (direct + tau-leap) at 1/2/4/8 MI355X vs Cython CPU", "value": value, "unit": "events/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / max(a.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "libm_probe_match": libm_probe_match(),
            "config": {"value_leg": "direct Gillespie (BASELINE config 3); tau-leap (config 4) in tau_leap",
                       "workload": "BASELINE config 3: 65536 haplotypes (8 sites) x 64 populations x 1 susceptibility "
                                   "group, direct Gillespie, bit-exact mode (PCG64 stream, reference summation order), "
                                   "index-case start, b=2.5 d=0.9 s=0.1 m=0.01/site, total migration 0.01, N=1e7",
                       "replicates_per_gpu": R, "events_per_replicate": N, "parallelism": "replicates x %d (one process per GPU, %d replicates each, no data-path collective)" % (world, R),
                       "events_per_rank": events_per_rank,
                       "trajectory_points": a.traj_points, "loop_iterations_per_event": li},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # what the memory system really moved (PMC bytes per launch / launch time / peak): `frac` above prices the
                         # ALGORITHMIC list bytes, most of which are served by L2 / Infinity Cache — it is not HBM utilisation
                         "hbm_frac": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "kernel": "vgx_quad_kernel" if R >= 2048 else "vgx_direct_kernel_p64s1c1",
                         "kernel_ms_per_launch": kernel_ms / max(a.steps, 1),
                         "bytes_per_event": bytes_per_event, "mean_occupancy_list_len": nocc_mean,
                         "event_weighted_list_len": nocc_w,
                         "note": "persistent sequential event loop: latency/issue-bound, not bandwidth-bound (chain_bound below); "
                                 "bytes_per_event from the EVENT-WEIGHTED list length (final states of 16 replicates, population weight "
                                 "= its infected hosts); the reference's dense layout would need %.3g B/event = %.3g GB/s at this event rate"
                                 % (dense_bytes, ev_per_launch * dense_bytes / launch_s / 1e9)},
            # the binding resource: dependent f64 additions in the reference's order.  Per event: the refresh of the event's list
            # (pyx:519-528), on average half of it for the choice (fast_choose.pxi:22-25), the popRate and migPopRate totals
            # (pyx:537-546); one v_fmac_f64 (DPP) per term, each instruction serving the four replicates of a wavefront
            "chain_bound": {"bound": "dependent f64 additions in the reference's order (one v_fmac_f64 per term)",
                            "steps_per_event": 1.5 * nocc_w + 2.0 * POPS,
                            "achieved": ev_per_launch * (1.5 * nocc_w + 2.0 * POPS) / launch_s, "peak": CHAIN_STEPS_PER_S * 4,
                            "unit": "chain steps/s", "frac": ev_per_launch * (1.5 * nocc_w + 2.0 * POPS) / launch_s / (CHAIN_STEPS_PER_S * 4),
                            "chains_per_instruction": 4},
        }
        # event log: resident in HBM per replicate (28 B/event); D2H through vgx_get_events on a sample
        t_d = time.perf_counter()
        nrep = min(R, 32)
        got = 0
        for r in range(nrep):
            got += ens.replicate_events(r).shape[1]
        t_d = time.perf_counter() - t_d
        line["event_log"] = {"device_bytes_per_event": 32, "d2h_sample_replicates": nrep, "d2h_events": got,
                             "d2h_s": t_d, "note": "copy-out widens to the reference's (6,N) float64 layout on the host and rebuilds "
                                                    "the event times with the host libm (PCG64 stream + logged rate denominators)"}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
    ens.close()
    ens = None
    loop.gather_out = None      # rank 0's [world, R, T, P, 2] result: release it before the other legs allocate
    loop = None
    torch.cuda.empty_cache()
    # tau-leaping leg (BASELINE config 4): one independent replicate per GPU (replicas only, no collective on the path)
    tau = None
    if not a.no_tau:
        try:
            tau = tau_leg(local, seed=2020 + rank, cpu=(rank == 0 and world == 1 and not a.no_cpu_baseline), warm=(world == 1 and not a.no_tau_warmup_start))
        except Exception as ex:  # never lose the headline line
            tau = {"error": repr(ex)}
        if world > 1:
            ok = 0.0 if "error" in tau else 1.0
            v = torch.tensor([ok, tau.get("events_drawn", 0.0) * ok, tau.get("ms_per_step", 0.0) * tau.get("steps", 0) * ok],
                             dtype=torch.float64, device="cuda")
            tot = v.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            mx = v.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            if rank == 0 and "error" not in tau:
                tau["ranks_ok"] = int(tot[0].item())
                tau["events_drawn"] = int(tot[1].item())
                tau["value"] = tot[1].item() / max(mx[2].item() * 1e-3, 1e-12)   # all ranks' events / slowest rank's device time
                tau["unit"] = "events/s (device time, %d replicas)" % world
                tau.pop("roofline", None)
    c5 = None
    if not a.no_extra:
        try:
            c5 = config5_leg(local, world=world, rank=rank)
        except Exception as ex:
            c5 = {"error": repr(ex)}
            if world > 1:
                raise      # a rank that leaves a collective alone would hang the others: fail the run visibly
    if rank == 0:
        if c5 is not None:
            line["config5"] = c5
        if tau is not None:
            line["tau_leap"] = tau
        if world == 1 and not a.no_extra:
            for name, fn in extra_legs[:-1]:
                try:
                    line[name] = fn(local)
                except Exception as ex:
                    line[name] = {"error": repr(ex)}
        # the numbers of the other legs once more, short, as the LAST object of the line (a reader that keeps only the tail of a long
        # line still gets them)
        def pick(d, *path):
            for k in path:
                if not isinstance(d, dict) or k not in d:
                    return None
                d = d[k]
            return d
        line["summary"] = {
            "headline_events_per_s": line["value"], "headline_roofline_frac": pick(line, "roofline", "frac"),
            "headline_hbm_frac": pick(line, "roofline", "hbm_frac"),
            "headline_chain_frac": pick(line, "chain_bound", "frac"),
            "tau_leap_ms_per_step": pick(line, "tau_leap", "ms_per_step"), "tau_leap_events_per_s": pick(line, "tau_leap", "value"),
            "tau_leap_roofline_frac": pick(line, "tau_leap", "roofline", "frac"), "tau_leap_wall_ms_per_step": pick(line, "tau_leap", "wall", "ms_per_step"),
            "tau_cpu_ns_per_channel_step": pick(line, "tau_leap", "cpu_baseline", "4096x8", "ns_per_channel_step"),
            "tau_warmup_start_ms_per_step": pick(line, "tau_leap", "warmup_start", "ms_per_step"),
            "config5_events_per_s": pick(line, "config5", "value"), "fast_mode_events_per_s": pick(line, "fast_mode", "value"),
            "single_trajectory_config2": pick(line, "single_trajectory", "config2"),
            "single_trajectory_config3": pick(line, "single_trajectory", "config3"),
            "single_trajectory_config3_general": pick(line, "single_trajectory", "config3_general"),
            "table3_K2_single_events_per_s": pick(line, "table3", "cells", "K=2,M=0.001", "single_trajectory", "events_per_s"),
            "table3_K2_single_vs_published": pick(line, "table3", "cells", "K=2,M=0.001", "single_trajectory", "vs_baseline"),
            "table3_K2_single_wall_vs_published": pick(line, "table3", "cells", "K=2,M=0.001", "single_trajectory", "wall", "vs_baseline"),
            "table3_K100_single_events_per_s": pick(line, "table3", "cells", "K=100,M=0.001", "single_trajectory", "events_per_s"),
            "table3_K100_single_wall_vs_published": pick(line, "table3", "cells", "K=100,M=0.001", "single_trajectory", "wall", "vs_baseline"),
            "table3_K2_ensemble_events_per_s": pick(line, "table3", "cells", "K=2,M=0.001", "ensemble", "events_per_s"),
            "table3_K10_ensemble_events_per_s": pick(line, "table3", "cells", "K=10,M=0.001", "ensemble", "events_per_s"),
            "table3_K100_ensemble_events_per_s": pick(line, "table3", "cells", "K=100,M=0.001", "ensemble", "events_per_s"),
            "table3_K10_ensemble_philox_events_per_s": pick(line, "table3", "cells", "K=10,M=0.001", "ensemble_philox", "events_per_s"),
            "table3_K100_ensemble_philox_events_per_s": pick(line, "table3", "cells", "K=100,M=0.001", "ensemble_philox", "events_per_s"),
            "spread_occupancy_events_per_s": pick(line, "spread_occupancy", "value"),
            "config3_general_events_per_s": pick(line, "config3_general", "value"),
            "propensity_scan_roofline_frac": pick(line, "propensity_scan", "roofline", "frac"),
            "cpu_baseline_events_per_s": pick(line, "cpu_baseline", "value"),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
