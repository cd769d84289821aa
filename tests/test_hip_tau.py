"""Tau-leaping on the GPU (vgx_tau.hip) against the CPU oracle.  The device draws its Poisson numbers from
per-compartment Philox streams, the reference from one sequential PCG64 stream, so the comparison is
distributional, as BASELINE.json asks ("within a stated distributional tolerance for tau-leaping"):
  * tau selection is deterministic given the state: the first accepted leap must have the oracle's length
    (relative tolerance 1e-9; summation order differs);
  * over N_SEEDS seeded runs (identical bit-exact direct warm-up per seed, then tau steps) the means of cumulative
    infections, recoveries, samples, mutations, migrations, the epidemic time and the final infectious total agree
    within 4.5 standard errors of the paired difference (absolute floor: half an event / 1e-9 time units);
  * from ONE common warm-up state, N_ENSEMBLE seeded tau runs per engine (the device's as one ensemble launch): mean,
    variance and the two quartiles of every counter, of the epidemic time, of the infectious total of every population
    and of every haplotype agree within 4.5 standard errors of the respective estimator (stated in the test);
  * bookkeeping invariants hold exactly on every run (compartment sums, counters vs multievent rows)."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu
N_SEEDS = 24


def run_tau_case(name, seed, engine, oracle=None):
    from vgsim_amd import Simulator
    ctor, phases = models.tau_case(name)
    ctor = dict(ctor, seed=seed)
    with helpers.quiet():
        sim = Simulator(**ctor)
        m = sim.simulation
        for setup, kw in phases:
            setup(sim)
            kw = dict(kw)
            if engine == "hip":
                sim.simulate(**kw)
            else:
                it = kw.pop("iterations")
                ss = kw.pop("sample_size", None)
                ss = it if ss is None else ss
                method = kw.pop("method", "direct")
                if method == "direct":
                    assert oracle.run_direct(m, it, ss, -1, 200) == 0
                else:
                    assert oracle.run_tau(m, it, ss, -1, 200) == 0
    return sim.simulation


@pytest.mark.parametrize("step_kernels", [True, False])
@pytest.mark.parametrize("name", ["tau_a", "tau_b", "tau_c", "tau_many_classes", "tau_wide_table"])
def test_first_leap_length_matches_oracle(oracle_mod, name, step_kernels, monkeypatch):
    """ChooseTau (pyx:2432-2450) after a bit-exact direct warm-up: the first leap's length against the oracle's.  On the step kernels
    (VGX_TAU_STEP_KERNELS=1) the first try of these cases is accepted on both sides: the leap IS the chosen tau, compared to 1e-9.
    The on-device loop of small models (vgx_taus.hip) draws from other streams and may halve where the oracle does not (pyx:2316-2321):
    there both sides report the step's rejected tries, and leap * 2^tries — the tau ChooseTau gave the step, whatever the halving
    loop then made of it — must agree to 1e-9 (a wrong starting try of the loop on the device would show here)."""
    if step_kernels:
        monkeypatch.setenv("VGX_TAU_STEP_KERNELS", "1")
    ctor, phases = models.tau_case(name)
    hip = run_tau_case(name, ctor["seed"], "hip")
    ref = run_tau_case(name, ctor["seed"], "oracle", oracle_mod)
    nd = phases[0][1]["iterations"]
    assert np.array_equal(hip.events.as_array()[:, :nd], ref.events.as_array()[:, :nd])  # direct warm-up: bit-exact
    assert hip.events.types[nd] == 6 and ref.events.types[nd] == 6
    dt_hip = hip.events.times[nd] - hip.events.times[nd - 1]
    dt_ref = ref.events.times[nd] - ref.events.times[nd - 1]
    if step_kernels:
        assert dt_hip == pytest.approx(dt_ref, rel=1e-9)
        return
    tries_hip = int(hip._engine.tau_tries(0, 0, 1)[0])
    tries_ref = oracle_mod.tau_tries(0)
    assert tries_ref >= 0 and 0 <= tries_hip <= 200
    assert dt_hip * 2.0 ** tries_hip == pytest.approx(dt_ref * 2.0 ** tries_ref, rel=1e-9), (dt_hip, tries_hip, dt_ref, tries_ref)


@pytest.mark.parametrize("name", ["tau_a", "tau_b", "tau_c", "tau_d", "tau_many_classes", "tau_wide_table", "tau_d:large", "tau_c:large",
                                  "tau_b:steps", "tau_many_classes:steps"])
def test_tau_moments_match_oracle(oracle_mod, name, monkeypatch):
    # Small models run the on-device step loop (vgx_taus.hip).  ":steps": the step kernels of vgx_tau.hip on the same models
    # (VGX_TAU_STEP_KERNELS=1); ":large": those kernels with the draw thresholds of large models (a compartment's events drawn
    # kind by kind between means of 16 and 64, channel by channel only from 64 on)
    if name.endswith(":steps"):
        name = name[:-6]
        monkeypatch.setenv("VGX_TAU_STEP_KERNELS", "1")
    if name.endswith(":large"):
        name = name[:-6]
        monkeypatch.setenv("VGX_TAU_LARGE_MODEL_THRESHOLDS", "1")
    ctor, _ = models.tau_case(name)
    keys = ("bCounter", "dCounter", "sCounter", "mCounter", "migPlus", "currentTime")
    diffs = {k: [] for k in keys + ("infected",)}
    means = {k: [] for k in keys + ("infected",)}
    for i in range(N_SEEDS):
        seed = ctor["seed"] + 1000 + i
        hip = run_tau_case(name, seed, "hip")
        ref = run_tau_case(name, seed, "oracle", oracle_mod)
        for k in keys:
            diffs[k].append(float(getattr(hip, k)) - float(getattr(ref, k)))
            means[k].append(float(getattr(ref, k)))
        diffs["infected"].append(float(hip.infectious.sum() - ref.infectious.sum()))
        means["infected"].append(float(ref.infectious.sum()))
        # exact invariants on the device result
        P = hip.popNum
        assert (hip.susceptible >= 0).all() and (hip.infectious >= 0).all()
        assert hip.globalInfectious == hip.infectious.sum()
        assert np.array_equal(hip.totalInfectious, hip.infectious.sum(axis=1))
        assert hip.events.ptr == ref.events.ptr or hip.globalInfectious == 0 or ref.globalInfectious == 0
        total_hosts = hip.susceptible.sum() + hip.infectious.sum()
        assert total_hosts == int(hip.sizes.sum()), "hosts are conserved by every channel"
    for k, d in diffs.items():
        d = np.asarray(d)
        se = d.std(ddof=1) / np.sqrt(len(d)) if len(d) > 1 else 0.0
        tol = 4.5 * se + (1e-9 if k == "currentTime" else 0.5)
        assert abs(d.mean()) <= tol, "%s: mean difference %.4g exceeds %.4g (se %.4g, ref mean %.4g)" % (
            k, d.mean(), tol, se, np.mean(means[k]))


N_ENSEMBLE = 512


@pytest.mark.parametrize("name", ["tau_a", "tau_b", "tau_c", "tau_d", "tau_c:large", "tau_d:large", "tau_b:steps", "tau_c:steps"])
def test_tau_distribution_matches_oracle_many_seeds(oracle_mod, name, monkeypatch):
    """One common start state (the case's direct warm-up, bit-exact on both engines), then N_ENSEMBLE tau runs that differ
    by their seed only: the device's in one ensemble launch, the oracle's one after the other.  Two independent samples
    of the same law: for every scalar quantity q, with n = N_ENSEMBLE per sample and s the pooled standard deviation,
      |mean_1 - mean_2|         <= 4.5 * sqrt(s1^2/n + s2^2/n)                      + floor
      |var_1 - var_2|           <= 4.5 * sqrt((m4_1 - s1^4)/n + (m4_2 - s2^4)/n)    + floor
      |quartile_1 - quartile_2| <= 4.5 * sqrt(2) * 1.36 * s / sqrt(n)               + 1     (integers: one count)
    (1.36 s / sqrt(n) = the standard error of a quartile of a near-normal sample); floor = 0.5 events, 1e-9 time units."""
    if name.endswith(":steps"):   # the step kernels on the small models (see test_tau_moments_match_oracle)
        name = name[:-6]
        monkeypatch.setenv("VGX_TAU_STEP_KERNELS", "1")
    if name.endswith(":large"):   # ... with the draw thresholds of large models
        name = name[:-6]
        monkeypatch.setenv("VGX_TAU_LARGE_MODEL_THRESHOLDS", "1")
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    ctor, phases = models.CASES[name]
    n = N_ENSEMBLE
    seeds = 50000 + np.arange(n, dtype=np.int64)
    nt = dict(phases[1][1])["iterations"]

    def warm(engine):
        with helpers.quiet():
            sim = Simulator(**ctor)
            phases[0][0](sim)
            kw = dict(phases[0][1])
            if engine == "hip":
                sim.simulate(**kw)
            else:
                assert oracle_mod.run_direct(sim.simulation, kw["iterations"], kw["iterations"], -1, 200) == 0
        return sim

    keys = ("bCounter", "dCounter", "sCounter", "mCounter", "iCounter", "migPlus", "currentTime")

    def features(m):
        return [float(getattr(m, k)) for k in keys] + [float(v) for v in m.infectious.sum(axis=1)] + \
               [float(v) for v in m.infectious.sum(axis=0)]

    base = warm("hip")
    ens = Ensemble(base, n, seeds=seeds)
    ens.simulate_tau(nt, sample_size=10 ** 12)
    dev = np.array([features(ens.replicate_state(r)) for r in range(n)])
    ens.close()
    ref = []
    for sd in seeds:
        one = warm("oracle").simulation
        one.user_seed = int(sd)
        assert oracle_mod.run_tau(one, nt, 10 ** 12, -1, 200) == 0
        ref.append(features(one))
    ref = np.array(ref)
    ref0 = warm("oracle").simulation
    assert np.array_equal(base.simulation.infectious, ref0.infectious)       # the common start state is the same state
    names = list(keys) + ["infectious[pop %d]" % i for i in range(ref0.popNum)] + ["infectious[hap %d]" % i for i in range(ref0.hapNum)]
    for j, what in enumerate(names):
        a, b = dev[:, j], ref[:, j]
        floor = 1e-9 if what == "currentTime" else 0.5
        s1, s2 = a.std(ddof=1), b.std(ddof=1)
        tol = 4.5 * np.sqrt(s1 ** 2 / n + s2 ** 2 / n) + floor
        assert abs(a.mean() - b.mean()) <= tol, "%s: means %.6g vs %.6g (tolerance %.3g)" % (what, a.mean(), b.mean(), tol)
        m4a, m4b = ((a - a.mean()) ** 4).mean(), ((b - b.mean()) ** 4).mean()
        tolv = 4.5 * np.sqrt(max(m4a - s1 ** 4, 0.0) / n + max(m4b - s2 ** 4, 0.0) / n) + floor
        assert abs(s1 ** 2 - s2 ** 2) <= tolv, "%s: variances %.6g vs %.6g (tolerance %.3g)" % (what, s1 ** 2, s2 ** 2, tolv)
        sp = np.sqrt(0.5 * (s1 ** 2 + s2 ** 2))
        tolq = 4.5 * np.sqrt(2.0) * 1.36 * sp / np.sqrt(n) + (1e-9 if what == "currentTime" else 1.0)
        for q in (0.25, 0.75):
            qa, qb = np.quantile(a, q), np.quantile(b, q)
            assert abs(qa - qb) <= tolq, "%s: %d %% quantiles %.6g vs %.6g (tolerance %.3g)" % (what, int(100 * q), qa, qb, tolq)


@pytest.mark.parametrize("case", ["tau_b", "tau_c", "tau_b:steps", "tau_c:large", "tau_d:large"])
def test_multievent_rows_account_for_counters(case, monkeypatch):
    """Sparse multievent log: rows with num > 0 only; their sums reproduce the counter increments and every
    MULTITYPE record points at its own [start, end) row range (":large": with the draw thresholds of large models, i.e.
    through the one-draw-per-kind form and the channel-by-channel kernel from a mean of 64 on)."""
    if case.endswith(":steps"):
        case = case[:-6]
        monkeypatch.setenv("VGX_TAU_STEP_KERNELS", "1")
    if case.endswith(":large"):
        case = case[:-6]
        monkeypatch.setenv("VGX_TAU_LARGE_MODEL_THRESHOLDS", "1")
    ctor, phases = models.CASES[case]
    hip = run_tau_case(case, ctor["seed"], "hip")
    nd = phases[0][1]["iterations"]
    me = hip.multievents
    assert me.ptr > 0 and (me.num[:me.ptr] > 0).all()
    starts, ends = hip.events.haplotypes[nd:hip.events.ptr], hip.events.populations[nd:hip.events.ptr]
    assert starts[0] == 0 and ends[-1] == me.ptr and (starts[1:] == ends[:-1]).all()
    by_type = np.bincount(me.types[:me.ptr], weights=me.num[:me.ptr], minlength=6)
    direct = np.bincount(hip.events.types[:nd], minlength=7)
    assert hip.bCounter == direct[0] + by_type[0]
    assert hip.dCounter == direct[1] + by_type[1]
    assert hip.sCounter == direct[2] + by_type[2]
    assert hip.mCounter == direct[3] + by_type[3]
    assert hip.iCounter == direct[4] + by_type[4]
    assert hip.migPlus == direct[5] + by_type[5]


def test_tau_ensemble_replicates_equal_single_runs():
    """Replicate r of a tau ensemble is the run a single engine makes with that seed: the Philox streams are keyed
    by (seed, attempt, compartment, step, retry) and all accumulation is integer, so equality is exact."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    ctor, phases = models.CASES["tau_b"]
    seeds = np.array([7, 8, 1234, 99], dtype=np.int64)

    def warm(seed):
        with helpers.quiet():
            sim = Simulator(**dict(ctor, seed=int(seed)))
            phases[0][0](sim)
        return sim

    # common start state: a direct warm-up with seed 7, then tau with different seeds
    base = warm(7)
    with helpers.quiet():
        base.simulate(2000)
    ens = Ensemble(base, len(seeds), seeds=seeds)
    res = ens.simulate_tau(60, sample_size=10 ** 12)
    for r, seed in enumerate(seeds):
        one = warm(7)
        with helpers.quiet():
            one.simulate(2000)
            one.simulation.user_seed = int(seed)      # same start state, replicate's seed for the tau phase
            one.simulate(60, sample_size=10 ** 12, method="tau")
        st = ens.replicate_state(r)
        m = one.simulation
        assert np.array_equal(st.infectious, m.infectious) and np.array_equal(st.susceptible, m.susceptible), r
        assert (st.bCounter, st.dCounter, st.sCounter, st.mCounter, st.iCounter, st.migPlus) == \
               (m.bCounter, m.dCounter, m.sCounter, m.mCounter, m.iCounter, m.migPlus)
        assert st.currentTime == m.currentTime and res.events[r] == m.events.ptr
    assert len({int(ens.replicate_state(r).bCounter) for r in range(len(seeds))}) > 1   # the seeds really differ
    ens.close()


def test_halving_sieve_leaves_the_accepted_steps_untouched():
    """Many sparsely filled compartments: the first tries of every step are certain rejections.  The sieve
    (vgx_tau_sieve_kernel) starts the halving loop later; since a try's random streams are keyed by its index, the
    steps that get accepted are bit for bit those of the full loop (vgx_run_opts.reserved[0] = 1)."""
    import ctypes as C
    from vgsim_amd import Simulator, _capi

    def run(full_loop):
        with helpers.quiet():
            s = Simulator(number_of_sites=9, populations_number=16, seed=31)
        s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
        s.set_total_migration_probability(0.01); s.set_population_size(10 ** 8)
        m = s.simulation
        m.infectious[:] = 3
        m.susceptible[:, 0] -= 3 * m.hapNum
        m.totalInfectious[:] = 3 * m.hapNum
        m.totalSusceptible[:] = m.susceptible.sum(axis=1)
        m.globalInfectious = int(m.totalInfectious.sum())
        m.first_simulation = True
        m.initial_infectious[:] = m.infectious
        m.initial_susceptible[:] = m.susceptible
        eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
        m.events.CreateEvents(12)
        m.events.ptr = 1
        m.events.CreateEvents(12)
        eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([31], dtype=np.int64))
        o = _capi.VgxRunOpts(); o.record_events = 0      # hundreds of thousands of multievent rows per step
        o.reserved[0] = 1 if full_loop else 0
        eng._check(eng.lib.vgx_simulate_tau(eng.handle, 12, 10 ** 15, -1.0, 1, C.byref(o)))
        eng.get_state(m, 0)
        c = eng.counters(0)
        out = (m, int(c.reserved[3]), int(c.reserved[0]), int(c.ev_ptr))
        eng.close()
        return out
    a, skipped_a, drawn_a, ptr_a = run(True)
    b, skipped_b, drawn_b, ptr_b = run(False)
    assert skipped_a == 0 and skipped_b >= 3           # certain rejections left out
    assert ptr_a == ptr_b == 13 and drawn_a == drawn_b > 0
    assert a.currentTime == b.currentTime > 0
    assert np.array_equal(a.infectious, b.infectious) and np.array_equal(a.susceptible, b.susceptible)
    assert not np.array_equal(a.infectious, a.initial_infectious)
    for k in a.COUNTERS:
        assert getattr(a, k) == getattr(b, k), k


@pytest.mark.parametrize("case", ["sparse_large", "tiny_full", "large_compartments", "one_mutant_rescues", "small_next_to_large", "classes_short_rows",
                                  "queue_grows", "many_classes", "wide_migration_table"])
def test_sparse_try_equals_the_dense_passes(case, monkeypatch):
    """(Step kernels of vgx_tau.hip, forced here also for the small shapes that the on-device step loop would take.)
    A try of the halving loop keeps its deltas as a list of moves and checks the bounds of GenerateEvents_tau
    (pyx:2522-2528) where the deltas are drawn (own deltas at once, compartments found below zero against the mutants of
    their neighbours, the upper bound per population).  ``vgx_run_opts.reserved[1] = 2`` writes both dense delta arrays in
    every try with the checks fused into the draw and scatter kernels, ``= 1`` additionally runs the check as one dense pass
    over all compartments.  Same draws, same decisions, hence bit for bit the same accepted steps in all three — on a large
    sparse state (thousands of rejected tries' worth of failing compartments), on a tiny population that sits at its upper
    bound (arrivals push compartments over ``sizes``: the sparse mode hands those tries to the dense one), with compartments
    that draw every channel on its own (vgx_tau_draw_big_kernel), with a mutation rate so high that compartments below
    zero on their own are regularly rescued by arriving mutants, and with single hosts next to compartments of 20 000 (their
    rescue depends on neighbours that are drawn channel by channel: the hash table of vgx_tau_arrivals_kernel decides)."""
    import ctypes as C
    from vgsim_amd import Simulator, _capi
    monkeypatch.setenv("VGX_TAU_STEP_KERNELS", "1")

    def run(mode):
        with helpers.quiet():
            if case == "sparse_large":
                s = Simulator(number_of_sites=9, populations_number=16, seed=31)
                s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
                s.set_total_migration_probability(0.01); s.set_population_size(10 ** 8)
                fill, steps = 2, 10
            elif case == "large_compartments":
                s = Simulator(number_of_sites=3, populations_number=4, seed=9)
                s.set_transmission_rate(3.0); s.set_recovery_rate(1.0); s.set_sampling_rate(0.2); s.set_mutation_rate(0.02)
                s.set_total_migration_probability(0.05); s.set_population_size(10 ** 7)
                fill, steps = 20000, 30
            elif case == "small_next_to_large":
                s = Simulator(number_of_sites=6, populations_number=3, seed=17)
                s.set_transmission_rate(5.2); s.set_recovery_rate(4.5); s.set_sampling_rate(0.5); s.set_mutation_rate(0.002)
                s.set_total_migration_probability(0.02); s.set_population_size(10 ** 9)
                fill, steps = 20000, 12
            elif case == "classes_short_rows":
                # several rate classes, rows shorter than one wave tile of the scan kernel (lanes beyond the row), two groups
                s = Simulator(number_of_sites=3, populations_number=5, number_of_susceptible_groups=2, seed=21)
                s.set_transmission_rate(2.5); s.set_recovery_rate(0.8); s.set_sampling_rate(0.2); s.set_mutation_rate(0.2)
                s.set_transmission_rate(4.0, haplotype=5); s.set_recovery_rate(0.3, haplotype=17); s.set_transmission_rate(1.0, haplotype=40)
                s.set_susceptibility(0.4, susceptibility_type=1); s.set_immunity_transition(0.05, source=1, target=0)
                s.set_susceptibility_type(1)
                s.set_total_migration_probability(0.1); s.set_population_size(10 ** 5)
                fill, steps = 30, 40
            elif case == "queue_grows":
                # 4096 haplotypes with 50 hosts each: most compartments draw events in every try, far more than a shard of the
                # queue holds at first (an eighth of its compartments): the try is run again with a larger queue
                s = Simulator(number_of_sites=6, populations_number=2, seed=23)
                s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.03)
                s.set_total_migration_probability(0.02); s.set_population_size(10 ** 7)
                fill, steps = 50, 12
            elif case == "many_classes":
                # 300 haplotypes with a recovery rate of their own: more than 256 rate classes, the class tables of the events
                # kernel stay in global memory and the scan / drift kernels take their general forms
                s = Simulator(number_of_sites=5, populations_number=3, seed=29)
                s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
                for hn in range(300):
                    s.set_recovery_rate(0.5 + 0.002 * hn, haplotype=hn)
                s.set_total_migration_probability(0.02); s.set_population_size(10 ** 6)
                fill, steps = 3, 25
            elif case == "wide_migration_table":
                # 16 transmission classes x 130 populations x 2 susceptibility groups: the out-migration table of a population
                # (4160 running sums) does not fit the events kernel's LDS budget and is bisected in global memory
                s = Simulator(number_of_sites=3, populations_number=130, number_of_susceptible_groups=2, seed=37)
                s.set_transmission_rate(2.0); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
                for hn in range(15):
                    s.set_transmission_rate(2.1 + 0.1 * hn, haplotype=hn)
                s.set_susceptibility(0.5, susceptibility_type=1); s.set_immunity_transition(0.02, source=1, target=0)
                s.set_susceptibility_type(1)
                s.set_total_migration_probability(0.05); s.set_population_size(10 ** 5)
                fill, steps = 4, 25
            elif case == "one_mutant_rescues":
                s = Simulator(number_of_sites=5, populations_number=4, seed=13)
                s.set_transmission_rate(0.5); s.set_recovery_rate(1.5); s.set_sampling_rate(0.5); s.set_mutation_rate(1.5)
                s.set_total_migration_probability(0.02); s.set_population_size(10 ** 6)
                fill, steps = 1, 40
            else:
                s = Simulator(number_of_sites=2, populations_number=3, seed=5)
                s.set_transmission_rate(6.0); s.set_recovery_rate(0.2); s.set_sampling_rate(0.05); s.set_mutation_rate(0.8)
                s.set_migration_probability(0.1); s.set_population_size(400)
                fill, steps = 20, 40          # 16 haplotypes x 20 = 320 of 400 hosts infected: the upper bound bites
        m = s.simulation
        m.infectious[:] = fill
        if case == "small_next_to_large":
            m.infectious[:, 1::2] = 1          # every other haplotype: one host
        m.susceptible[:, 0] -= m.infectious.sum(axis=1)
        m.totalInfectious[:] = m.infectious.sum(axis=1)
        m.totalSusceptible[:] = m.susceptible.sum(axis=1)
        m.globalInfectious = int(m.totalInfectious.sum())
        m.first_simulation = True
        m.initial_infectious[:] = m.infectious
        m.initial_susceptible[:] = m.susceptible
        eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
        m.events.CreateEvents(steps)
        m.events.ptr = 1
        m.events.CreateEvents(steps)
        eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([77], dtype=np.int64))
        o = _capi.VgxRunOpts(); o.record_events = 0
        o.reserved[0] = 1                      # every try of the halving loop: many rejected ones
        o.reserved[1] = mode
        eng._check(eng.lib.vgx_simulate_tau(eng.handle, steps, 10 ** 15, -1.0, 1, C.byref(o)))
        eng.get_state(m, 0)
        c = eng.counters(0)
        out = (m, int(c.reserved[0]), int(c.ev_ptr))
        eng.close()
        return out
    a, drawn_a, ptr_a = run(1)
    assert (a.infectious >= 0).all() and (a.infectious.sum(axis=1) + a.susceptible.sum(axis=1) == a.sizes).all()
    assert not np.array_equal(a.infectious, a.initial_infectious)
    for mode in (0, 2):
        b, drawn_b, ptr_b = run(mode)
        assert ptr_a == ptr_b and drawn_a == drawn_b > 0, mode
        assert a.currentTime == b.currentTime > 0, mode
        assert np.array_equal(a.infectious, b.infectious) and np.array_equal(a.susceptible, b.susceptible), mode
        for k in a.COUNTERS:
            assert getattr(a, k) == getattr(b, k), (mode, k)


@pytest.mark.parametrize("sites,weights,asym", [(7, None, False), (8, [1.0, 2.0, 0.5, 1.5], False), (9, None, False), (10, None, False),
                                                (7, None, True), (3, [1.0, 2.0, 0.5, 1.5], True)])
def test_tiled_drift_leap_length_matches_oracle(oracle_mod, sites, weights, asym):
    """The two-pass tiled drift (high sites over row tiles, low sites in an LDS tile) for 1-4 high sites, flat and unequal
    derived-state weights: the first accepted leap after a bit-exact direct warm-up has the oracle's length."""
    from vgsim_amd import Simulator

    def run(engine):
        with helpers.quiet():
            s = Simulator(number_of_sites=sites, populations_number=2, seed=3)
            s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.3)
            if weights is not None:
                s.set_mutation_probabilities(weights)
            s.set_migration_probability(0.01)      # one common probability: closed form of the migration pressure (colsum)
            if asym:                               # a general matrix: the [P x P] x [P x H] product (migin)
                s.set_migration_probability(0.03, source=1, target=0)
                s.set_contact_density(0.7, population=1)
            m = s.simulation
            if engine == "hip":
                s.simulate(3000)
                s.simulate(2, sample_size=10 ** 9, method="tau")
            else:
                assert oracle_mod.run_direct(m, 3000, 3000, -1, 200) == 0
                assert oracle_mod.run_tau(m, 2, 10 ** 9, -1, 200) == 0
        return m
    hip, ref = run("hip"), run("oracle")
    assert np.array_equal(hip.events.as_array()[:, :3000], ref.events.as_array()[:, :3000])
    assert hip.events.types[3000] == 6 and ref.events.types[3000] == 6
    assert len(np.nonzero(hip.infectious.sum(axis=0))[0]) > (50 if sites > 3 else 20)   # mutants spread over many haplotypes
    dt_hip = hip.events.times[3000] - hip.events.times[2999]
    dt_ref = ref.events.times[3000] - ref.events.times[2999]
    # the accepted leap is the chosen tau after however many halvings each side's own random draws needed (pyx:2316-2321):
    # equal up to a power of two
    k = np.log2(dt_ref / dt_hip)
    assert abs(k - round(k)) < 1e-8 and abs(round(k)) <= 12, (dt_hip, dt_ref)
    assert dt_hip * 2.0 ** round(k) == pytest.approx(dt_ref, rel=1e-9)


def test_cross_compartment_list_grows_on_demand():
    """Four million compartments of 40 hosts with a high mutation rate: far more mutants per leap than the initial
    capacity of the cross-compartment list (4M entries).  The overflowing try is discarded, the list doubled and the same
    try run again; hosts are conserved and every mutant is accounted for."""
    import ctypes as C
    from vgsim_amd import Simulator, _capi
    with helpers.quiet():
        s = Simulator(number_of_sites=9, populations_number=16, seed=5)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.5)
    s.set_population_size(2 * 10 ** 9)
    m = s.simulation
    m.infectious[:] = 40
    m.susceptible[:, 0] -= 40 * m.hapNum
    eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=1)
    m.events.CreateEvents(2); m.events.ptr = 1; m.events.CreateEvents(2)
    eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([5], dtype=np.int64))
    o = _capi.VgxRunOpts(); o.record_events = 0
    before = int(m.infectious.sum()) + 1          # + the index case the first call adds (pyx:435-448)
    eng._check(eng.lib.vgx_simulate_tau(eng.handle, 2, 10 ** 15, -1.0, 1, C.byref(o)))
    eng.get_state(m, 0)
    c = eng.counters(0)
    eng.close()
    assert c.loop_iterations == 2 and m.mCounter > 2 * 4194304        # more mutants per leap than the list held
    assert (m.infectious >= 0).all()
    assert int(m.infectious.sum()) == before + m.bCounter - m.dCounter - m.sCounter
    assert int(m.susceptible.sum() + m.infectious.sum()) == int(m.sizes.sum())


def test_direct_tau_direct_on_one_object():
    """direct -> tau -> direct (case `tau_then_direct`, golden recorded from the reference): the first phase is bit-exact,
    the log has the reference's length and capacity, and the bookkeeping stays consistent through the method changes."""
    hip = helpers.run_case_hip("tau_then_direct").simulation
    meta, z = helpers.load_golden("tau_then_direct")
    chain = hip.events.as_array()
    assert hip.events.ptr == meta["stats"]["ptr"] == 3030 and chain.shape[1] == meta["size"]
    assert np.array_equal(chain[1:, :1500].astype(np.int64), z["ints"][:, :1500])
    assert (chain[1, 1500:1530] == 6).all() and (chain[1, 1530:3030] != 6).all()      # 30 leaps, then single events again
    assert np.all(np.diff(chain[0, :3030]) >= 0)
    me = hip.multievents
    by_type = np.bincount(me.types[:me.ptr], weights=me.num[:me.ptr], minlength=6)
    single = np.bincount(chain[1, :3030].astype(int), minlength=7)
    assert hip.bCounter == single[0] + by_type[0] and hip.dCounter == single[1] + by_type[1]
    assert hip.sCounter == single[2] + by_type[2] and hip.mCounter == single[3] + by_type[3]
    assert hip.iCounter == single[4] + by_type[4] and hip.migPlus == single[5] + by_type[5]
    assert int(hip.susceptible.sum() + hip.infectious.sum()) == int(hip.sizes.sum())
    assert np.array_equal(hip.totalInfectious, hip.infectious.sum(axis=1)) and hip.globalInfectious == hip.infectious.sum()


def test_tau_restart_rechecks_lockdowns_and_keeps_the_log():
    """Tau as the first call on a model whose attempts die out (<= 100 steps, iterations > 100 -> Restart, pyx:2331): the
    lockdown records of failed attempts stay, Restart re-checks every population at time 0, swapLockdown survives.  The
    draws differ from the oracle's (other streams), so structural facts are checked on the device result: counters equal
    the log, hosts are conserved."""
    from vgsim_amd import Simulator
    with helpers.quiet():
        s = Simulator(number_of_sites=1, populations_number=2, number_of_susceptible_groups=1, seed=3)
        s.set_transmission_rate(1.3); s.set_recovery_rate(1.0); s.set_sampling_rate(0.05)
        s.set_population_size(400); s.set_migration_probability(0.05)
        s.set_npi([0.2, 0.01, 0.002])
        s.simulate(400, sample_size=10 ** 9, method='tau', attempts=30)
    m = s.simulation
    assert m.swapLockdown == len(m.loc.times)                  # every switch ever made is in the log, Restarts included
    assert all(t >= 0.0 for t in m.loc.times)
    hosts = m.susceptible.sum() + m.infectious.sum()
    assert hosts == int(m.sizes.sum()) and (m.infectious >= 0).all()
    assert m.multievents.ptr == 0 or (m.multievents.num[:m.multievents.ptr] > 0).all()


def test_tau_without_multievent_rows():
    """``record_multievents=False``: same trajectory (the rows are bookkeeping only), MULTITYPE records with empty ranges."""
    a = run_tau_case("tau_b", 7, "hip")
    from vgsim_amd import Simulator
    ctor, phases = models.CASES["tau_b"]
    with helpers.quiet():
        sim = Simulator(**dict(ctor, seed=7))
        phases[0][0](sim)
        sim.simulate(**phases[0][1])
        sim.simulate(record_multievents=False, **phases[1][1])
    b = sim.simulation
    assert np.array_equal(a.infectious, b.infectious) and a.bCounter == b.bCounter and a.currentTime == b.currentTime
    assert np.array_equal(a.events.times[:a.events.ptr], b.events.times[:b.events.ptr])
    nd = phases[0][1]["iterations"]
    assert b.multievents.ptr == 0 and (b.events.haplotypes[nd:b.events.ptr] == 0).all()


def _c4_scaled(seed):
    """The recipe of bench.py's config-4 leg (uniform fill of 3 hosts per compartment, uniform migration, one rate class, one
    susceptibility group) at 7 sites x 4 populations: 65 536 compartments, one high site — the shape at which the engine takes
    the SAME kernel instantiations as at config 4 (vgx_tau_drift_fast_kernel<true, true> + vgx_tau_muthigh_kernel for the drift,
    the sieve from the histogram, vgx_tau_scan_fast_kernel<true, false>, vgx_tau_events_kernel with the LDS tables) and the
    oracle still runs a leap in 0.1 s."""
    from vgsim_amd import Simulator
    with helpers.quiet():
        s = Simulator(number_of_sites=7, populations_number=4, seed=seed)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01); s.set_population_size(10 ** 7)
    m = s.simulation
    m.infectious[:] = 3
    m.susceptible[:, 0] -= 3 * m.hapNum
    m.totalInfectious[:] = 3 * m.hapNum
    m.totalSusceptible[:] = m.susceptible.sum(axis=1)
    m.globalInfectious = int(m.totalInfectious.sum())
    m.first_simulation = True
    m.initial_infectious[:] = m.infectious
    m.initial_susceptible[:] = m.susceptible
    return s


def test_config4_kernel_instantiations_match_oracle_distribution(oracle_mod):
    """Config 4's own kernels pinned distributionally: N seeds x 6 leaps on the scaled-down config-4 state, the device's as one
    ensemble launch, against the oracle's runs of the same state.  For the five event kinds (= the per-kind totals of the
    multievent rows), the epidemic time, the infectious total of every population, of every value of the HIGH site's digit
    and of every value of the last site's digit: |mean_1 - mean_2| <= 4.5 sqrt(s1^2/n + s2^2/n) + floor and
    |var_1 - var_2| <= 4.5 sqrt((m4_1 - s1^4)/n + (m4_2 - s2^4)/n) + floor (floor: half an event / 1e-9 time units)."""
    from vgsim_amd.ensemble import Ensemble
    n, nt = 48, 3          # tau as the first call of a model: capacity 2 x iterations (pyx:2298, 2306) -> 6 leaps
    seeds = 70000 + np.arange(n, dtype=np.int64)
    keys = ("bCounter", "dCounter", "sCounter", "mCounter", "migPlus", "currentTime")

    def features(m):
        inf = m.infectious
        H = inf.shape[1]
        hi = inf.reshape(inf.shape[0], 4, H // 4).sum(axis=(0, 2))          # first (high) site: digit value 0..3
        lo = inf.reshape(inf.shape[0], H // 4, 4).sum(axis=(0, 1))          # last site
        return [float(getattr(m, k)) for k in keys] + [float(v) for v in inf.sum(axis=1)] + [float(v) for v in hi] + [float(v) for v in lo]

    base = _c4_scaled(11)
    ens = Ensemble(base, n, seeds=seeds)
    res = ens.simulate_tau(nt, sample_size=10 ** 12, record_events=True)
    assert (res.events == 2 * nt).all()
    dev = np.array([features(ens.replicate_state(r)) for r in range(n)])
    # the per-kind totals of the multievent rows ARE the counters (exact bookkeeping, two replicates)
    for r in (0, n - 1):
        rows = ens.engine.multievents(r)
        st = ens.replicate_state(r)
        by_type = np.bincount(rows["types"], weights=rows["num"], minlength=6)
        assert (by_type[0], by_type[1], by_type[2], by_type[3], by_type[5]) == (st.bCounter, st.dCounter, st.sCounter, st.mCounter, st.migPlus)
        assert (rows["num"] > 0).all()
    ens.close()
    ref = []
    for sd in seeds:
        one = _c4_scaled(int(sd)).simulation
        assert oracle_mod.run_tau(one, nt, 10 ** 12, -1, 200) == 0
        assert one.events.ptr == 2 * nt
        ref.append(features(one))
    ref = np.array(ref)
    names = list(keys) + ["infectious[pop %d]" % i for i in range(4)] + ["infectious[high digit %d]" % i for i in range(4)] + \
        ["infectious[last digit %d]" % i for i in range(4)]
    for j, what in enumerate(names):
        a, b = dev[:, j], ref[:, j]
        floor = 1e-9 if what == "currentTime" else 0.5
        s1, s2 = a.std(ddof=1), b.std(ddof=1)
        tol = 4.5 * np.sqrt(s1 ** 2 / n + s2 ** 2 / n) + floor
        assert abs(a.mean() - b.mean()) <= tol, "%s: means %.8g vs %.8g (tolerance %.3g)" % (what, a.mean(), b.mean(), tol)
        m4a, m4b = ((a - a.mean()) ** 4).mean(), ((b - b.mean()) ** 4).mean()
        tolv = 4.5 * np.sqrt(max(m4a - s1 ** 4, 0.0) / n + max(m4b - s2 ** 4, 0.0) / n) + floor
        assert abs(s1 ** 2 - s2 ** 2) <= tolv, "%s: variances %.6g vs %.6g (tolerance %.3g)" % (what, s1 ** 2, s2 ** 2, tolv)


def _filled(sites, P, S, seed, fill, migration=True, classes=1, uneven=False):
    """A model whose compartments are written straight into the arrays (``fill(rng, shape) -> counts``); classes = 3: two
    haplotypes with rates of their own (three rate classes); uneven: one population of another size (the weights cd / actualSizes
    of uniform migration then differ: the drift pass's second column sum)."""
    from vgsim_amd import Simulator
    with helpers.quiet():
        s = Simulator(number_of_sites=sites, populations_number=P, number_of_susceptible_groups=S, seed=seed)
    s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.01)
    if migration:
        s.set_total_migration_probability(0.01)
    if S > 1:
        s.set_susceptibility_type(1); s.set_susceptibility(0.3, susceptibility_type=1); s.set_immunity_transition(0.02, source=1, target=0)
    if classes == 3:
        s.set_transmission_rate(3.1, haplotype=5); s.set_recovery_rate(0.5, haplotype=9)
    s.set_population_size(10 ** 8)
    if uneven:
        s.set_population_size(3 * 10 ** 7, population=1)
    m = s.simulation
    m.infectious[:] = fill(np.random.default_rng(seed), m.infectious.shape)
    m.susceptible[:, 0] -= m.infectious.sum(axis=1)
    m.totalInfectious[:] = m.infectious.sum(axis=1)
    m.totalSusceptible[:] = m.susceptible.sum(axis=1)
    m.globalInfectious = int(m.totalInfectious.sum())
    m.first_simulation = True
    m.initial_infectious[:] = m.infectious
    m.initial_susceptible[:] = m.susceptible
    return s


def _fill_small(rng, shape):
    return rng.integers(0, 7, size=shape)


def _fill_saturated(rng, shape):      # mostly small counts, a few hundred compartments of 255 hosts and (far) more
    a = rng.integers(0, 5, size=shape)
    idx = rng.integers(0, a.size, size=300)
    a.reshape(-1)[idx] = rng.choice([254, 255, 256, 300, 1000, 70000], size=300)
    return a


def _fill_sparse_mixed(rng, shape):    # 1 % occupied, from single hosts to counts far beyond a byte (the drift pass's screen: mostly empty turns)
    a = np.zeros(shape, dtype=np.int64)
    n = a.size // 100
    idx = rng.choice(a.size, size=n, replace=False)
    a.reshape(-1)[idx] = rng.choice([1, 1, 2, 3, 5, 40, 67, 254, 255, 300, 5000], size=n)
    return a


@pytest.mark.parametrize("sites,P,S,fill,migration,uneven", [
    (7, 4, 1, _fill_small, True, False), (8, 3, 2, _fill_saturated, True, False), (9, 3, 1, _fill_saturated, True, False),
    (10, 2, 1, _fill_small, False, False), (10, 9, 2, _fill_saturated, True, False),
    (9, 2, 1, _fill_sparse_mixed, True, False), (10, 3, 2, _fill_sparse_mixed, True, False), (8, 4, 1, _fill_sparse_mixed, False, False),
    (8, 3, 1, _fill_small, True, True), (9, 3, 2, _fill_sparse_mixed, True, True)])
def test_byte_drift_pass_equals_the_two_pass_form(monkeypatch, sites, P, S, fill, migration, uneven):
    """vgx_tau_drift8_kernel (one read of the one-byte counts, neighbour sums inside a 4^8 tile and from the row's other tiles)
    against the two-pass form it replaces (VGX_TAU_NO_BYTE_DRIFT=1: column sums, high-site pass, low-site pass on the 4-byte
    counts).  The byte form collects the terms of a compartment's drift (fused multiply-adds) and sums the susceptible
    compartments' drift over another partition of the compartments: the leap lengths agree to 1e-12, and — a last-bit difference
    of tau moves no Poisson draw — the accepted steps, events and states are identical.  Also where bytes are saturated (counts of
    255 and more: flagged tiles, sums formed again from the 4-byte counts), with the one-byte copy kept in step by the apply pass
    over several leaps.  Since round 4 the byte form screens the compartments in single precision before it forms a drift exactly
    (only the launch's smallest candidate is kept): sparse states with counts of every size and populations of uneven size (the
    form that is not compiled for the usual shape) are here for that."""
    def run(two_pass):
        if two_pass:
            monkeypatch.setenv("VGX_TAU_NO_BYTE_DRIFT", "1")
        else:
            monkeypatch.delenv("VGX_TAU_NO_BYTE_DRIFT", raising=False)
        s = _filled(sites, P, S, 100 + sites, fill, migration, uneven=uneven)
        with helpers.quiet():
            s.simulate(3, sample_size=10 ** 12, method="tau", record_multievents=False)
        return s.simulation
    a, b = run(False), run(True)
    assert a.events.ptr == b.events.ptr == 6
    np.testing.assert_allclose(a.events.times[:6], b.events.times[:6], rtol=1e-12, atol=0)
    assert np.array_equal(a.infectious, b.infectious) and np.array_equal(a.susceptible, b.susceptible)
    for k in a.COUNTERS:
        assert getattr(a, k) == getattr(b, k), k
    assert a.bCounter > 0


@pytest.mark.parametrize("sites,P,S,fill,classes", [(8, 3, 2, _fill_saturated, 1), (9, 3, 1, _fill_small, 1), (7, 4, 1, _fill_small, 1), (10, 2, 1, _fill_small, 1),
                                                    (7, 4, 2, _fill_small, 3), (8, 2, 1, _fill_saturated, 3)])
def test_front_pass_only_ends_lost_tries_early(monkeypatch, sites, P, S, fill, classes):
    """The front pass of a try (vgx_tau_front_kernel + vgx_tau_events_kernel<., true>: the compartments that can fall below zero on
    their own, drawn first and without bookkeeping) may only end a try early that the try proper would have rejected: eight leaps
    with it and without it (VGX_TAU_NO_FRONT=1) are the same leaps — same number of tries (the sieve's and the loop's), same times,
    same events, same state."""
    def run(off):
        if off:
            monkeypatch.setenv("VGX_TAU_NO_FRONT", "1")
        else:
            monkeypatch.delenv("VGX_TAU_NO_FRONT", raising=False)
        s = _filled(sites, P, S, 700 + sites, fill, True, classes=classes)
        with helpers.quiet():
            s.simulate(8, sample_size=10 ** 12, method="tau", record_multievents=False)
        return s.simulation
    a, b = run(False), run(True)
    assert a.events.ptr == b.events.ptr == 16
    assert np.array_equal(a.events.times[:16], b.events.times[:16])
    assert np.array_equal(a.infectious, b.infectious) and np.array_equal(a.susceptible, b.susceptible)
    for k in a.COUNTERS:
        assert getattr(a, k) == getattr(b, k), k


def _fill_sparse(rng, shape):          # 1 % of the compartments occupied: mostly a few hosts, some hundreds, a few far beyond a byte
    a = np.zeros(shape, dtype=np.int64)
    n = a.size // 100
    idx = rng.choice(a.size, size=n, replace=False)
    a.reshape(-1)[idx] = rng.choice([1, 1, 2, 3, 5, 40, 254, 255, 300, 5000], size=n)
    return a


def _fill_one_region(rng, shape):      # sparse overall, but 8192 occupied compartments in ONE region of the lists (capacity 1024)
    a = np.zeros(shape, dtype=np.int64)
    h = np.arange(65536)
    a[0, h[(h % 2048) < 256]] = rng.integers(1, 6, size=8192)
    a[1, rng.choice(shape[1], size=500, replace=False)] = 3
    return a


@pytest.mark.parametrize("sites,P,S,fill", [(8, 3, 2, _fill_sparse), (9, 2, 1, _fill_sparse), (10, 2, 1, _fill_sparse), (7, 4, 1, _fill_sparse),
                                             (9, 2, 1, _fill_one_region)])
def test_tries_over_the_lists_of_occupied_compartments(monkeypatch, sites, P, S, fill):
    """Sparse states: the drift pass lists the occupied compartments and a try's scan and front pass go over the lists
    (vgx_tau_listscan_kernel) instead of streaming all P x H compartments.  Same buckets, same thresholds: eight leaps with the lists
    and without them (VGX_TAU_NO_OCCLIST=1) are the same leaps.  `_fill_one_region` puts 8192 occupied compartments into one region of the lists
    (capacity 1024): that region is swept compartment by compartment, and the run is still the same."""
    def run(off):
        if off:
            monkeypatch.setenv("VGX_TAU_NO_OCCLIST", "1")
        else:
            monkeypatch.delenv("VGX_TAU_NO_OCCLIST", raising=False)
        s = _filled(sites, P, S, 900 + sites, fill, True)
        with helpers.quiet():
            s.simulate(8, sample_size=10 ** 12, method="tau", record_multievents=False)
        return s.simulation
    a, b = run(False), run(True)
    assert a.events.ptr == b.events.ptr == 16
    assert np.array_equal(a.events.times[:16], b.events.times[:16])
    assert np.array_equal(a.infectious, b.infectious) and np.array_equal(a.susceptible, b.susceptible)
    for k in a.COUNTERS:
        assert getattr(a, k) == getattr(b, k), k
    assert a.bCounter > 0


def _fill_heavy(rng, shape):           # sparse, with lineages of 10^5 .. 10^7 hosts: their EMPTY neighbours (mutants arriving) and the empty
    a = _fill_sparse(rng, shape)       # compartments of their columns (migrants arriving) hold the smallest candidate of ChooseTau
    a[0, 12345 % shape[1]] = 4 * 10 ** 6
    a[shape[0] - 1, 777] = 10 ** 5
    a[shape[0] - 1, 778] = 9 * 10 ** 6
    return a


@pytest.mark.parametrize("sites,P,S,fill,uneven", [(8, 3, 2, _fill_sparse, False), (9, 2, 1, _fill_sparse, False), (10, 3, 1, _fill_sparse_mixed, False),
                                                    (8, 4, 1, _fill_heavy, False), (10, 2, 2, _fill_heavy, False), (9, 3, 1, _fill_heavy, True),
                                                    (9, 2, 1, _fill_one_region, False), (7, 4, 1, _fill_sparse, False), (9, 3, 2, _fill_sparse_mixed, True)])
def test_drift_over_the_lists_equals_the_dense_pass(monkeypatch, sites, P, S, fill, uneven):
    """Sparse states with uniform migration: the column sums' pass lists the occupied compartments and the drift pass goes over the lists
    (vgx_tau_drift8s_*: the occupied compartments, the empty neighbours of the large ones, the empty compartments of the columns with
    large sums) instead of streaming every byte (VGX_TAU_DENSE_DRIFT=1: vgx_tau_drift8_kernel).  The compartments' candidates are the
    same bit patterns; the susceptible compartments' drift is summed in another order: leap lengths to 1e-12, steps, events and states
    identical.  `_fill_heavy`: the minimum is an EMPTY compartment's (next to a lineage of millions / in its column)."""
    def run(dense):
        if dense:
            monkeypatch.setenv("VGX_TAU_DENSE_DRIFT", "1")
        else:
            monkeypatch.delenv("VGX_TAU_DENSE_DRIFT", raising=False)
        s = _filled(sites, P, S, 1300 + sites, fill, True, uneven=uneven)
        with helpers.quiet():
            s.simulate(6, sample_size=10 ** 12, method="tau", record_multievents=False)
        return s.simulation
    a, b = run(False), run(True)
    assert a.events.ptr == b.events.ptr == 12
    np.testing.assert_allclose(a.events.times[:12], b.events.times[:12], rtol=1e-12, atol=0)
    assert np.array_equal(a.infectious, b.infectious) and np.array_equal(a.susceptible, b.susceptible)
    for k in a.COUNTERS:
        assert getattr(a, k) == getattr(b, k), k
    assert a.bCounter > 0


@pytest.mark.parametrize("sites,P,S,fill,uneven", [(8, 3, 1, _fill_heavy, False), (8, 2, 2, _fill_sparse, False), (9, 3, 1, _fill_heavy, True)])
def test_sparse_drift_leap_length_matches_oracle(oracle_mod, sites, P, S, fill, uneven):
    """The leap ChooseTau (pyx:2432-2450) gives a SPARSE state, formed by the drift pass that takes only the occupied dwords and the empty
    compartments that can matter (vgx_tau_drift8s_*; from a call's second step on: the first one counts the occupied compartments),
    against the oracle's on the same state: leap * 2^(rejected tries) to 1e-9.  `_fill_heavy`: the minimum is an empty compartment's."""
    def start():
        return _filled(sites, P, S, 1500 + sites, fill, True, uneven=uneven)
    # (a call of n iterations makes 2 n steps: upstream's two CreateEvents calls, pyx:2298 / 2306)
    one, two = start(), start()
    with helpers.quiet():
        one.simulate(1, sample_size=10 ** 12, method="tau", record_multievents=False)     # steps 1, 2
        two.simulate(2, sample_size=10 ** 12, method="tau", record_multievents=False)     # the same two, then steps 3, 4
    m1, m2 = one.simulation, two.simulation
    assert m1.events.ptr == 2 and m2.events.ptr == 4 and np.array_equal(m1.events.times[:2], m2.events.times[:2])
    dt_hip = float(m2.events.times[2] - m2.events.times[1])                               # step 3: chosen for the state after step 2
    tries_hip = int(m2._engine.tau_tries(0, 2, 1)[0])
    # the oracle's next step from that state
    assert oracle_mod.run_tau(m1, 1, 10 ** 12, -1, 200) == 0
    dt_ref = float(m1.events.times[2] - m1.events.times[1])
    tries_ref = oracle_mod.tau_tries(0)
    assert dt_hip > 0 and dt_ref > 0 and tries_ref >= 0
    assert dt_hip * 2.0 ** tries_hip == pytest.approx(dt_ref * 2.0 ** tries_ref, rel=1e-9), (dt_hip, tries_hip, dt_ref, tries_ref)


def test_lists_and_front_pass_with_several_replicates(monkeypatch):
    """Three replicates of a sparse 8-site model on the step kernels (their tries end at different places: no front pass alone,
    one list of occupied compartments per replicate): the runs with the lists and the front pass equal those without."""
    import ctypes as C
    from vgsim_amd import _capi

    def run(plain):
        for k in ("VGX_TAU_NO_FRONT", "VGX_TAU_NO_OCCLIST"):
            if plain:
                monkeypatch.setenv(k, "1")
            else:
                monkeypatch.delenv(k, raising=False)
        s = _filled(8, 3, 2, 1234, _fill_sparse, True)
        m = s.simulation
        eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=3)
        m.events.CreateEvents(6); m.events.CreateEvents(6)
        eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([21, 22, 23], dtype=np.int64))
        o = _capi.VgxRunOpts(); o.record_events = 0
        eng._check(eng.lib.vgx_simulate_tau(eng.handle, 6, 10 ** 15, -1.0, 1, C.byref(o)))
        out = []
        for r in range(3):
            c = eng.counters(r)
            eng.get_state(m, r)
            out.append((c.ev_ptr, c.loop_iterations, c.reserved[0], m.infectious.copy(), m.susceptible.copy(), m.currentTime))
        eng.close()
        return out
    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert x[:3] == y[:3] and x[5] == y[5]
        assert np.array_equal(x[3], y[3]) and np.array_equal(x[4], y[4])
    assert a[0][2] > 0 and not np.array_equal(a[0][3], a[1][3])


@pytest.mark.parametrize("sites,pops", [(2, 3), (3, 4)])
def test_small_model_loop_does_not_depend_on_the_workgroup_size(monkeypatch, sites, pops):
    """vgx_taus.hip runs a replicate's step loop in one workgroup of 64, 256 or 512 threads (few replicates: many lanes per step; large
    ensembles: many small workgroups per CU).  Streams are keyed by the channel and the sums are formed in one order, so the three
    give the same runs: same states and times after 300 steps of three replicates of a model with two susceptibility groups."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble

    def run(tt):
        monkeypatch.setenv("VGX_TAUS_THREADS", str(tt))
        with helpers.quiet():
            s = Simulator(number_of_sites=sites, populations_number=pops, number_of_susceptible_groups=2, seed=7)
        s.set_transmission_rate(2.5); s.set_recovery_rate(0.9); s.set_sampling_rate(0.1); s.set_mutation_rate(0.05)
        s.set_total_migration_probability(0.002); s.set_population_size(10 ** 6)
        s.set_susceptibility_type(1); s.set_susceptibility(0.3, susceptibility_type=1); s.set_immunity_transition(0.02, source=1, target=0)
        with helpers.quiet():
            s.simulate(2000, sample_size=10 ** 12)
        ens = Ensemble(s, 3)
        ens.simulate_tau(300, sample_size=10 ** 15, seeds=np.array([5, 6, 7], dtype=np.int64))
        out = []
        for r in range(3):
            st = ens.replicate_state(r)
            out.append((st.infectious.copy(), st.susceptible.copy(), st.currentTime))
        ens.close()
        return out
    ref = run(512)
    for tt in (64, 256):
        got = run(tt)
        for x, y in zip(ref, got):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2], tt
    assert ref[0][2] > 0 and not np.array_equal(ref[0][0], ref[1][0])


@pytest.mark.parametrize("sites,P,S", [(8, 3, 2), (2, 3, 1)])
def test_staged_start_state_gives_the_same_run(sites, P, S):
    """vgx_stage_tau (snapshot, conversion and upload of the start state ahead of the call: the bench's hand-over) against
    vgx_simulate_tau doing that work itself: same steps, same state, also for a second call after a new vgx_set_state."""
    import ctypes as C
    from vgsim_amd import _capi

    def run(stage):
        s = _filled(sites, P, S, 300 + sites, _fill_small, True)
        m = s.simulation
        eng = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=2)
        out = []
        for call in range(2):
            m.events.CreateEvents(4); m.events.CreateEvents(4)
            eng.set_params(m); eng.set_state(m); eng.set_seeds(np.array([11, 12], dtype=np.int64))
            if stage:
                eng.stage_tau()
            o = _capi.VgxRunOpts(); o.record_events = 0
            eng._check(eng.lib.vgx_simulate_tau(eng.handle, 4, 10 ** 15, -1.0, 1, C.byref(o)))
            for r in range(2):
                c = eng.counters(r)
                out.append((c.ev_ptr, c.loop_iterations, c.reserved[0]))
            eng.get_state(m, 1)            # continue the second call from replicate 1's end state
            out.append((m.infectious.copy(), m.susceptible.copy(), m.currentTime))
        eng.close()
        return out
    a, b = run(False), run(True)
    for x, y in zip(a, b):
        if isinstance(x[0], np.ndarray):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2]
        else:
            assert x == y
    assert a[0][1] >= 4 and a[0][2] > 0
