"""Model definitions shared by the golden-vector generator (run against the reference's ``Simulator``)
and by the parity tests (run against this repo's ``Simulator`` + oracle / HIP engine).

Each case is a list of *phases*; a phase = (setup callable applied to the simulator, simulate kwargs).
Cases g1..g9 restate the parameters of the reference's nine golden models
(``testing/getting_reference.py:1-107`` == ``testing/check_simulator.py:37-151``); the others follow
SURVEY.md §8(c)/(d) (config-2 model, scan stress, example.py parameters, tau cases, scaled-down config 3).
"""


def _ctor(**kw):
    return kw


# ---------------------------------------------------------------- the nine reference goldens (seed 2020)
def _g1(s):
    s.set_transmission_rate(4.0)
    s.set_recovery_rate(1.5)
    s.set_sampling_rate(0.3)


def _g2(s):
    s.set_transmission_rate(4, haplotype=3)


def _g3(s):
    s.set_susceptibility_type(1)


def _g4(s):
    s.set_mutation_rate(0.01)
    s.set_susceptibility_type(1, haplotype=0)
    for h in (1, 2, 3):
        s.set_susceptibility_type(2, haplotype=h)
    s.set_immunity_transition(0.01, source=0, target=1)
    s.set_immunity_transition(0.01, source=1, target=2)
    s.set_immunity_transition(0.02, source=2, target=1)


def _g5(s):
    s.set_population_size(2000000)
    s.set_contact_density(1.3, population=0)
    s.set_contact_density(0.8, population=1)
    s.set_migration_probability(0.01, source=0, target=1)
    s.set_migration_probability(0.005, source=1, target=0)


def _g6(s):
    s.set_migration_probability(0.01, source=0, target=1)
    s.set_migration_probability(0.005, source=2, target=1)


def _g7(s):
    _g6(s)
    s.set_sampling_multiplier(2.5, population=1)
    s.set_sampling_multiplier(2, population=2)
    s.set_npi([0.5, 0.30, 0.15], population=0)


def _g8(s):
    s.set_mutation_rate(0.01)
    s.set_mutation_probabilities([1, 0, 0, 1])


def _g9(s):
    s.set_transmission_rate(5.0, haplotype=12)
    s.set_recovery_rate(1.5)
    s.set_sampling_rate(0.3)
    s.set_mutation_rate(0.01)
    s.set_mutation_probabilities([1, 0, 0, 1])
    s.set_migration_probability(0.01, source=0, target=1)
    s.set_migration_probability(0.005, source=2, target=1)
    s.set_sampling_multiplier(2.5, population=1)
    s.set_sampling_multiplier(2, population=2)
    s.set_npi([0.5, 0.30, 0.15], population=1)
    s.set_susceptibility_type(1, haplotype=0)
    for h in (1, 2, 3):
        s.set_susceptibility_type(2, haplotype=h)
    s.set_immunity_transition(0.000001, source=0, target=1)
    s.set_immunity_transition(0.000001, source=1, target=2)
    s.set_immunity_transition(0.000002, source=2, target=1)


_NINE = {
    1: (_ctor(seed=2020), _g1),
    2: (_ctor(number_of_sites=1, seed=2020), _g2),
    3: (_ctor(number_of_susceptible_groups=2, seed=2020), _g3),
    4: (_ctor(number_of_sites=1, number_of_susceptible_groups=3, seed=2020), _g4),
    5: (_ctor(populations_number=2, seed=2020), _g5),
    6: (_ctor(populations_number=3, seed=2020), _g6),
    7: (_ctor(populations_number=3, seed=2020), _g7),
    8: (_ctor(number_of_sites=2, seed=2020), _g8),
    9: (_ctor(number_of_sites=2, populations_number=3, number_of_susceptible_groups=3, seed=2020), _g9),
}


# ---------------------------------------------------------------- further cases
def _c2(s):  # BASELINE config 2: 1 haplotype, 1 population, 10^6 hosts
    s.set_transmission_rate(4.0)
    s.set_recovery_rate(1.5)
    s.set_sampling_rate(0.3)


def _stress(s):  # every compartment occupied, all event types
    s.set_transmission_rate(2.5)
    s.set_recovery_rate(0.9)
    s.set_sampling_rate(0.1)
    s.set_mutation_rate(0.4)
    s.set_susceptibility(0.3, susceptibility_type=1)
    s.set_susceptibility_type(1)
    s.set_immunity_transition(0.02, source=1, target=0)
    s.set_migration_probability(0.02)


def _stress_classes(s):  # several fitness classes + per-site mutation rates + NPI, H=256, P=8
    s.set_transmission_rate(2.5)
    s.set_transmission_rate(3.5, haplotype='G***')
    s.set_transmission_rate(3.0, haplotype='*T*C')
    s.set_recovery_rate(0.9)
    s.set_recovery_rate(0.7, haplotype='C***')
    s.set_sampling_rate(0.1)
    s.set_mutation_rate(0.3)
    s.set_mutation_rate(0.6, mutation=2)
    s.set_mutation_probabilities([2, 1, 0, 3], mutation=1)
    s.set_susceptibility_type(1)
    s.set_susceptibility_type(2, haplotype='G***')
    s.set_susceptibility(0.4, susceptibility_type=1)
    s.set_susceptibility(0.1, susceptibility_type=2)
    s.set_susceptibility(0.6, susceptibility_type=2, haplotype='G***')
    s.set_immunity_transition(0.05, source=1, target=0)
    s.set_immunity_transition(0.02, source=2, target=1)
    s.set_total_migration_probability(0.05)
    s.set_population_size(200000)
    s.set_sampling_multiplier(2.0, population=3)
    s.set_contact_density(1.4, population=1)
    s.set_npi([0.2, 0.002, 0.0005], population=0)
    s.set_npi([0.3, 0.004, 0.001], population=5)


def _example_1(s):  # testing/example.py:11-45
    s.set_transmission_rate(0.25)
    s.set_transmission_rate(0.5, haplotype="GG")
    s.set_recovery_rate(0.099)
    s.set_sampling_rate(0.001)
    mutation_rate = 0.00003
    s.set_mutation_rate(mutation_rate)
    s.set_mutation_probabilities([1, 1, 1, 2])
    s.set_mutation_rate(3 * mutation_rate, haplotype='G*', mutation=1)
    s.set_susceptibility_type(1)
    s.set_susceptibility_type(2, haplotype='G*')
    s.set_susceptibility(0.1, susceptibility_type=1)
    s.set_susceptibility(0.5, susceptibility_type=1, haplotype='G*')
    s.set_susceptibility(0.0, susceptibility_type=2)
    s.set_immunity_transition(1 / 90, source=1, target=0)
    s.set_immunity_transition(1 / 180, source=2, target=0)
    s.set_population_size(10000000, population=0)
    s.set_population_size(5000000, population=1)
    s.set_population_size(1000000, population=2)
    s.set_migration_probability(10 / 365 / 2)
    s.set_sampling_multiplier(3, population=1)
    s.set_sampling_multiplier(0, population=2)
    s.set_npi([0.1, 0.01, 0.002])


def _example_2(s):  # testing/example.py:51-56
    s.set_immunity_transition(0.05, source=0, target=1)
    s.set_immunity_transition(0.05, source=0, target=2)
    s.set_contact_density(0.7, population=0)
    s.set_contact_density(0.7, population=1)
    s.set_migration_probability(2 / 365 / 2, source=0, target=2)
    s.set_migration_probability(2 / 365 / 2, source=1, target=2)


def _continuation_2(s):  # parameter changes between two direct calls
    s.set_contact_density(0.6, population=0)
    s.set_immunity_transition(0.04, source=0, target=1)
    s.set_transmission_rate(3.1, haplotype=5)


def _tau_common(P, S):
    def f(s):
        s.set_transmission_rate(2.5)
        s.set_recovery_rate(0.9)
        s.set_sampling_rate(0.1)
        s.set_mutation_rate(0.01)
        if P > 1:
            s.set_migration_probability(0.005)
            s.set_npi([0.3, 0.0005, 0.0001], population=0)
        if S > 1:
            s.set_susceptibility_type(1)
            s.set_susceptibility(0.2, susceptibility_type=1)
            s.set_immunity_transition(0.03, source=1, target=0)
    return f


def _c3_scaled(s):  # SURVEY §8(d) config-3 recipe at a shape the reference can still run
    s.set_transmission_rate(2.5)
    s.set_recovery_rate(0.9)
    s.set_sampling_rate(0.1)
    s.set_mutation_rate(0.01)
    s.set_total_migration_probability(0.01)
    s.set_population_size(10 ** 7)


def _c3_spread_warm(s):
    _c3_scaled(s)
    s.set_mutation_rate(0.4)


def _c3_spread_timed(s):
    s.set_mutation_rate(0.01)


def _nothing(s):
    pass


def _direct(n, **kw):
    d = dict(iterations=n)
    d.update(kw)
    return d


# name -> (constructor kwargs, [ (setup, simulate kwargs), ... ])
CASES = {}
for _k, (_c, _f) in _NINE.items():
    CASES["g%d" % _k] = (_c, [(_f, _direct(100000))])          # the reference's own goldens
    CASES["g%d_short" % _k] = (_c, [(_f, _direct(5000))])      # full chains small enough to commit
CASES["c2"] = (_ctor(number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=2020),
               [(_c2, _direct(2000000))])
CASES["stress_h64"] = (_ctor(number_of_sites=3, populations_number=2, number_of_susceptible_groups=2, seed=77),
                       [(_stress, _direct(60000))])
CASES["stress_h256"] = (_ctor(number_of_sites=4, populations_number=8, number_of_susceptible_groups=3, seed=4242),
                        [(_stress_classes, _direct(20000))])
CASES["example"] = (_ctor(number_of_sites=2, populations_number=3, number_of_susceptible_groups=3, seed=1234),
                    [(_example_1, _direct(10000000, epidemic_time=110))])
# NOTE: example.py's second phase (tau after _example_2, testing/example.py:50-61) is a no-op upstream
# (sample_size defaults to iterations=1000 < sCounter); with an effective sample_size the reference
# build hangs or segfaults inside SimulatePopulation_tau, so no golden can be recorded for it.
CASES["continuation"] = (_ctor(number_of_sites=2, populations_number=3, number_of_susceptible_groups=2, seed=31),
                         [(_g7, _direct(3000)), (_continuation_2, _direct(4000, sample_size=10 ** 9))])
CASES["sample_stop"] = (_ctor(number_of_sites=1, populations_number=2, seed=5),
                        [(_g5, _direct(50000, sample_size=40))])
CASES["time_stop"] = (_ctor(number_of_sites=1, populations_number=2, seed=9),
                      [(_g5, _direct(200000, epidemic_time=7.3))])
CASES["extinct"] = (_ctor(seed=3), [(lambda s: s.set_transmission_rate(0.5), _direct(50, attempts=3))])
CASES["extinct_restart"] = (_ctor(seed=3), [(lambda s: s.set_transmission_rate(0.5), _direct(1000, attempts=3))])
for _name, (_sites, _P, _S, _seed, _nd, _nt) in {"tau_a": (0, 1, 1, 7, 2000, 200), "tau_b": (2, 3, 2, 7, 2000, 200),
                                                  "tau_c": (1, 2, 3, 5, 3000, 150),
                                                  # a long warm-up: thousands of hosts per compartment, hundreds of events per
                                                  # compartment and leap (the engine's per-channel kernel for large compartments)
                                                  "tau_d": (1, 2, 1, 11, 30000, 40)}.items():
    CASES[_name] = (_ctor(number_of_sites=_sites, populations_number=_P, number_of_susceptible_groups=_S, seed=_seed),
                    [(_tau_common(_P, _S), _direct(_nd)),
                     (_nothing, dict(iterations=_nt, sample_size=10 ** 12, method='tau'))])
# direct -> tau -> direct on one object (capacity rules of events.pxi:52-68 across methods, stale rate caches after tau)
CASES["tau_then_direct"] = (_ctor(number_of_sites=1, populations_number=2, number_of_susceptible_groups=2, seed=21),
                            [(_tau_common(2, 2), _direct(1500)),
                             (_nothing, dict(iterations=30, sample_size=10 ** 12, method='tau')),
                             (_nothing, _direct(1500, sample_size=10 ** 9))])
CASES["c3_s5_p16"] = (_ctor(number_of_sites=5, populations_number=16, number_of_susceptible_groups=1, seed=2021),
                      [(_c3_scaled, _direct(6000))])
CASES["c3_s6_p8_spread"] = (_ctor(number_of_sites=6, populations_number=8, number_of_susceptible_groups=1, seed=2022),
                            [(_c3_spread_warm, _direct(4000)), (_c3_spread_timed, _direct(2000, sample_size=10 ** 9))])

def _p70(s):  # more than one 64-lane tile of populations; lockdowns in several demes; 2 susceptibility groups
    s.set_transmission_rate(2.2)
    s.set_transmission_rate(3.0, haplotype=2)
    s.set_recovery_rate(0.8)
    s.set_sampling_rate(0.05)
    s.set_mutation_rate(0.2)
    s.set_susceptibility_type(1)
    s.set_susceptibility(0.5, susceptibility_type=1)
    s.set_immunity_transition(0.05, source=1, target=0)
    s.set_population_size(50000)
    s.set_population_size(20000, population=69)
    s.set_total_migration_probability(0.08)
    s.set_contact_density(1.5, population=65)
    for pn in (0, 3, 64, 69):
        s.set_npi([0.3, 0.004, 0.001], population=pn)
    s.set_sampling_multiplier(3.0, population=66)


CASES["p70"] = (_ctor(number_of_sites=1, populations_number=70, number_of_susceptible_groups=2, seed=99),
                [(_p70, _direct(15000))])
CASES["big_seed"] = (_ctor(number_of_sites=1, seed=2 ** 40 + 12345), [(_g2, _direct(3000))])   # two-word SeedSequence entropy



# ---------------------------------------------------------------- recombination branch of Birth (pyx:575-596)
def _recomb(s):
    s.set_mutation_rate(0.3)
    s.set_sampling_rate(0.1)
    s.set_transmission_rate(3.0, haplotype='T**')
    s.set_susceptibility(0.5, susceptibility_type=1)
    s.set_immunity_transition(0.05, source=1, target=0)
    s.set_susceptibility_type(1)
    s.set_migration_probability(0.02)


def _recomb_positions(s):  # sites inside the genome: the breakpoint falls on either side of the last site
    s.set_mutation_rate(0.2)
    s.set_sampling_rate(0.05)
    s.set_transmission_rate(2.6, haplotype='*G*')
    s.set_total_migration_probability(0.05)
    s.set_mutation_position(1, 300000)
    s.set_mutation_position(2, 600000)


CASES["recomb_a"] = (_ctor(number_of_sites=3, populations_number=2, number_of_susceptible_groups=2, seed=2020,
                           recombination_probability=0.3), [(_recomb, _direct(8000))])
CASES["recomb_pos"] = (_ctor(number_of_sites=3, populations_number=3, number_of_susceptible_groups=1, seed=7,
                             recombination_probability=0.15), [(_recomb_positions, _direct(6000))])
CASES["recomb_restart"] = (_ctor(number_of_sites=2, populations_number=1, number_of_susceptible_groups=1, seed=2,
                                 recombination_probability=0.5),
                           [(lambda s: s.set_transmission_rate(1.3), _direct(3000))])
RECOMBINATION_CASES = ("recomb_a", "recomb_pos", "recomb_restart")


# ---------------------------------------------------------------- the reference's command-line example model
def _cmd_example(s):
    """testing/cmd_example/example.{rt,pp,mg,su,st} (copied as data to tests/golden/cmd_example/) applied the way
    VGsim_cmd.py:113-142 does.  The files are parsed with plain string operations here so that the same setter calls
    reach the reference when the goldens are recorded and this repository's Simulator in the tests."""
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cmd_example")

    def rows(name, skip):
        with open(os.path.join(d, name)) as f:
            return [ln.rstrip().split(" ") for ln in f.read().splitlines()[skip:] if ln.strip() and ln[0] != "#"]
    for i, r in enumerate(rows("example.rt", 2)):
        s.set_transmission_rate(float(r[1]), i)
        s.set_recovery_rate(float(r[2]), i)
        s.set_sampling_rate(float(r[3]), i)
        for j, mut in enumerate(r[4:]):
            a = [float(v) for v in mut.split(",")]
            s.set_mutation_rate(a[0], i, j)
            allele = (i >> (2 * (len(r) - 5 - j))) & 3
            probs = a[1:]
            probs.insert(allele, 0.0)
            s.set_mutation_probabilities(probs, i, j)
    pp = rows("example.pp", 2)
    mg = rows("example.mg", 1)
    for i, r in enumerate(pp):
        s.set_population_size(int(r[1]), i)
        s.set_contact_density(float(r[2]), i)
        s.set_npi([float(v) for v in r[3].split(",")], i)
        s.set_sampling_multiplier(float(r[4]), i)
        for j in range(len(pp)):
            if i != j:
                s.set_migration_probability(float(mg[i][j]), i, j)
    for i, r in enumerate(rows("example.su", 2)):
        for j, v in enumerate(r[2:]):
            s.set_susceptibility(float(v), i, j)
        s.set_susceptibility_type(int(r[1]), i)
    for i, r in enumerate(rows("example.st", 1)):
        for j, v in enumerate(r):
            if i != j:
                s.set_immunity_transition(float(v), i, j)


CASES["cmd_example"] = (_ctor(number_of_sites=2, populations_number=3, number_of_susceptible_groups=3, seed=17),
                        [(_cmd_example, _direct(12000))])

# ---------------------------------------------------------------- cases without a reference fixture
# Added after the fixtures were recorded (no reference build this round): the oracle — pinned by the
# fixtures above — is the checker, bit for bit.
def _lockdown_restart(s):   # tiny demes, low NPI thresholds, R0 near 1: lockdowns switch INSIDE attempts that die out
    s.set_transmission_rate(1.6)
    s.set_recovery_rate(1.0)
    s.set_sampling_rate(0.1)
    s.set_population_size(300)
    s.set_migration_probability(0.05)
    s.set_npi([0.2, 0.02, 0.004])


ORACLE_ONLY_CASES = {
    # 22 failed attempts, several of which leave lockdown records behind (Restart does not clear `loc`, pyx:714-738)
    "lockdown_restart": (_ctor(number_of_sites=1, populations_number=2, number_of_susceptible_groups=1, seed=1),
                         [(_lockdown_restart, _direct(3000, sample_size=10 ** 9, attempts=50))]),
}

def _tau_many_classes(s):   # 300 haplotypes with a recovery rate of their own: more than 256 rate classes
    s.set_transmission_rate(2.5)
    s.set_recovery_rate(0.9)
    s.set_sampling_rate(0.1)
    s.set_mutation_rate(0.08)
    for hn in range(300):
        s.set_recovery_rate(0.5 + 0.002 * hn, haplotype=hn)
    s.set_migration_probability(0.01)


def _tau_wide_table(s):     # 16 transmission classes x 130 populations x 2 susceptibility groups: 4160 out-migration channels
    s.set_transmission_rate(2.0)
    s.set_recovery_rate(0.9)
    s.set_sampling_rate(0.1)
    s.set_mutation_rate(0.08)
    for hn in range(15):
        s.set_transmission_rate(2.1 + 0.1 * hn, haplotype=hn)
    s.set_susceptibility_type(1)
    s.set_susceptibility(0.5, susceptibility_type=1)
    s.set_immunity_transition(0.02, source=1, target=0)
    s.set_population_size(100000)
    s.set_total_migration_probability(0.05)


# tau cases of the device engine's less common table layouts (class tables in global memory / the migration table bisected in
# global memory): the oracle is the checker, distributionally (tests/test_hip_tau.py)
TAU_ORACLE_ONLY_CASES = {
    "tau_many_classes": (_ctor(number_of_sites=5, populations_number=3, number_of_susceptible_groups=1, seed=41),
                         [(_tau_many_classes, _direct(4000)), (_nothing, dict(iterations=120, sample_size=10 ** 12, method='tau'))]),
    "tau_wide_table": (_ctor(number_of_sites=3, populations_number=130, number_of_susceptible_groups=2, seed=43),
                       [(_tau_wide_table, _direct(4000)), (_nothing, dict(iterations=80, sample_size=10 ** 12, method='tau'))]),
}


def tau_case(name):
    return CASES[name] if name in CASES else TAU_ORACLE_ONLY_CASES[name]


# cases whose full (6,N) chain is committed; the others commit head/tail columns + sha256 + counters
FULL_CHAIN_LIMIT = 20000


def build(simulator_cls, name):
    """Construct the simulator of a case and return (simulator, phases)."""
    ctor, phases = CASES[name] if name in CASES else ORACLE_ONLY_CASES[name]
    return simulator_cls(**ctor), phases
