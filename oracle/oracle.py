"""ctypes adapter for the CPU oracle (oracle/vgx_oracle.c) — TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this module; the
product package ``vgsim_amd`` never does.  ``run_direct`` / ``run_tau`` take a host model object
(``vgsim_amd.BirthDeathModel``: parameters and state in numpy arrays named as in the reference), run
the C restatement in place on those arrays and leave the model exactly as the reference's
``SimulatePopulation`` / ``SimulatePopulation_tau`` would (event log, counters, state).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("VGX_ORACLE_LIBRARY") or os.path.join(HERE, "libvgx_oracle.so")

LOG_LIBM, LOG_PORTABLE = 0, 1
RNG_PHILOX = 16   # flag for log_mode: the direct path draws from the engine's counter-based stream (vgx_oracle.h)

_F = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int64)
_U = C.POINTER(C.c_uint64)

_FIELDS = [
    ("sites", C.c_int64), ("hapNum", C.c_int64), ("popNum", C.c_int64), ("susNum", C.c_int64),
    ("user_seed", C.c_int64),
    ("bRate", _F), ("dRate", _F), ("sRate", _F), ("mRate", _F), ("hapMutType", _F), ("susceptibility", _F),
    ("suscType", _I), ("suscepTransition", _F), ("sizes", _I),
    ("contactDensity", _F), ("contactDensityBeforeLockdown", _F), ("contactDensityAfterLockdown", _F),
    ("startLD", _F), ("endLD", _F), ("samplingMultiplier", _F), ("migrationRates", _F),
    ("susceptible", _I), ("infectious", _I), ("initial_susceptible", _I), ("initial_infectious", _I),
    ("totalSusceptible", _I), ("totalInfectious", _I), ("lockdownON", _I),
    ("tmRate", _F), ("eventHapPopRate", _F), ("tEventHapPopRate", _F), ("hapPopRate", _F),
    ("susceptHapPopRate", _F), ("suscepCumulTransition", _F), ("immuneSourcePopRate", _F),
    ("infectPopRate", _F), ("immunePopRate", _F), ("popRate", _F), ("migPopRate", _F), ("actualSizes", _F),
    ("maxEffectiveBirthMigration", _F), ("effectiveMigration", _F),
    ("first_simulation", C.c_int64), ("globalInfectious", C.c_int64),
    ("bCounter", C.c_int64), ("dCounter", C.c_int64), ("sCounter", C.c_int64), ("mCounter", C.c_int64),
    ("iCounter", C.c_int64), ("swapLockdown", C.c_int64), ("migPlus", C.c_int64), ("migNonPlus", C.c_int64),
    ("good_attempt", C.c_int64),
    ("currentTime", C.c_double), ("totalRate", C.c_double), ("totalMigrationRate", C.c_double),
    ("rn", C.c_double), ("tau_l", C.c_double),
    ("ev_size", C.c_int64), ("ev_ptr", C.c_int64), ("ev_times", _F), ("ev_types", _I), ("ev_haplotypes", _I),
    ("ev_populations", _I), ("ev_newHaplotypes", _I), ("ev_newPopulations", _I),
    ("mev_size", C.c_int64), ("mev_ptr", C.c_int64), ("mev_num", _I), ("mev_times", _F), ("mev_types", _I),
    ("mev_haplotypes", _I), ("mev_populations", _I), ("mev_newHaplotypes", _I), ("mev_newPopulations", _I),
    ("loc_cap", C.c_int64), ("loc_n", C.c_int64), ("loc_states", _I), ("loc_populations", _I), ("loc_times", _F),
    ("infectiousAuxTau", _F), ("susceptibleAuxTau", _F), ("infectiousDelta", _I), ("susceptibleDelta", _I),
    ("sparse", C.c_int64), ("log_mode", C.c_int64), ("iterations_done", C.c_int64), ("error", C.c_int64),
    ("occ", _U),
    ("rng_state_hi", C.c_uint64), ("rng_state_lo", C.c_uint64), ("rng_inc_hi", C.c_uint64), ("rng_inc_lo", C.c_uint64),
    ("recombination", C.c_double), ("genome_length", C.c_int64), ("sitesPosition", _I),
    ("rec_cap", C.c_int64), ("rec_n", C.c_int64), ("rec_idevents", _I), ("rec_his", _I), ("rec_hi2s", _I),
    ("rec_nhis", _I), ("rec_posRecombs", _I),
    ("memory_optimization", C.c_int64), ("currentHapNum", C.c_int64), ("maxHapNum", C.c_int64), ("addMemoryNum", C.c_int64),
    ("hapToNum", _I), ("numToHap", _I),
]


class VgoModel(C.Structure):
    _fields_ = _FIELDS


class VgoPcg64(C.Structure):
    _fields_ = [("state_hi", C.c_uint64), ("state_lo", C.c_uint64), ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if os.environ.get("VGX_ORACLE_LIBRARY"):   # e.g. the AddressSanitizer build (make -C oracle asan)
        return LIB
    if force:
        subprocess.check_call(["make", "-s", "-C", HERE, "clean"])
    subprocess.check_call(["make", "-s", "-C", HERE])   # a no-op when the library is newer than its sources
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        for fn in (L.vgo_simulate_direct, L.vgo_simulate_tau):
            fn.argtypes = [C.POINTER(VgoModel), C.c_int64, C.c_int64, C.c_float, C.c_int64]
            fn.restype = C.c_int
        L.vgo_update_all_rates.argtypes = [C.POINTER(VgoModel)]
        L.vgo_update_all_rates.restype = None
        L.vgo_prop_num.argtypes = [C.POINTER(VgoModel)]
        L.vgo_prop_num.restype = C.c_int64
        L.vgo_pcg64_seed.argtypes = [C.POINTER(VgoPcg64), C.c_uint64, C.c_uint32]
        L.vgo_pcg64_double.argtypes = [C.POINTER(VgoPcg64)]
        L.vgo_pcg64_double.restype = C.c_double
        L.vgo_pcg64_next64.argtypes = [C.POINTER(VgoPcg64)]
        L.vgo_pcg64_next64.restype = C.c_uint64
        L.vgo_pcg64_advance.argtypes = [C.POINTER(VgoPcg64), C.c_uint64, C.c_uint64]
        L.vgo_poisson.argtypes = [C.POINTER(VgoPcg64), C.c_double]
        L.vgo_poisson.restype = C.c_int64
        L.vgo_portable_log.argtypes = [C.c_double]
        L.vgo_portable_log.restype = C.c_double
        L.vgo_get_genealogy.argtypes = [C.POINTER(VgoGenealogy)]
        L.vgo_get_genealogy.restype = C.c_int
        L.vgo_hypergeometric.argtypes = [C.POINTER(VgoGenRng), C.c_int64, C.c_int64, C.c_int64]
        L.vgo_hypergeometric.restype = C.c_int64
        _lib = L
    return _lib


def _ptr(a):
    if a.dtype == np.float64:
        return a.ctypes.data_as(_F)
    if a.dtype == np.int64:
        return a.ctypes.data_as(_I)
    if a.dtype == np.uint64:
        return a.ctypes.data_as(_U)
    raise TypeError(a.dtype)


class OracleState:
    """The reference's rate caches (pyx:150-155, 182-199), kept beside a host model between calls so
    tests can inspect them; allocated lazily, owned by Python."""

    def __init__(self, model, record_multievents=False, loc_cap=1 << 16):
        H, P, S = model.hapNum, model.popNum, model.susNum
        z = np.zeros
        self.tmRate = z(H)
        self.eventHapPopRate = z((P, H, 4))
        self.tEventHapPopRate = z((P, H))
        self.hapPopRate = z((P, H))
        self.susceptHapPopRate = z((P, H, S))
        self.suscepCumulTransition = z(S)
        self.immuneSourcePopRate = z((P, S))
        self.infectPopRate, self.immunePopRate, self.popRate = z(P), z(P), z(P)
        self.migPopRate, self.maxEffectiveBirthMigration = z(P), z(P)
        self.effectiveMigration = z((P, P))
        self.infectiousAuxTau = z((P, H))
        self.susceptibleAuxTau = z((P, S))
        self.infectiousDelta = z((P, H), dtype=np.int64)
        self.susceptibleDelta = z((P, S), dtype=np.int64)
        self.occ = z((P, (H + 63) // 64), dtype=np.uint64)
        self.loc_states = z(loc_cap, dtype=np.int64)
        self.loc_populations = z(loc_cap, dtype=np.int64)
        self.loc_times = z(loc_cap)
        self.record_multievents = record_multievents
        self.mev = None          # dict of dense multievent arrays (reference layout) when recorded
        self.mev_ptr = 0
        self.iterations_done = 0


def _struct(model, st, sparse, log_mode):
    m = VgoModel()
    m.sites, m.hapNum, m.popNum, m.susNum = model.sites, model.hapNum, model.popNum, model.susNum
    m.user_seed = model.user_seed
    for name in ("bRate", "dRate", "sRate", "mRate", "hapMutType", "susceptibility", "suscType",
                 "suscepTransition", "sizes", "contactDensity", "contactDensityBeforeLockdown",
                 "contactDensityAfterLockdown", "startLD", "endLD", "samplingMultiplier", "migrationRates",
                 "susceptible", "infectious", "initial_susceptible", "initial_infectious", "totalSusceptible",
                 "totalInfectious", "lockdownON", "actualSizes"):
        a = getattr(model, name)
        assert a.flags["C_CONTIGUOUS"], name
        setattr(m, name, _ptr(a))
    for name in ("tmRate", "eventHapPopRate", "tEventHapPopRate", "hapPopRate", "susceptHapPopRate",
                 "suscepCumulTransition", "immuneSourcePopRate", "infectPopRate", "immunePopRate", "popRate",
                 "migPopRate", "maxEffectiveBirthMigration", "effectiveMigration", "infectiousAuxTau",
                 "susceptibleAuxTau", "infectiousDelta", "susceptibleDelta", "occ", "loc_states",
                 "loc_populations", "loc_times"):
        setattr(m, name, _ptr(getattr(st, name)))
    m.loc_cap, m.loc_n = len(st.loc_states), 0
    m.first_simulation = int(model.first_simulation)
    m.globalInfectious = int(model.globalInfectious)
    for c in model.COUNTERS + ("good_attempt",):
        setattr(m, c, int(getattr(model, c)))
    m.currentTime, m.totalRate, m.totalMigrationRate = model.currentTime, model.totalRate, model.totalMigrationRate
    m.tau_l = model.tau_l
    ev = model.events
    m.ev_size, m.ev_ptr = ev.size, ev.ptr
    m.ev_times = _ptr(ev.times)
    m.ev_types, m.ev_haplotypes, m.ev_populations = _ptr(ev.types), _ptr(ev.haplotypes), _ptr(ev.populations)
    m.ev_newHaplotypes, m.ev_newPopulations = _ptr(ev.newHaplotypes), _ptr(ev.newPopulations)
    m.sparse, m.log_mode = int(sparse), int(log_mode)
    m.iterations_done = 0
    m.recombination = float(model.recombination)
    m.genome_length = int(model.genome_length)
    st.sitesPosition = np.ascontiguousarray(model.sitesPosition, dtype=np.int64)
    m.sitesPosition = _ptr(st.sitesPosition)
    cap = (ev.size + 25000) if model.recombination else 0   # failed attempts (<= 100 events each) keep their records
    st.rec = {k: np.zeros(cap, dtype=np.int64) for k in ("idevents", "his", "hi2s", "nhis", "posRecombs")}
    m.rec_cap, m.rec_n = cap, 0
    for k, a in st.rec.items():
        setattr(m, "rec_" + k, _ptr(a))
    return m


def _absorb(model, st, m):
    model.first_simulation = bool(m.first_simulation)
    model.globalInfectious = m.globalInfectious
    for c in model.COUNTERS + ("good_attempt",):
        setattr(model, c, getattr(m, c))
    model.currentTime, model.totalRate, model.totalMigrationRate = m.currentTime, m.totalRate, m.totalMigrationRate
    model.tau_l = m.tau_l
    model.events.ptr = m.ev_ptr
    for k in range(m.loc_n):
        model.loc.AddLockdown(st.loc_states[k], st.loc_populations[k], st.loc_times[k])
    st.iterations_done = m.iterations_done
    for k in range(min(m.rec_n, m.rec_cap)):   # failed attempts' records stay, like upstream (Restart keeps `rec`)
        model.rec.AddRecombination_forward(*(st.rec[c][k] for c in ("idevents", "his", "hi2s", "nhis", "posRecombs")))
    st.rng_final = (m.rng_state_hi, m.rng_state_lo, m.rng_inc_hi, m.rng_inc_lo)


def get_state(model):
    st = getattr(model, "_oracle_state", None)
    if st is None:
        st = model._oracle_state = OracleState(model)
    return st


def run_direct(model, iterations, sample_size, time, attempts, sparse=False, log_mode=LOG_LIBM):
    """The oracle's SimulatePopulation (pyx:396-429) on a host model; returns the error code."""
    st = get_state(model)
    model.events.CreateEvents(iterations)  # PrepareParameters pyx:434
    m = _struct(model, st, sparse, log_mode)
    rc = lib().vgo_simulate_direct(C.byref(m), iterations, sample_size, float(time), attempts)
    _absorb(model, st, m)
    return rc


class MemoptTable:
    """State of the reference's haplotype table (pyx:105-125) beside a host model, for ``run_direct_memopt``."""

    def __init__(self, model):
        H, P = model.hapNum, model.popNum
        block = int(4 ** max(model.sites - 2, 1))            # pyx:105-111: 4^(sites-2), at least 4
        self.currentHapNum, self.maxHapNum, self.addMemoryNum = 0, block, block
        self.hapToNum = np.zeros(H, dtype=np.int64)
        self.numToHap = np.zeros(max(H, block) + 1, dtype=np.int64)
        self.infectious = np.zeros((P, H), dtype=np.int64)   # counts by PROGRAM NUMBER (upstream's layout), hapNum columns
        self.initial_infectious = np.zeros((P, H), dtype=np.int64)

    def infectious_by_haplotype(self, model):
        out = np.zeros((model.popNum, model.hapNum), dtype=np.int64)
        n = self.currentHapNum
        out[:, self.numToHap[:n]] = self.infectious[:, :n]
        return out


def run_direct_memopt(model, iterations, sample_size, time, attempts, log_mode=LOG_LIBM):
    """The oracle's SimulatePopulation with ``memory_optimization=True`` (AddMemory pyx:264-274, AddHaplotype pyx:355-377, the
    lookup of Mutation pyx:651-660 restated op for op; see vgx_oracle.h for the one difference: bounds are honoured).  The
    counts live in the reference's program-number layout in ``model._memopt`` (a :class:`MemoptTable`); the host model's own
    ``infectious`` (indexed by haplotype in this repository) is refreshed from it after the call.  Returns the error code."""
    st = get_state(model)
    tb = getattr(model, "_memopt", None)
    if tb is None:
        tb = model._memopt = MemoptTable(model)
        if model.hapNum < 4 or model.infectious.any():
            raise ValueError("run_direct_memopt: needs at least one site and a model that has not been infected yet")
    model.events.CreateEvents(iterations)
    m = _struct(model, st, False, log_mode)
    m.infectious, m.initial_infectious = _ptr(tb.infectious), _ptr(tb.initial_infectious)
    m.memory_optimization = 1
    m.currentHapNum, m.maxHapNum, m.addMemoryNum = tb.currentHapNum, tb.maxHapNum, tb.addMemoryNum
    m.hapToNum, m.numToHap = _ptr(tb.hapToNum), _ptr(tb.numToHap)
    rc = lib().vgo_simulate_direct(C.byref(m), iterations, sample_size, float(time), attempts)
    tb.currentHapNum, tb.maxHapNum, tb.addMemoryNum = m.currentHapNum, m.maxHapNum, m.addMemoryNum
    _absorb(model, st, m)
    model.infectious[:] = tb.infectious_by_haplotype(model)
    return rc


def run_tau(model, iterations, sample_size, time, attempts, record_multievents=False, log_mode=LOG_LIBM):
    """The oracle's SimulatePopulation_tau (pyx:2293-2346); multievents in the reference's dense layout
    (propNum rows per step, zeros included) when ``record_multievents``."""
    st = get_state(model)
    model.events.CreateEvents(iterations)  # pyx:2298 -> pyx:434
    model.events.CreateEvents(iterations)  # pyx:2306
    m = _struct(model, st, False, log_mode)
    prop = lib().vgo_prop_num(C.byref(m))
    m.mev_ptr = st.mev_ptr
    if record_multievents:
        cap = st.mev_ptr + iterations * prop  # multievents.CreateEvents(iterations*propNum), ev:136-152
        old = st.mev
        st.mev = {k: np.zeros(cap, dtype=(float if k == "times" else np.int64))
                  for k in ("num", "times", "types", "haplotypes", "populations", "newHaplotypes", "newPopulations")}
        if old is not None:
            for k in st.mev:
                st.mev[k][:st.mev_ptr] = old[k][:st.mev_ptr]
        m.mev_size = cap
        m.mev_num, m.mev_times, m.mev_types = _ptr(st.mev["num"]), _ptr(st.mev["times"]), _ptr(st.mev["types"])
        m.mev_haplotypes, m.mev_populations = _ptr(st.mev["haplotypes"]), _ptr(st.mev["populations"])
        m.mev_newHaplotypes, m.mev_newPopulations = _ptr(st.mev["newHaplotypes"]), _ptr(st.mev["newPopulations"])
    rc = lib().vgo_simulate_tau(C.byref(m), iterations, sample_size, float(time), attempts)
    st.mev_ptr = m.mev_ptr
    _absorb(model, st, m)
    return rc


def tau_tries(step):
    """Rejected tries (halvings, pyx:2316-2321) of step ``step`` of the last ``run_tau`` call (test hook)."""
    f = lib().vgo_tau_tries
    f.restype, f.argtypes = C.c_int64, [C.c_int64]
    return int(f(int(step)))


def prop_num(model):
    """propNum of SimulatePopulation_tau (pyx:2301): the channels the reference draws in every step."""
    st = get_state(model)
    return int(lib().vgo_prop_num(C.byref(_struct(model, st, False, LOG_LIBM))))


def update_all_rates(model, sparse=False):
    st = get_state(model)
    m = _struct(model, st, sparse, LOG_LIBM)
    lib().vgo_update_all_rates(C.byref(m))
    model.totalRate, model.totalMigrationRate = m.totalRate, m.totalMigrationRate
    return st


class VgoGenRng(C.Structure):
    _fields_ = [("g", VgoPcg64), ("has_uint32", C.c_int64), ("uinteger", C.c_uint64)]


class VgoGenealogy(C.Structure):
    _fields_ = [("popNum", C.c_int64), ("hapNum", C.c_int64), ("sCounter", C.c_int64), ("ev_ptr", C.c_int64),
                ("ev_times", _F), ("ev_types", _I), ("ev_haplotypes", _I), ("ev_populations", _I),
                ("ev_newHaplotypes", _I), ("ev_newPopulations", _I),
                ("mev_num", _I), ("mev_times", _F), ("mev_types", _I), ("mev_haplotypes", _I), ("mev_populations", _I),
                ("mev_newHaplotypes", _I), ("mev_newPopulations", _I),
                ("infectious", _I), ("infectiousDelta", _I), ("rng", VgoGenRng),
                ("tree", _I), ("tree_pop", _I), ("times", _F),
                ("mut_cap", C.c_int64), ("mut_n", C.c_int64), ("mut_node", _I), ("mut_AS", _I), ("mut_DS", _I),
                ("mut_site", _I), ("mut_time", _F),
                ("mig_cap", C.c_int64), ("mig_n", C.c_int64), ("mig_node", _I), ("mig_old", _I), ("mig_new", _I),
                ("mig_time", _F), ("nodes_used", C.c_int64)]


def run_genealogy(model, seed=None, multievents=None):
    """The oracle's GetGenealogy (pyx:743-1000) on a host model whose forward phases were run by this oracle.
    ``seed=None`` continues the simulation's random stream (pyx:766-767).  Walks ``model.infectious`` back in place like
    the reference; returns a dict with tree, tree_pop, times, mut_* and mig_* arrays.  ``multievents``: the rows to walk for
    MULTITYPE events when the chain was not produced by this oracle (an object with the arrays num, times, types,
    haplotypes, populations, newHaplotypes, newPopulations and ``ptr``: the model's own log)."""
    st = get_state(model)
    ev = model.events
    g = VgoGenealogy()
    g.popNum, g.hapNum, g.sCounter, g.ev_ptr = model.popNum, model.hapNum, int(model.sCounter), ev.ptr
    g.ev_times, g.ev_types, g.ev_haplotypes = _ptr(ev.times), _ptr(ev.types), _ptr(ev.haplotypes)
    g.ev_populations, g.ev_newHaplotypes, g.ev_newPopulations = _ptr(ev.populations), _ptr(ev.newHaplotypes), _ptr(ev.newPopulations)
    n_mut = int((ev.types[:ev.ptr] == 3).sum())
    n_mig = int((ev.types[:ev.ptr] == 5).sum())
    if multievents is not None:
        keep = {k: np.ascontiguousarray(getattr(multievents, k)[:multievents.ptr], dtype=(np.float64 if k == "times" else np.int64))
                for k in ("num", "times", "types", "haplotypes", "populations", "newHaplotypes", "newPopulations")}
        g.mev_num, g.mev_times, g.mev_types = _ptr(keep["num"]), _ptr(keep["times"]), _ptr(keep["types"])
        g.mev_haplotypes, g.mev_populations = _ptr(keep["haplotypes"]), _ptr(keep["populations"])
        g.mev_newHaplotypes, g.mev_newPopulations = _ptr(keep["newHaplotypes"]), _ptr(keep["newPopulations"])
        n_mut += int(keep["num"][keep["types"] == 3].sum())
        n_mig += int(keep["num"][keep["types"] == 5].sum())
    elif st.mev is not None:
        mv = st.mev
        g.mev_num, g.mev_times, g.mev_types = _ptr(mv["num"]), _ptr(mv["times"]), _ptr(mv["types"])
        g.mev_haplotypes, g.mev_populations = _ptr(mv["haplotypes"]), _ptr(mv["populations"])
        g.mev_newHaplotypes, g.mev_newPopulations = _ptr(mv["newHaplotypes"]), _ptr(mv["newPopulations"])
        n_mut += int(mv["num"][:st.mev_ptr][mv["types"][:st.mev_ptr] == 3].sum())
        n_mig += int(mv["num"][:st.mev_ptr][mv["types"][:st.mev_ptr] == 5].sum())
    assert model.infectious.flags["C_CONTIGUOUS"]
    g.infectious, g.infectiousDelta = _ptr(model.infectious), _ptr(st.infectiousDelta)
    if seed is None:
        g.rng.g.state_hi, g.rng.g.state_lo, g.rng.g.inc_hi, g.rng.g.inc_lo = st.rng_final
    else:
        lib().vgo_pcg64_seed(C.byref(g.rng.g), int(seed), 0)
    nodes = 2 * int(model.sCounter) - 1
    out = {"tree": np.zeros(max(nodes, 1), dtype=np.int64), "tree_pop": np.zeros(max(nodes, 1), dtype=np.int64),
           "times": np.zeros(max(nodes, 1))}
    g.tree, g.tree_pop, g.times = _ptr(out["tree"]), _ptr(out["tree_pop"]), _ptr(out["times"])
    g.mut_cap, g.mig_cap = n_mut + 1, n_mig + nodes + 1
    for k in ("mut_node", "mut_AS", "mut_DS", "mut_site"):
        out[k] = np.zeros(g.mut_cap, dtype=np.int64)
        setattr(g, k, _ptr(out[k]))
    out["mut_time"] = np.zeros(g.mut_cap)
    g.mut_time = _ptr(out["mut_time"])
    for k in ("mig_node", "mig_old", "mig_new"):
        out[k] = np.zeros(g.mig_cap, dtype=np.int64)
        setattr(g, k, _ptr(out[k]))
    out["mig_time"] = np.zeros(g.mig_cap)
    g.mig_time = _ptr(out["mig_time"])
    rc = lib().vgo_get_genealogy(C.byref(g))
    for k in list(out):
        if k.startswith("mut_"):
            out[k] = out[k][:g.mut_n]
        elif k.startswith("mig_"):
            out[k] = out[k][:g.mig_n]
    out["rc"], out["nodes_used"] = rc, g.nodes_used
    return out


def pcg64_stream(seed, attempt, n):
    g = VgoPcg64()
    lib().vgo_pcg64_seed(C.byref(g), seed, attempt)
    state = (g.state_hi << 64) | g.state_lo
    inc = (g.inc_hi << 64) | g.inc_lo
    return state, inc, np.array([lib().vgo_pcg64_double(C.byref(g)) for _ in range(n)])
