"""ctypes binding of libvgx.so (C ABI in ``include/vgx.h``): the only way the Python host layer reaches
the HIP kernels.  There is no CPU fallback: a missing library or a missing GPU raises ``RuntimeError``.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VGX_LIBRARY", os.path.join(HERE, "libvgx.so"))  # VGX_LIBRARY: diagnostic builds only

_F = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int64)

VGX_OK = 0


class VgxDims(C.Structure):
    _fields_ = [("sites", C.c_int64), ("hapNum", C.c_int64), ("popNum", C.c_int64), ("susNum", C.c_int64)]


class VgxParams(C.Structure):
    _fields_ = [("bRate", _F), ("dRate", _F), ("sRate", _F), ("mRate", _F), ("hapMutType", _F),
                ("susceptibility", _F), ("suscType", _I), ("suscepTransition", _F), ("sizes", _I),
                ("contactDensityBeforeLockdown", _F), ("contactDensityAfterLockdown", _F), ("startLD", _F),
                ("endLD", _F), ("samplingMultiplier", _F), ("migrationRates", _F)]


class VgxState(C.Structure):
    _fields_ = [("susceptible", _I), ("infectious", _I), ("initial_susceptible", _I), ("initial_infectious", _I),
                ("totalSusceptible", _I), ("totalInfectious", _I), ("lockdownON", _I), ("contactDensity", _F),
                ("first_simulation", C.c_int64), ("globalInfectious", C.c_int64),
                ("bCounter", C.c_int64), ("dCounter", C.c_int64), ("sCounter", C.c_int64), ("mCounter", C.c_int64),
                ("iCounter", C.c_int64), ("swapLockdown", C.c_int64), ("migPlus", C.c_int64), ("migNonPlus", C.c_int64),
                ("good_attempt", C.c_int64),
                ("currentTime", C.c_double), ("totalRate", C.c_double), ("totalMigrationRate", C.c_double),
                ("tau_l", C.c_double), ("ev_ptr", C.c_int64), ("ev_size", C.c_int64)]


class VgxRunOpts(C.Structure):
    _fields_ = [("record_events", C.c_int64), ("max_loop_factor", C.c_int64), ("traj_points", C.c_int64),
                ("traj_t0", C.c_double), ("traj_t1", C.c_double), ("mode", C.c_int64), ("kernel", C.c_int64),
                ("reserved", C.c_int64 * 2)]


class VgxCounters(C.Structure):
    _fields_ = [("ev_ptr", C.c_int64), ("ev_first_new", C.c_int64), ("loop_iterations", C.c_int64),
                ("restarts", C.c_int64), ("lockdown_records", C.c_int64), ("error", C.c_int64),
                ("multievent_rows", C.c_int64), ("reserved", C.c_int64 * 5)]


class VgxGenealogyIO(C.Structure):
    _fields_ = [("popNum", C.c_int64), ("hapNum", C.c_int64), ("sCounter", C.c_int64), ("ev_ptr", C.c_int64),
                ("ev_times", _F), ("ev_types", _I), ("ev_haplotypes", _I), ("ev_populations", _I),
                ("ev_newHaplotypes", _I), ("ev_newPopulations", _I),
                ("mev_rows", C.c_int64), ("mev_num", _I), ("mev_times", _F), ("mev_types", _I), ("mev_haplotypes", _I),
                ("mev_populations", _I), ("mev_newHaplotypes", _I), ("mev_newPopulations", _I),
                ("infectious", _I), ("rng_state", C.c_uint64 * 4), ("rng_has_uint32", C.c_int64), ("rng_uinteger", C.c_uint64),
                ("tree", _I), ("tree_pop", _I), ("times", _F),
                ("mut_cap", C.c_int64), ("mut_n", C.c_int64), ("mut_node", _I), ("mut_AS", _I), ("mut_DS", _I),
                ("mut_site", _I), ("mut_time", _F),
                ("mig_cap", C.c_int64), ("mig_n", C.c_int64), ("mig_node", _I), ("mig_old", _I), ("mig_new", _I),
                ("mig_time", _F), ("nodes_used", C.c_int64)]


class VgxRowScan(C.Structure):
    _fields_ = [("rows", C.c_int64), ("H", C.c_int64), ("S", C.c_int64), ("infectious", _I), ("eventRates123", _F),
                ("numToHap", _I), ("bRate", _F), ("susceptibility", _F), ("rowSusceptible", _F), ("rowContact", _F), ("u", _F),
                ("birthRate", _F), ("tEvent", _F), ("hapPopRate", _F), ("susceptHapPopRate", _F), ("rowTotal", _F),
                ("chosen", _I), ("rnOut", _F)]


# every entry point include/vgx.h declares: (restype, argtypes)
_H = C.c_void_p
SIGNATURES = {
    "vgx_create": (C.c_int, [C.POINTER(VgxDims), C.c_int64, C.c_int, C.POINTER(_H)]),
    "vgx_destroy": (None, [_H]),
    "vgx_last_error": (C.c_char_p, [_H]),
    "vgx_device_count": (C.c_int, []),
    "vgx_set_params": (C.c_int, [_H, C.POINTER(VgxParams)]),
    "vgx_set_recombination": (C.c_int, [_H, C.c_double, C.c_int64, _I]),
    "vgx_get_recombinations": (C.c_int, [_H, C.c_int64, C.c_int64, _I, _I, _I, _I, _I, _I]),
    "vgx_set_state": (C.c_int, [_H, C.POINTER(VgxState)]),
    "vgx_get_state": (C.c_int, [_H, C.c_int64, C.POINTER(VgxState)]),
    "vgx_set_seeds": (C.c_int, [_H, _I]),
    "vgx_stage_tau": (C.c_int, [_H]),
    "vgx_simulate_direct": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_float, C.c_int64, C.POINTER(VgxRunOpts)]),
    "vgx_simulate_tau": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_float, C.c_int64, C.POINTER(VgxRunOpts)]),
    "vgx_get_counters": (C.c_int, [_H, C.c_int64, C.POINTER(VgxCounters)]),
    "vgx_get_counters_all": (C.c_int, [_H, _I]),
    "vgx_get_events": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_int64, _F, _I, _I, _I, _I, _I]),
    "vgx_get_lockdowns": (C.c_int, [_H, C.c_int64, C.c_int64, _I, _I, _F, _I]),
    "vgx_get_tau_tries": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]),
    "vgx_get_multievents": (C.c_int, [_H, C.c_int64, C.c_int64, _I, _F, _I, _I, _I, _I, _I, _I]),
    "vgx_get_trajectories": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "vgx_get_trajectories_int": (C.c_int, [_H, C.c_void_p]),
    "vgx_clock_mismatches": (C.c_int64, [_H]),
    "vgx_last_kernel_ms": (C.c_double, [_H]),
    "vgx_last_kernel_launches": (C.c_int64, [_H]),
    "vgx_last_direct_kernel": (C.c_int, [_H]),
    "vgx_device_bytes": (C.c_int64, [_H]),
    "vgx_get_profile": (C.c_int, [_H, C.c_int64, _I]),
    "vgx_get_genealogy": (C.c_int, [C.POINTER(VgxGenealogyIO), C.c_char_p, C.c_int64]),
    "vgx_rng_position": (None, [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_uint64 * 4)]),
    "vgx_propensity_scan": (C.c_int, [C.POINTER(VgxRowScan)]),
    "vgx_propensity_scan_bench": (C.c_int, [C.POINTER(VgxRowScan), C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vgx_propensity_scan_error": (C.c_char_p, []),
    "vgx_test_philox": (C.c_int, [C.c_int, C.POINTER(C.c_uint32 * 4), C.POINTER(C.c_uint32 * 2), C.POINTER(C.c_uint32 * 4)]),
    "vgx_test_poisson": (C.c_int, [C.c_double, C.c_int64, C.c_uint64, _I]),
    "vgx_test_div_by_const": (C.c_int, [_F, _F, C.c_int64, _F, _F, _F]),
}

_lib = None


def load_library():
    """Load libvgx.so and bind every symbol of include/vgx.h; raises if the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "vgsim_amd: the HIP extension %s is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C vgsim_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    if a.dtype == np.float64:
        return a.ctypes.data_as(_F)
    if a.dtype == np.int64:
        return a.ctypes.data_as(_I)
    raise TypeError(a.dtype)


class VgxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libvgx error %d: %s" % (code, msg))
        self.code = code


class HipEngine:
    """One libvgx engine: a model shape and ``n_replicates`` independent seeded trajectories on one GPU."""

    def __init__(self, sites, hapNum, popNum, susNum, n_replicates=1, device=0):
        self.lib = load_library()
        self.dims = VgxDims(sites, hapNum, popNum, susNum)
        self.R = int(n_replicates)
        self.handle = _H()
        rc = self.lib.vgx_create(C.byref(self.dims), self.R, int(device), C.byref(self.handle))
        if rc != VGX_OK:
            self.handle = None
            raise VgxError(rc, self.lib.vgx_last_error(None).decode())
        self.P, self.H, self.S = popNum, hapNum, susNum

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vgx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != VGX_OK:
            raise VgxError(rc, self.lib.vgx_last_error(self.handle).decode())

    # ---------------------------------------------------------------- hand-over
    def set_params(self, m):
        p = VgxParams()
        conv = []   # the arrays the pointers refer to (contiguous, of the declared dtype): alive until the call returns
        for name, ctype in VgxParams._fields_:
            a = np.ascontiguousarray(getattr(m, name), dtype=np.int64 if ctype is _I else np.float64)
            conv.append(a)
            setattr(p, name, _p(a))
        self._keep = conv
        self._check(self.lib.vgx_set_params(self.handle, C.byref(p)))
        pos = np.ascontiguousarray(getattr(m, "sitesPosition", np.zeros(0)), dtype=np.int64)
        self._check(self.lib.vgx_set_recombination(self.handle, float(getattr(m, "recombination", 0.0)),
                                                   int(getattr(m, "genome_length", 0)), _p(pos) if len(pos) else None))

    def recombinations(self, replicate=0):
        """Forward recombination records of the last direct call: (idevents, his, hi2s, nhis, posRecombs)."""
        n = C.c_int64(0)
        self._check(self.lib.vgx_get_recombinations(self.handle, replicate, 0, None, None, None, None, None, C.byref(n)))
        cols = [np.zeros(n.value, dtype=np.int64) for _ in range(5)]
        if n.value:
            self._check(self.lib.vgx_get_recombinations(self.handle, replicate, n.value, *[_p(c) for c in cols], C.byref(n)))
        return cols

    def _state_struct(self, m):
        s = VgxState()
        for name in ("susceptible", "infectious", "initial_susceptible", "initial_infectious", "totalSusceptible",
                     "totalInfectious", "lockdownON", "contactDensity"):
            setattr(s, name, _p(getattr(m, name)))
        return s

    def set_state(self, m):
        s = self._state_struct(m)
        s.first_simulation = int(m.first_simulation)
        s.globalInfectious = int(m.globalInfectious)
        for c in m.COUNTERS + ("good_attempt",):
            setattr(s, c, int(getattr(m, c)))
        s.currentTime, s.totalRate, s.totalMigrationRate, s.tau_l = m.currentTime, m.totalRate, m.totalMigrationRate, m.tau_l
        s.ev_ptr, s.ev_size = m.events.ptr, m.events.size
        self._first_call = bool(m.first_simulation)     # (the call after this snapshots the initial state: get_state reads it back then only)
        self._check(self.lib.vgx_set_state(self.handle, C.byref(s)))

    def get_state(self, m, replicate=0):
        s = self._state_struct(m)
        if not getattr(self, "_first_call", True):
            # the initial state is written by the first simulation's snapshot alone (pyx:419-424): at config 4's size copying it back
            # after every call is 2 GB of memcpy for nothing
            s.initial_susceptible = None
            s.initial_infectious = None
        self._check(self.lib.vgx_get_state(self.handle, replicate, C.byref(s)))
        m.first_simulation = bool(s.first_simulation)
        m.globalInfectious = s.globalInfectious
        for c in m.COUNTERS + ("good_attempt",):
            setattr(m, c, getattr(s, c))
        m.currentTime, m.totalRate, m.totalMigrationRate, m.tau_l = s.currentTime, s.totalRate, s.totalMigrationRate, s.tau_l

    def stage_tau(self):
        """Put the state of the last ``set_state`` on the device in the tau kernels' layout now (``vgx_stage_tau``): the next
        ``vgx_simulate_tau`` then starts from resident inputs."""
        self._check(self.lib.vgx_stage_tau(self.handle))

    def set_seeds(self, seeds):
        a = np.ascontiguousarray(np.asarray(seeds, dtype=np.int64))
        assert a.shape == (self.R,)
        self._check(self.lib.vgx_set_seeds(self.handle, _p(a)))

    def counters(self, replicate=0):
        c = VgxCounters()
        self._check(self.lib.vgx_get_counters(self.handle, replicate, C.byref(c)))
        return c

    def fetch_events(self, events, replicate, first, count):
        """Copy device log rows [first, first+count) into a host ``Events`` object at the same indices."""
        if count <= 0:
            return
        sl = slice(first, first + count)
        bufs = [np.zeros(count, dtype=np.float64)] + [np.zeros(count, dtype=np.int64) for _ in range(5)]
        self._check(self.lib.vgx_get_events(self.handle, replicate, first, count, *[_p(b) for b in bufs]))
        events.times[sl] = bufs[0]
        for name, b in zip(events.COLUMNS, bufs[1:]):
            getattr(events, name)[sl] = b

    def lockdowns(self, replicate=0, cap=4096):
        st, pp = np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.int64)
        tt = np.zeros(cap, dtype=np.float64)
        n = C.c_int64(0)
        self._check(self.lib.vgx_get_lockdowns(self.handle, replicate, cap, _p(st), _p(pp), _p(tt), C.byref(n)))
        k = min(n.value, cap)
        return st[:k], pp[:k], tt[:k]

    # ---------------------------------------------------------------- the hot path for one host model
    def simulate_direct(self, m, iterations, sample_size, time, attempts, opts=None):
        """``BirthDeathModel.SimulatePopulation`` (pyx:396-429) for a host model object: upload, run the
        persistent kernel, read the model back exactly as the reference would leave it."""
        self.set_params(m)
        self.set_state(m)
        self.set_seeds([m.user_seed] * self.R)
        rc = self.lib.vgx_simulate_direct(self.handle, iterations, sample_size, float(time), attempts,
                                          C.byref(opts) if opts is not None else None)
        self._check(rc)
        self._absorb(m)

    def simulate_tau(self, m, iterations, sample_size, time, attempts, opts=None):
        """``BirthDeathModel.SimulatePopulation_tau`` (pyx:2293-2346)."""
        self.set_params(m)
        self.set_state(m)
        self.set_seeds([m.user_seed] * self.R)
        rc = self.lib.vgx_simulate_tau(self.handle, iterations, sample_size, float(time), attempts,
                                       C.byref(opts) if opts is not None else None)
        self._check(rc)
        mev_base = m.multievents.ptr   # rows of earlier tau calls stay in the log
        self._absorb(m, tau=True, mev_base=mev_base)

    def multievents(self, replicate=0):
        """Rows (num > 0 only) of the last tau call: dict of arrays num, times, types, haplotypes, ..."""
        n = C.c_int64(0)
        self._check(self.lib.vgx_get_multievents(self.handle, replicate, 0, None, None, None, None, None, None, None, C.byref(n)))
        k = n.value
        cols = {name: np.zeros(k, dtype=np.int64) for name in ("num", "types", "haplotypes", "populations", "newHaplotypes", "newPopulations")}
        times = np.zeros(k, dtype=np.float64)
        if k:
            self._check(self.lib.vgx_get_multievents(self.handle, replicate, k, _p(cols["num"]), _p(times), _p(cols["types"]),
                                                     _p(cols["haplotypes"]), _p(cols["populations"]), _p(cols["newHaplotypes"]),
                                                     _p(cols["newPopulations"]), C.byref(n)))
        cols["times"] = times
        return cols

    def counters_all(self):
        """``[R, 4]`` int64: ev_ptr, loop_iterations, restarts, tau events drawn of every replicate."""
        out = np.zeros((self.R, 4), dtype=np.int64)
        self._check(self.lib.vgx_get_counters_all(self.handle, _p(out)))
        return out

    def _absorb(self, m, replicate=0, tau=False, mev_base=0):
        self.get_state(m, replicate)
        c = self.counters(replicate)
        first = c.ev_first_new
        self.fetch_events(m.events, replicate, first, c.ev_ptr - first)
        m.events.ptr = c.ev_ptr
        if tau:
            rows = self.multievents(replicate)
            if len(rows["times"]) == 0:    # nothing recorded (record_multievents=False or no event drawn): empty row ranges
                m.events.haplotypes[first:c.ev_ptr] = 0
                m.events.populations[first:c.ev_ptr] = 0
            rows = canonical_multievents(rows, m.events.haplotypes[first:c.ev_ptr], m.events.populations[first:c.ev_ptr],
                                         m.sites, m.susNum)
            if c.restarts > 0:  # Restart rewinds multievents.ptr too (pyx:716)
                m.multievents.ptr = 0
                mev_base = 0
            # MULTITYPE records carry the [start, end) range of their rows (pyx:2325), here within the sparse log
            m.events.haplotypes[first:c.ev_ptr] += mev_base
            m.events.populations[first:c.ev_ptr] += mev_base
            m.multievents.extend(rows.pop("times"), **rows)
            self.last_events_drawn = c.reserved[0]
        st, pp, tt = self.lockdowns(replicate)
        for k in range(len(st)):
            m.loc.AddLockdown(st[k], pp[k], tt[k])
        if not tau and getattr(m, "recombination", 0.0) and hasattr(m, "rec"):
            for row in zip(*self.recombinations(replicate)):
                m.rec.AddRecombination_forward(*row)
        self.last_counters = c

    def tau_tries(self, replicate, first, count):
        """Rejected tries (halvings of tau_l, pyx:2316-2321) of steps [first, first + count) of the last tau call."""
        out = np.zeros(count, dtype=np.int32)
        self._check(self.lib.vgx_get_tau_tries(self.handle, replicate, first, count, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    @property
    def last_kernel_ms(self):
        return self.lib.vgx_last_kernel_ms(self.handle)

    @property
    def last_kernel(self):
        """Name of the kernel the last direct call ran on (vgx_run_opts.kernel)."""
        return {1: "wave", 2: "lane", 3: "quad", 4: "quadg", 5: "solo", 6: "lone"}[self.lib.vgx_last_direct_kernel(self.handle)]

    @property
    def device_bytes(self):
        return self.lib.vgx_device_bytes(self.handle)


def canonical_multievents(rows, starts, ends, sites, susNum):
    """Rows of one tau call in the reference's order and granularity.  The device appends a step's rows in whatever
    order its threads get there (and one row per drawn transmission / mutant / migrant); the reference writes one row
    per channel in a fixed order (UpdateCompartmentCounts_tau, pyx:2540-2593: all migrations by (source, target,
    group, haplotype); then per population the immunity transitions, then per haplotype recovery, sampling, mutations
    by (site, derived state), transmissions by group).  Sorting each step's rows by that key and merging equal channels
    gives the reference's rows with ``num > 0`` — what the backward pass (pyx:873-994) walks — independent of
    scheduling.  ``starts`` / ``ends`` (the MULTITYPE events' row ranges, relative to this call) are updated in place."""
    n = len(rows["times"])
    if n == 0:
        return rows
    num, typ, hap = rows["num"], rows["types"], rows["haplotypes"]
    pop, nh, npop = rows["populations"], rows["newHaplotypes"], rows["newPopulations"]
    step = np.searchsorted(np.asarray(ends), np.arange(n), side="right")
    mig = typ == 5
    k1 = np.where(mig, 0, 1)
    k2 = pop.copy()                                  # source population / population
    k3 = np.where(mig, npop, np.where(typ == 4, 0, 1))
    k4 = np.where(mig, nh, hap)                      # migration: group; else source group / haplotype
    k5 = np.zeros(n, dtype=np.int64)
    k5[mig] = hap[mig]
    k5[typ == 4] = nh[typ == 4]
    k5[typ == 1] = 0
    k5[typ == 2] = 1
    mu = typ == 3
    if mu.any():
        d = np.abs(nh[mu] - hap[mu])
        low = np.zeros(d.shape, dtype=np.int64)      # digit position from the least significant end
        t = d.copy()
        while (t >= 4).any():
            big = t >= 4
            t[big] //= 4
            low[big] += 1
        digit4 = 4 ** low
        AS, DS = (hap[mu] // digit4) % 4, (nh[mu] // digit4) % 4
        k5[mu] = 2 + (sites - 1 - low) * 3 + np.where(DS > AS, DS - 1, DS)
    k5[typ == 0] = 2 + 3 * sites + nh[typ == 0]
    order = np.lexsort((k5, k4, k3, k2, k1, step))
    key = np.stack([step, k1, k2, k3, k4, k5])[:, order]
    first = np.ones(n, dtype=bool)
    first[1:] = (key[:, 1:] != key[:, :-1]).any(axis=0)
    idx = np.nonzero(first)[0]
    out = {c: rows[c][order][idx] for c in ("types", "haplotypes", "populations", "newHaplotypes", "newPopulations")}
    out["num"] = np.add.reduceat(num[order], idx)
    out["times"] = rows["times"][order][idx]
    per_step = np.bincount(key[0, idx], minlength=len(starts))
    e = np.cumsum(per_step)
    starts[:] = e - per_step
    ends[:] = e
    return out


def get_genealogy(m, seed=None, rng_position=None, rng_raw=None):
    """``BirthDeathModel.GetGenealogy(seed)`` (pyx:743-1000) on a host model: the backward pass over ``m.events`` (and
    ``m.multievents`` for tau chains) in libvgx's host code.  ``seed``: reseeds the stream as the reference does
    (``RndmWrapper(seed=(seed, 0))``, pyx:766-767); ``None`` continues from ``rng_position`` = (attempt, draws) of the
    simulation's stream, or from ``rng_raw`` = the raw generator state a previous pass returned (``out["rng_raw"]``).
    Walks ``m.infectious`` back in place; returns a dict of arrays."""
    lib = load_library()
    ev, mv = m.events, m.multievents
    io = VgxGenealogyIO()
    io.popNum, io.hapNum, io.sCounter, io.ev_ptr = m.popNum, m.hapNum, int(m.sCounter), int(ev.ptr)
    io.ev_times, io.ev_types, io.ev_haplotypes = _p(ev.times), _p(ev.types), _p(ev.haplotypes)
    io.ev_populations, io.ev_newHaplotypes, io.ev_newPopulations = _p(ev.populations), _p(ev.newHaplotypes), _p(ev.newPopulations)
    types = ev.types[:ev.ptr]
    n_mut, n_mig = int((types == 3).sum()), int((types == 5).sum())
    if mv.ptr > 0:
        io.mev_rows = int(mv.ptr)
        io.mev_num, io.mev_times, io.mev_types = _p(mv.num), _p(mv.times), _p(mv.types)
        io.mev_haplotypes, io.mev_populations = _p(mv.haplotypes), _p(mv.populations)
        io.mev_newHaplotypes, io.mev_newPopulations = _p(mv.newHaplotypes), _p(mv.newPopulations)
        n_mut += int(mv.num[:mv.ptr][mv.types[:mv.ptr] == 3].sum())
        n_mig += int(mv.num[:mv.ptr][mv.types[:mv.ptr] == 5].sum())
    if not m.infectious.flags["C_CONTIGUOUS"]:
        raise ValueError("infectious must be C-contiguous")
    io.infectious = _p(m.infectious)
    pos = (C.c_uint64 * 4)()
    if seed is not None:
        lib.vgx_rng_position(int(seed), 0, 0, C.byref(pos))
    elif rng_raw is not None:
        for i in range(4):
            pos[i] = rng_raw[i]
        io.rng_has_uint32, io.rng_uinteger = int(rng_raw[4]), int(rng_raw[5])
    else:
        att, draws = rng_position if rng_position is not None else (0, 0)
        lib.vgx_rng_position(int(m.user_seed), int(att), int(draws), C.byref(pos))
    for i in range(4):
        io.rng_state[i] = pos[i]
    nodes = max(2 * int(m.sCounter) - 1, 1)
    out = {"tree": np.zeros(nodes, dtype=np.int64), "tree_pop": np.zeros(nodes, dtype=np.int64), "times": np.zeros(nodes)}
    io.tree, io.tree_pop, io.times = _p(out["tree"]), _p(out["tree_pop"]), _p(out["times"])
    io.mut_cap, io.mig_cap = n_mut + 1, n_mig + nodes + 1
    for k in ("mut_node", "mut_AS", "mut_DS", "mut_site"):
        out[k] = np.zeros(io.mut_cap, dtype=np.int64)
        setattr(io, k, _p(out[k]))
    out["mut_time"] = np.zeros(io.mut_cap)
    io.mut_time = _p(out["mut_time"])
    for k in ("mig_node", "mig_old", "mig_new"):
        out[k] = np.zeros(io.mig_cap, dtype=np.int64)
        setattr(io, k, _p(out[k]))
    out["mig_time"] = np.zeros(io.mig_cap)
    io.mig_time = _p(out["mig_time"])
    err = C.create_string_buffer(512)
    rc = lib.vgx_get_genealogy(C.byref(io), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode() or "vgx_get_genealogy failed (%d)" % rc)
    for k in list(out):
        if k.startswith("mut_"):
            out[k] = out[k][:io.mut_n]
        elif k.startswith("mig_"):
            out[k] = out[k][:io.mig_n]
    out["nodes_used"] = int(io.nodes_used)
    out["rng_raw"] = tuple(int(io.rng_state[i]) for i in range(4)) + (int(io.rng_has_uint32), int(io.rng_uinteger))
    return out


def propensity_scan(infectious, rates123, numToHap, bRate, susceptibility, rowSusceptible, rowContact, u, bench_rows=0, repeats=3):
    """The dense propensity row pass (``vgx_propensity_scan``, K3): returns a dict of the output arrays.  With ``bench_rows``
    the first row is replicated that many times in HBM and timed (``ms_update``, ``ms_choose`` in the result)."""
    lib = load_library()
    inf = np.ascontiguousarray(infectious, dtype=np.int64)
    rows, H = inf.shape
    S = int(np.asarray(susceptibility).reshape(H, -1).shape[1])
    n = bench_rows if bench_rows else rows

    def fit(a, shape):   # bench mode: per-row inputs are given per synthetic row, or repeated
        a = np.ascontiguousarray(a, dtype=np.float64).reshape((-1,) + shape)
        return np.ascontiguousarray(np.resize(a, (n,) + shape)) if len(a) != n else a
    keep = dict(infectious=inf, eventRates123=np.ascontiguousarray(rates123, dtype=np.float64).reshape(rows, H, 3),
                numToHap=np.ascontiguousarray(numToHap, dtype=np.int64), bRate=np.ascontiguousarray(bRate, dtype=np.float64),
                susceptibility=np.ascontiguousarray(susceptibility, dtype=np.float64).reshape(H, S),
                rowSusceptible=fit(rowSusceptible, (S,)), rowContact=fit(rowContact, ()), u=fit(u, ()))
    ro = 1 if bench_rows else rows
    out = dict(birthRate=np.zeros((ro, H)), tEvent=np.zeros((ro, H)), hapPopRate=np.zeros((ro, H)),
               susceptHapPopRate=np.zeros((ro, H, S)), rowTotal=np.zeros(ro), chosen=np.zeros(ro, dtype=np.int64), rnOut=np.zeros(ro))
    io = VgxRowScan()
    io.rows, io.H, io.S = rows, H, S
    for k, v in list(keep.items()) + list(out.items()):
        setattr(io, k, _p(v))
    if bench_rows:
        a, b = C.c_double(0), C.c_double(0)
        rc = lib.vgx_propensity_scan_bench(C.byref(io), int(bench_rows), int(repeats), C.byref(a), C.byref(b))
        out["ms_update"], out["ms_choose"] = a.value, b.value
    else:
        rc = lib.vgx_propensity_scan(C.byref(io))
    if rc != VGX_OK:
        raise VgxError(rc, lib.vgx_propensity_scan_error().decode())
    return out
