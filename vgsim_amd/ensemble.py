"""Replicate ensembles: many independent seeded trajectories of one model on one GPU, one GPU per process.

The reference has no notion of an ensemble (users start separate ``Simulator`` processes by hand, SURVEY.md
§8e); this is the data-parallel axis the engine shards: replicate r of rank k runs on GPU k as its own
persistent wavefront, nothing is exchanged while simulating, and the only collective is one gather of the
fixed-shape summary trajectories ``[replicates, points, populations, 2]`` to rank 0 (RCCL over xGMI when the
process group uses the ``nccl`` backend, gloo on CPU in the tests).
"""
import ctypes as C

import numpy as np

from . import _capi


class EnsembleResult:
    """Per-replicate outcome of one ensemble call (numpy arrays of length n_replicates)."""

    def __init__(self, R):
        z = lambda: np.zeros(R, dtype=np.int64)  # noqa: E731
        self.events = z()           # events.ptr
        self.loop_iterations = z()  # incl. rejected migrations
        self.restarts = z()
        self.kernel_ms = 0.0

    @property
    def total_events(self):
        return int(self.events.sum())


class Ensemble:
    def __init__(self, simulator, n_replicates, seeds=None, device=0):
        """``simulator``: a configured ``vgsim_amd.Simulator`` (or its ``.simulation`` model) giving parameters
        and the common start state; ``seeds``: one user seed per replicate (default seed, seed+1, ...)."""
        self.model = getattr(simulator, "simulation", simulator)
        m = self.model
        m._check_supported()
        self.R = int(n_replicates)
        self.engine = _capi.HipEngine(m.sites, m.hapNum, m.popNum, m.susNum, n_replicates=self.R, device=device)
        self.seeds = np.arange(m.user_seed, m.user_seed + self.R, dtype=np.int64) if seeds is None \
            else np.ascontiguousarray(seeds, dtype=np.int64)
        assert self.seeds.shape == (self.R,)
        self.traj_shape = None

    def close(self):
        self.engine.close()

    def simulate(self, iterations, sample_size=None, epidemic_time=-1, attempts=200, record_events=False,
                 traj_points=0, traj_window=(0.0, 1.0), seeds=None, mode='exact', kernel='auto'):
        """Direct Gillespie for every replicate from the model's current state (``SimulatePopulation`` semantics
        per replicate, pyx:396-429).  ``mode``: 'exact' (reference summation order, bit-exact) or 'fast'
        (order-free sums).  Returns an :class:`EnsembleResult`."""
        if mode not in ('exact', 'fast', 'fast_philox'):
            raise ValueError("mode must be 'exact', 'fast' or 'fast_philox'")
        m, eng = self.model, self.engine
        if seeds is not None:
            self.seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        if sample_size is None:
            sample_size = iterations
        if epidemic_time is None:
            epidemic_time = -1
        # Events.CreateEvents bookkeeping on a scratch copy of the counters (the host model keeps its own log)
        ptr, size = m.events.ptr, m.events.size
        size = size + iterations if ptr == 0 else max(size, ptr + iterations)
        eng.set_params(m)
        saved = (m.events.ptr, m.events.size)
        m.events.size = size
        try:
            eng.set_state(m)
        finally:
            m.events.ptr, m.events.size = saved
        eng.set_seeds(self.seeds)
        o = _capi.VgxRunOpts()
        o.record_events = 1 if record_events else 0
        o.traj_points = int(traj_points)
        o.traj_t0, o.traj_t1 = float(traj_window[0]), float(traj_window[1])
        o.mode = {'exact': 0, 'fast': 1, 'fast_philox': 2}[mode]
        o.kernel = {'auto': 0, 'wave': 1, 'lane': 2, 'quad': 3, 'quadg': 4, 'solo': 5, 'lone': 6}[kernel]
        rc = eng.lib.vgx_simulate_direct(eng.handle, int(iterations), int(sample_size), float(np.float32(epidemic_time)),
                                         int(attempts), C.byref(o))
        eng._check(rc)
        res = EnsembleResult(self.R)
        call = eng.counters_all()
        res.events[:], res.loop_iterations[:], res.restarts[:] = call[:, 0], call[:, 1], call[:, 2]
        res.kernel_ms = eng.last_kernel_ms
        self.traj_shape = (self.R, int(traj_points), m.popNum, 2) if traj_points > 0 else None
        return res

    def simulate_tau(self, iterations, sample_size=None, epidemic_time=-1, attempts=200, record_events=False, seeds=None):
        """Poisson tau-leaping for every replicate from the model's current state (``SimulatePopulation_tau``
        semantics per replicate, pyx:2293-2346).  ``EnsembleResult.events`` counts MULTITYPE records (steps);
        ``events_drawn`` the sum of the drawn channel multiplicities."""
        m, eng = self.model, self.engine
        if seeds is not None:
            self.seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        if sample_size is None:
            sample_size = iterations
        if epidemic_time is None:
            epidemic_time = -1
        ptr, size = m.events.ptr, m.events.size
        for _ in range(2):  # CreateEvents is called twice on the tau path (pyx:2298 -> pyx:434, pyx:2306)
            size = size + iterations if ptr == 0 else max(size, ptr + iterations)
        eng.set_params(m)
        saved = (m.events.ptr, m.events.size)
        m.events.size = size
        try:
            eng.set_state(m)
        finally:
            m.events.ptr, m.events.size = saved
        eng.set_seeds(self.seeds)
        o = _capi.VgxRunOpts()
        o.record_events = 1 if record_events else 0
        rc = eng.lib.vgx_simulate_tau(eng.handle, int(iterations), int(sample_size), float(np.float32(epidemic_time)),
                                      int(attempts), C.byref(o))
        eng._check(rc)
        res = EnsembleResult(self.R)
        call = eng.counters_all()
        res.events[:], res.loop_iterations[:], res.restarts[:] = call[:, 0], call[:, 1], call[:, 2]
        res.events_drawn = call[:, 3].copy()
        res.kernel_ms = eng.last_kernel_ms
        self.traj_shape = None
        return res

    def replicate_state(self, replicate):
        """A host model object holding the state (compartments, counters, times) of one replicate."""
        import copy
        m = copy.copy(self.model)
        for name in ("susceptible", "infectious", "initial_susceptible", "initial_infectious", "totalSusceptible",
                     "totalInfectious", "lockdownON", "contactDensity"):
            setattr(m, name, getattr(self.model, name).copy())
        self.engine.get_state(m, replicate)
        return m

    def replicate_events(self, replicate):
        """(6, n) float64 event chain of one replicate (needs ``record_events=True``)."""
        from ._model import Events
        c = self.engine.counters(replicate)
        ev = Events()
        ev.CreateEvents(max(int(c.ev_ptr), 1))
        self.engine.fetch_events(ev, replicate, c.ev_first_new, c.ev_ptr - c.ev_first_new)
        ev.ptr = c.ev_ptr
        return ev.as_array()[:, :c.ev_ptr]

    def genealogy(self, replicate, seed):
        """Backward pass (``GetGenealogy``, pyx:743-1000) over the recorded chain of one replicate of the last direct
        ``simulate(record_events=True)`` call; returns the dict of ``_capi.get_genealogy`` (tree, times, mut_*, mig_*)."""
        from ._model import Events
        m = self.replicate_state(replicate)
        chain = self.replicate_events(replicate)
        ev = Events()
        ev.CreateEvents(max(chain.shape[1], 1))
        ev.times[:chain.shape[1]] = chain[0]
        for k, name in enumerate(("types", "haplotypes", "populations", "newHaplotypes", "newPopulations")):
            getattr(ev, name)[:chain.shape[1]] = chain[k + 1].astype(np.int64)
        ev.ptr = chain.shape[1]
        m.events = ev
        c = self.engine.counters(replicate)
        pos = (int(c.reserved[1]), 2 * int(c.reserved[2])) if c.reserved[1] >= 0 else None
        m.user_seed = int(self.seeds[replicate])
        return _capi.get_genealogy(m, seed, rng_position=pos)

    def trajectories(self, out=None):
        """Summary trajectories of the last call, ``[R, T, P, 2]`` float64 (infectious, susceptible per population).
        ``out`` may be a CUDA torch tensor (filled on the device, no host round trip) or None (numpy)."""
        if self.traj_shape is None:
            raise RuntimeError("the last simulate() call did not record trajectories (traj_points=0)")
        eng = self.engine
        if out is None:
            a = np.empty(self.traj_shape, dtype=np.float64)
            eng._check(eng.lib.vgx_get_trajectories(eng.handle, a.ctypes.data_as(C.c_void_p), 0))
            return a
        assert tuple(out.shape) == self.traj_shape and out.is_contiguous() and str(out.dtype) == "torch.float64"
        eng._check(eng.lib.vgx_get_trajectories(eng.handle, C.c_void_p(out.data_ptr()), 1 if out.is_cuda else 0))
        return out

    def gather_trajectories(self, dst=0, out=None, async_op=False, wire_dtype=None, device=None):
        """One collective for the whole ensemble: every rank's ``[R, T, P, 2]`` block to rank ``dst``
        (``torch.distributed.gather``; backend nccl = RCCL over xGMI, gloo on CPU).  Returns the stacked
        ``[world, R, T, P, 2]`` tensor on ``dst`` and None elsewhere; ``out`` may be a preallocated result tensor on
        ``dst``.  With ``async_op=True`` the trajectories are first copied out of the engine (so the next ``simulate``
        may overwrite them) and a :class:`PendingGather` is returned: the transfer overlaps the next step and
        ``.wait()`` gives the result.  ``wire_dtype=torch.int32`` sends the compartment totals as 32-bit integers (they are
        whole numbers; refused unless every population size is below 2^31): half the bytes on every xGMI link and in the
        result on ``dst``, which then has that dtype.  Without a process group (one rank) the result is a host tensor, or — with
        ``device='cuda'`` — a tensor on the engine's GPU, filled there: where an RCCL gather leaves it on rank ``dst``."""
        import torch
        import torch.distributed as dist

        def narrow(t):
            if wire_dtype is None or wire_dtype == torch.float64:
                return t
            if wire_dtype != torch.int32:
                raise ValueError("wire_dtype must be None, torch.float64 or torch.int32")
            if int(np.max(self.model.sizes)) >= 2 ** 31:
                raise ValueError("wire_dtype=int32 needs population sizes below 2^31")
            return t.to(torch.int32)

        if not (dist.is_available() and dist.is_initialized()):
            if device is not None and str(device).startswith("cuda"):
                dev = torch.device("cuda", torch.cuda.current_device()) if str(device) == "cuda" else torch.device(device)
                res = narrow(self.trajectories(torch.empty(self.traj_shape, dtype=torch.float64, device=dev)))[None]
            else:
                res = narrow(torch.from_numpy(self.trajectories()))[None]
            return PendingGather(None, res, None) if async_op else res
        backend = dist.get_backend()
        if backend == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            if wire_dtype == torch.int32:      # narrowed on the device straight from the engine's buffer: no f64 copy
                narrow(torch.empty(0))         # (the checks)
                mine = torch.empty(self.traj_shape, dtype=torch.int32, device=dev)
                eng = self.engine
                eng._check(eng.lib.vgx_get_trajectories_int(eng.handle, C.c_void_p(mine.data_ptr())))
            else:
                mine = narrow(self.trajectories(torch.empty(self.traj_shape, dtype=torch.float64, device=dev)))
        else:
            mine = narrow(torch.from_numpy(self.trajectories()))
        world, rank = dist.get_world_size(), dist.get_rank()
        if rank == dst:   # gather straight into the rows of the result: no second copy of [world, R, T, P, 2]
            if out is None:
                out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
            assert tuple(out.shape) == (world,) + tuple(mine.shape) and out.is_contiguous()
            assert out.dtype == mine.dtype and out.device == mine.device, "`out` must have the wire dtype and live where the collective runs"
            work = dist.gather(mine, list(out.unbind(0)), dst=dst, async_op=async_op)
        else:
            out = None
            work = dist.gather(mine, None, dst=dst, async_op=async_op)
        return PendingGather(work, out, mine) if async_op else out


class PendingGather:
    """Handle of an asynchronous trajectory gather: keeps the send/receive buffers alive until ``wait()``."""

    def __init__(self, work, result, keep):
        self.work, self.result, self.keep = work, result, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        self.keep = None
        return self.result
