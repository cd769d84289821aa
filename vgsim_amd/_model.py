"""Host-side model object: the MI355X engine's counterpart of VGsim's ``BirthDeathModel``.

Mirrors the construction contract, setters/validation and read-back attributes of the reference's
native class (``src/_BirthDeath.pyx:30-229`` fields and defaults, ``pyx:1187-1702`` setters, error
types and messages as pinned by the reference's ``tests/test_interface.py``).  Parameters and
compartment state live in numpy arrays with the reference's names; the two hot entry points,
``SimulatePopulation`` (pyx:396) and ``SimulatePopulation_tau`` (pyx:2293), hand them to the HIP
engine through the C ABI in ``include/vgx.h`` (``vgsim_amd/_capi.py``).  There is no CPU fallback:
without ``libvgx.so`` and a GPU the simulate calls raise.

``memory_optimization=True`` runs the same (natively sparse) engine and only maintains the haplotype table
(``_refresh_haplotype_table``).
"""
import sys

import os
import warnings

import numpy as np

from ._params import ParameterTable, need_count, need_number
from ._report import Reporting

BIRTH, DEATH, SAMPLING, MUTATION, SUSCCHANGE, MIGRATION, MULTITYPE = range(7)  # events.pxi:2-8

_RED_WARNING = '\033[41m{}\033[0m'.format('WARNING!')
_SIZE_ADVICE = ("is more than 10% different from the population size. The migration probabilities might be unrealistically high.",
                "We recommend to check your model with print_populations() method before proceding to simulation.",
                "Check the documentation file:https://vg-sim.readthedocs.io/en/latest/Migration.html for more details.")


class Events:
    """Append-only SoA event log (events.pxi:24-68)."""

    COLUMNS = ("types", "haplotypes", "populations", "newHaplotypes", "newPopulations")

    def __init__(self):
        self.size = 0
        self.ptr = 0
        self.times = np.zeros(0, dtype=float)
        for c in self.COLUMNS:
            setattr(self, c, np.zeros(0, dtype=np.int64))

    def CreateEvents(self, iterations):
        """Capacity rule of events.pxi:52-68."""
        if self.ptr == 0:
            self.size += iterations
            self.times = np.zeros(self.size, dtype=float)
            for c in self.COLUMNS:
                setattr(self, c, np.zeros(self.size, dtype=np.int64))
        elif iterations + self.ptr - self.size > 0:
            extra = iterations + self.ptr - self.size
            self.times = np.concatenate((self.times, np.zeros(extra, dtype=float)))
            for c in self.COLUMNS:
                setattr(self, c, np.concatenate((getattr(self, c), np.zeros(extra, dtype=np.int64))))
            self.size = iterations + self.ptr

    def as_array(self):
        """(6, size) float64 array, the layout ``export_chain_events`` saves (pyx:1849-1851)."""
        return np.array([self.times, self.types, self.haplotypes, self.populations,
                         self.newHaplotypes, self.newPopulations], dtype=float)


class MultiEvents:
    """Tau-leap channel log (events.pxi:105-152).

    The reference appends one row per channel per step, zeros included (pyx:2537 carries a TODO to
    stop doing so); at the shapes this engine targets that is terabytes per step, so only rows with
    ``num > 0`` are stored and each MULTITYPE event carries the [start, end) range of its rows.
    """

    COLUMNS = ("num", "types", "haplotypes", "populations", "newHaplotypes", "newPopulations")

    def __init__(self):
        self.size = 0
        self.ptr = 0
        self.times = np.zeros(0, dtype=float)
        for c in self.COLUMNS:
            setattr(self, c, np.zeros(0, dtype=np.int64))

    def extend(self, times, **cols):
        n = len(times)
        self.times = np.concatenate((self.times[:self.ptr], np.asarray(times, dtype=float)))
        for c in self.COLUMNS:
            setattr(self, c, np.concatenate((getattr(self, c)[:self.ptr], np.asarray(cols[c], dtype=np.int64))))
        self.ptr += n
        self.size = self.ptr


class Lockdowns:
    """models.pxi:52-66."""

    def __init__(self):
        self.states, self.populationsId, self.times = [], [], []

    def AddLockdown(self, state, populationId, time):
        self.states.append(bool(state))
        self.populationsId.append(int(populationId))
        self.times.append(float(time))


class Mutations:
    """models.pxi:1-32 (filled by the backward pass)."""

    def __init__(self):
        self.nodeId, self.AS, self.DS, self.site, self.time = [], [], [], [], []

    def get_mutation(self, id_mut):
        return self.nodeId[id_mut], self.DS[id_mut], self.AS[id_mut], self.site[id_mut], self.time[id_mut]


class Migrations:
    """models.pxi:35-51."""

    def __init__(self):
        self.nodeId, self.time, self.oldPop, self.newPop = [], [], [], []

    def get_migration(self, id_mig):
        return self.nodeId[id_mig], self.time[id_mig], self.oldPop[id_mig], self.newPop[id_mig]


class Recombination:
    """models.pxi:69-89: forward record of recombinant infections (filled by ``Birth``, pyx:593)."""

    def __init__(self):
        self.idevents, self.his, self.hi2s, self.nhis, self.posRecombs = [], [], [], [], []

    def AddRecombination_forward(self, Id, hi, hi2, posRecomb, nhi):
        self.idevents.append(int(Id))
        self.his.append(int(hi))
        self.hi2s.append(int(hi2))
        self.nhis.append(int(nhi))
        self.posRecombs.append(int(posRecomb))


class BirthDeathModel(ParameterTable, Reporting):
    COUNTERS = ("bCounter", "dCounter", "sCounter", "mCounter", "iCounter", "swapLockdown", "migPlus",
                "migNonPlus")

    def __init__(self, number_of_sites, populations_number, number_of_susceptible_groups, seed,
                 sampling_probability, memory_optimization, genome_length, recombination_probability):
        # order of the checks and their messages: pyx:70-98
        need_count(seed, 'seed', positive=False)
        self.user_seed = seed
        self.first_simulation = False
        for flag, what in ((sampling_probability, 'sampling probability'), (memory_optimization, 'memory optimization')):
            if flag not in (True, False):
                raise ValueError('Incorrect value of %s. Value of %s should be True or False.' % (what, what))
        self._sampling_probability = sampling_probability
        self._memory_optimization = memory_optimization
        need_count(number_of_sites, 'number of sites', positive=False)
        need_count(number_of_susceptible_groups, 'number of susceptible groups')
        need_count(populations_number, 'populations number')
        self.sites, self.susNum, self.popNum = number_of_sites, number_of_susceptible_groups, populations_number
        self.hapNum = int(4 ** self.sites)
        need_number(recombination_probability, 'recombination probability', upper=1)
        self.recombination = recombination_probability
        need_count(genome_length, 'genome length')
        self._genome_length = genome_length
        if self.sites > self._genome_length:
            raise ValueError('Incorrect value of number of sites or genome length. Genome length should be more or equal '
                             'number of sites.')
        self.sitesPosition = np.zeros(self.sites, dtype=np.int64)
        if self.sites > 1:
            self._spread_sites()
        # haplotype table of memory_optimization (pyx:105-125): starts at 4^(sites-2) slots (at least 4) and grows by as many
        block = int(4 ** max(self.sites - 2, 1)) if self._memory_optimization else 0
        self.maxHapNum = block if self._memory_optimization else self.hapNum
        self.addMemoryNum = block

        for c in self.COUNTERS:
            setattr(self, c, 0)
        self.globalInfectious = 0
        self.good_attempt = 0
        self.currentTime = 0.0
        self.tau_l = 0.01
        self.totalRate = 0.0
        self.totalMigrationRate = 0.0

        self.events = Events()
        self.multievents = MultiEvents()
        self.loc = Lockdowns()
        self.rec = Recombination()

        H, P, S = self.hapNum, self.popNum, self.susNum
        # parameters and their defaults: pyx:157-204
        self.suscType = np.zeros(H, dtype=np.int64)
        self.bRate = np.full(H, 2.0)
        self.dRate = np.full(H, 1.0)
        self.sRate = np.full(H, 0.01)
        self.mRate = np.full((H, self.sites), 0.01)
        self.susceptibility = np.zeros((H, S), dtype=float)
        self.susceptibility[:, 0] = 1.0
        self.hapMutType = np.ones((H, self.sites, 3), dtype=float)

        self.sizes = np.full(P, 1000000, dtype=np.int64)
        self.totalSusceptible = np.full(P, 1000000, dtype=np.int64)
        self.totalInfectious = np.zeros(P, dtype=np.int64)
        self.lockdownON = np.zeros(P, dtype=np.int64)
        self.susceptible = np.zeros((P, S), dtype=np.int64)
        self.susceptible[:, 0] = 1000000
        self.infectious = np.zeros((P, H), dtype=np.int64)
        self.initial_susceptible = np.zeros((P, S), dtype=np.int64)
        self.initial_infectious = np.zeros((P, H), dtype=np.int64)

        self.actualSizes = np.zeros(P, dtype=float)
        self.contactDensity = np.ones(P, dtype=float)
        self.contactDensityBeforeLockdown = np.ones(P, dtype=float)
        self.contactDensityAfterLockdown = np.zeros(P, dtype=float)
        self.startLD = np.ones(P, dtype=float)
        self.endLD = np.ones(P, dtype=float)
        self.samplingMultiplier = np.ones(P, dtype=float)
        self.suscepTransition = np.zeros((S, S), dtype=float)
        self.migrationRates = np.zeros((P, P), dtype=float)
        # haplotype <-> program-number tables (pyx:105-125): the identity without memory_optimization
        if self._memory_optimization:
            self.currentHapNum = 0
            self.hapToNum = np.zeros(H, dtype=np.int64)
            self.numToHap = np.zeros(self.maxHapNum, dtype=np.int64)
        else:
            self.currentHapNum = self.hapNum
            self.hapToNum = np.arange(H, dtype=np.int64)
            self.numToHap = np.arange(H, dtype=np.int64)

        self._engine = None  # HIP engine handle, created lazily at the first simulate call
        # backward pass (pyx:770-774 allocates the real arrays; a 1-element tree means "not simulated", pyx:1950)
        self.tree = np.zeros(1, dtype=np.int64)
        self.tree_pop = np.zeros(1, dtype=np.int64)
        self.times = np.zeros(1, dtype=float)
        self.mut, self.mig = Mutations(), Migrations()
        self._rng_position = None   # (attempt, uniforms drawn) of the last direct simulate call's random stream
        self._rng_raw = None        # raw generator state after the last genealogy pass

    # ------------------------------------------------------------------ hot-path entry points
    def _check_supported(self, method='direct'):
        if self._memory_optimization and (self.popNum > 1 or method == 'tau'):
            # Upstream's table code is only meaningful for ONE population on the direct path: AddHaplotype shifts the counts
            # of the mutating population alone (pyx:366-369), so with several populations the other populations' counts end up
            # under the wrong program numbers, and tau-leaping indexes the program-number arrays by haplotype (pyx:2351-2593),
            # reading past their maxHapNum columns.  There is nothing well-defined to reproduce; the engine's state is sparse
            # in the haplotype dimension whatever the flag says, so the run is the plain layout's trajectory with the table kept
            # as bookkeeping (``_refresh_haplotype_table``).  The reference accepts the call, so this one does too — with a
            # warning; VGX_STRICT_MEMOPT=1 turns it into the refusal.
            msg = ('memory_optimization=True with several populations or method="tau": the reference corrupts its state there '
                   '(_BirthDeath.pyx:366-369, 2351-2593); this engine keeps only the occupied haplotypes in either case and runs the '
                   'trajectory of memory_optimization=False, with the haplotype table kept as bookkeeping')
            if os.environ.get('VGX_STRICT_MEMOPT'):
                raise ValueError('memory_optimization=True is supported for one population and the direct method only. ' + msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
        if self.recombination != 0 and self.sites < 2:
            # upstream allocates the scratch vector of the recombination branch only for sites > 1 (pyx:98-102) and
            # crashes in Birth otherwise
            raise ValueError('Incorrect value of recombination probability. Recombination needs at least two sites.')

    def _refresh_haplotype_table(self):
        """``memory_optimization=True`` (pyx:105-125, 264-274, 355-377, 651-660), one population, direct method.  The
        engine's state is sparse in the haplotype dimension whatever the flag says (ordered occupancy lists),
        and the reference's table keeps program numbers in haplotype order (sorted insert, pyx:365-375), i.e. the scan
        order of the plain layout: the trajectory is that of ``memory_optimization=False`` (checked against the oracle's
        op-for-op restatement of the table code, tests/test_memopt.py).  What remains of the option is its bookkeeping,
        rebuilt here after every simulate call: ``numToHap`` = the haplotypes seen so far in ascending order, ``hapToNum``
        its inverse, ``maxHapNum`` grown in ``addMemoryNum`` steps like ``AddMemory``.  (After a Restart the reference's
        table also keeps the haplotypes that appeared only in the discarded attempts — at most 100 events each; this one
        holds those of the kept attempt.)"""
        if not self._memory_optimization:
            return
        seen = [self.numToHap[:self.currentHapNum], np.nonzero(self.initial_infectious.any(axis=0))[0],
                np.nonzero(self.infectious.any(axis=0))[0]]
        n = self.events.ptr
        seen.append(self.events.newHaplotypes[:n][self.events.types[:n] == MUTATION])
        mv = self.multievents
        seen.append(mv.newHaplotypes[:mv.ptr][mv.types[:mv.ptr] == MUTATION])
        haps = np.unique(np.concatenate([np.asarray(a, dtype=np.int64) for a in seen]))
        self.currentHapNum = len(haps)
        while self.maxHapNum < self.currentHapNum:
            self.maxHapNum = min(self.hapNum, self.maxHapNum + max(self.addMemoryNum, 1))
        self.numToHap = np.zeros(self.maxHapNum, dtype=np.int64)
        self.numToHap[:len(haps)] = haps
        self.hapToNum = np.zeros(self.hapNum, dtype=np.int64)
        self.hapToNum[haps] = np.arange(len(haps))

    def _compute_actual_sizes(self):
        """pyx:289-297 on the host, only for the CheckSizes printout (the engine computes its own)."""
        P = self.popNum
        m = self.migrationRates
        for pn1 in range(P):
            d = 1.0
            a = 0.0
            for pn2 in range(P):
                if pn1 == pn2:
                    continue
                d -= m[pn1, pn2]
                a += m[pn2, pn1] * self.sizes[pn2]
            m[pn1, pn1] = d
            a += d * self.sizes[pn1]
            self.actualSizes[pn1] = a

    def CheckSizes(self):
        """stdout contract of pyx:456-471: the effective population sizes and a warning where one is off by 10 % or more."""
        self._compute_actual_sizes()
        print('Actual sizes: ' + ''.join('%s ' % v for v in self.actualSizes.tolist()))
        off = np.nonzero(np.abs(self.actualSizes / self.sizes - 1) >= 0.1)[0]
        if len(off):
            print(_RED_WARNING, 'Actual population size in deme: ' + ", ".join(str(pn) for pn in off))
            for line in _SIZE_ADVICE:
                print("\t" + line)

    def _print_termination(self, sample_size, time):
        """Why the loop stopped, in the wording and order of pyx:420-429 / pyx:2337-2346."""
        reasons = (
            (self.totalRate == 0.0 or self.globalInfectious == 0, 'Simulation finished because no infections individuals remain!'),
            (self.events.ptr >= self.events.size, "Achieved maximal number of iterations."),
            (sample_size != -1 and self.sCounter > sample_size, "Achieved sample size."),
            (time != -1 and self.currentTime > time, "Achieved internal time limit."),
        )
        for happened, text in reasons:
            if happened:
                print(text)
        if self.sCounter <= 1:
            print(_RED_WARNING, 'Simulated less 2 samples, so genealogy will not work!')

    def _get_engine(self):
        if self._engine is None:
            from ._capi import HipEngine  # raises loudly if libvgx.so or the GPU is missing
            self._engine = HipEngine(self.sites, self.hapNum, self.popNum, self.susNum, n_replicates=1)
        return self._engine

    def SimulatePopulation(self, iterations, sample_size, time, attempts, mode='exact', kernel='auto'):
        """pyx:396-429: direct Gillespie on the GPU.  ``mode``: 'exact' = the reference's floating-point summation
        order (bit-exact log); 'fast' = order-free sums (same random stream and event semantics; include/vgx.h
        vgx_run_opts.mode).  ``kernel``: 'auto', 'wave' (one replicate per wavefront), 'quad' (four per wavefront, one per
        16-lane row: the one-class form where the model allows it, the general form otherwise), 'quadg' (the general
        four-per-wavefront form even for one-class models), 'lane' (one replicate per lane, small models), 'solo' (one trajectory
        of a small model: dense state in registers) or 'lone' (one trajectory of a large haplotype space: occupancy lists in LDS;
        vgx_run_opts.kernel)."""
        self._check_supported()
        if mode not in ('exact', 'fast', 'fast_philox'):
            raise ValueError("mode must be 'exact', 'fast' or 'fast_philox'")
        if kernel not in ('auto', 'wave', 'lane', 'quad', 'quadg', 'solo', 'lone'):
            raise ValueError("kernel must be 'auto', 'wave', 'lane', 'quad', 'quadg', 'solo' or 'lone'")
        self.events.CreateEvents(iterations)
        self.CheckSizes()
        time = float(np.float32(time))  # `float time` in the reference signature
        opts = None
        if mode != 'exact' or kernel != 'auto':
            from . import _capi
            opts = _capi.VgxRunOpts()
            opts.record_events = 1
            opts.mode = {'exact': 0, 'fast': 1, 'fast_philox': 2}[mode]
            opts.kernel = {'auto': 0, 'wave': 1, 'lane': 2, 'quad': 3, 'quadg': 4, 'solo': 5, 'lone': 6}[kernel]
        eng = self._get_engine()
        eng.simulate_direct(self, iterations, sample_size, time, attempts, opts)
        c = eng.last_counters
        self._rng_position = (int(c.reserved[1]), 2 * int(c.reserved[2])) if c.reserved[1] >= 0 else None
        self._rng_raw = None
        self._refresh_haplotype_table()
        self._print_termination(sample_size, time)

    def SimulatePopulation_tau(self, iterations, sample_size, time, attempts, record_multievents=True):
        """pyx:2293-2346: Poisson tau-leaping on the GPU.  ``record_multievents=False`` keeps the MULTITYPE records (step
        times) but not the per-channel rows a later ``genealogy()`` would walk: for long dense runs whose rows would not
        fit (the reference allocates iterations x propNum rows up front, pyx:2305)."""
        self._check_supported('tau')
        self.events.CreateEvents(iterations)   # via PrepareParameters (pyx:2298 -> pyx:434)
        self.events.CreateEvents(iterations)   # pyx:2306
        self.CheckSizes()
        time = float(np.float32(time))
        opts = None
        if not record_multievents:
            from . import _capi
            opts = _capi.VgxRunOpts()
            opts.record_events = 0
        self._get_engine().simulate_tau(self, iterations, sample_size, time, attempts, opts)
        self._rng_position, self._rng_raw = None, None
        self._refresh_haplotype_table()
        self._print_termination(sample_size, time)

    # ------------------------------------------------------------------ reporting (pyx:2048-2068, 2284, 2607-2613, 1849-1851)
    def Stats(self, time_simulation):
        """The counter block ``Simulator.simulate`` prints after every call (pyx:2048-2068)."""
        rows = [("Number of samples:", self.sCounter), ("Total number of iterations:", self.events.ptr),
                ('Success number:', self.good_attempt), ("Epidemic time:", self.currentTime),
                ('Simulation time:', time_simulation), ('Number of infections:', self.bCounter),
                ('Number of recoveries:', self.dCounter)]
        if self.sites >= 1:
            rows.append(('Number of mutations:', self.mCounter))
        if self.popNum >= 2:
            rows += [('Number of accepted migrations:', self.migPlus), ('Number of rejected migrations:', self.migNonPlus)]
        if np.any(self.suscepTransition.sum(axis=1) != 0.0):
            rows.append(('Number of immunity transitions:', self.iCounter))
        for label, value in rows:
            print(label, value)
        print('-' * 34)

    def get_proportion(self):
        """Share of loop iterations spent on rejected migrations (pyx:2284-2285)."""
        return self.migNonPlus / (self.events.ptr - 1)

    def PrintCounters(self):
        for label, name in (("Birth", "bCounter"), ("Death", "dCounter"), ("Sampling", "sCounter"), ("Mutation", "mCounter")):
            print(label + " counter(mutable): ", getattr(self, name))
        print("Immunity transition counter(mutable):", self.iCounter)
        print("Migration counter(mutable):", self.migPlus)

    def export_chain_events(self, name_file):
        np.save(name_file, self.events.as_array())

    def set_chain_events(self, name_file):
        """pyx:1705-1719 made to work: loads a ``(6, N)`` float64 chain written by ``export_chain_events`` into the
        event log (upstream assigns the float rows to typed int64 views and sets ``self.ptr`` instead of
        ``events.ptr``).  The counters a later ``genealogy()`` needs (``sCounter``) are recounted from the chain."""
        tokens = np.load(name_file + '.npy')
        if tokens.ndim != 2 or tokens.shape[0] != 6:
            raise ValueError('Incorrect chain of events: a (6, N) array is expected.')
        n = tokens.shape[1]
        while n > 0 and not tokens[:, n - 1].any():   # unused capacity of the exporting log (events.size > events.ptr)
            n -= 1
        self.events = Events()
        self.events.CreateEvents(max(n, 1))
        self.events.times[:n] = tokens[0, :n]
        for k, name in enumerate(("types", "haplotypes", "populations", "newHaplotypes", "newPopulations")):
            getattr(self.events, name)[:n] = tokens[k + 1, :n].astype(np.int64)
        self.events.ptr = n
        self.sCounter = int((self.events.types[:n] == SAMPLING).sum())

    # ------------------------------------------------------------------ backward pass (pyx:743-1000, models.pxi:1-48)
    def GetGenealogy(self, seed):
        """pyx:743-1000: coalesces the sampled lineages backwards over the event log (libvgx host code,
        ``vgx_get_genealogy``).  ``seed`` reseeds the random stream like the reference (``RndmWrapper(seed=(seed, 0))``);
        ``None`` continues the stream of the last direct ``simulate`` call (after a tau call, whose device stream is
        Philox, it starts at the beginning of stream ``(user_seed, 0)``)."""
        if self.sCounter < 2:
            print("Less than two cases were sampled...")
            print("_________________________________")
            sys.exit(0)
        from . import _capi
        out = _capi.get_genealogy(self, seed, rng_position=self._rng_position,
                                  rng_raw=self._rng_raw if seed is None else None)
        self._rng_raw = out["rng_raw"]
        self.tree, self.tree_pop, self.times = out["tree"], out["tree_pop"], out["times"]
        self.mut, self.mig = Mutations(), Migrations()
        self.mut.nodeId, self.mut.AS, self.mut.DS = out["mut_node"].tolist(), out["mut_AS"].tolist(), out["mut_DS"].tolist()
        self.mut.site, self.mut.time = out["mut_site"].tolist(), out["mut_time"].tolist()
        self.mig.nodeId, self.mig.time = out["mig_node"].tolist(), out["mig_time"].tolist()
        self.mig.oldPop, self.mig.newPop = out["mig_old"].tolist(), out["mig_new"].tolist()

    def _need_tree(self):
        if self.tree.shape[0] == 1:
            print('Genealogy was not simulated. Use VGsim.genealogy() method to simulate it.')
            sys.exit(1)

    def export_ts(self):
        """pyx:1909-1947: the genealogy as a tskit tree sequence (one tree over ``[0, genome_length)``; nodes, edges,
        migrations, sites and mutations in tskit's time direction, time 0 = the latest sample).  Needs tskit."""
        self._need_tree()
        try:
            import tskit
        except ImportError as e:
            raise ImportError('export_ts needs the tskit package, which is not installed.') from e
        tc = tskit.TableCollection()
        tc.sequence_length = self.genome_length
        t0 = self.times[0]
        for i in range(len(self.mig.nodeId)):
            node, t, old, new = self.mig.get_migration(i)
            tc.migrations.add_row(0.0, 1.0, node, old, new, t0 - t)
        for _ in range(self.popNum):
            tc.populations.add_row(None)
        n_nodes = 2 * self.sCounter - 1
        is_leaf = [1] * n_nodes
        for i in range(n_nodes - 1):
            tc.edges.add_row(0.0, self.genome_length, self.tree[i], i)
            is_leaf[self.tree[i]] = 0
        for i in range(n_nodes):
            tc.nodes.add_row(is_leaf[i], t0 - self.times[i], self.tree_pop[i])
        for pos in self.sitesPosition:
            pos = pos + 1 if pos == 0 else pos - 1 if pos == self.genome_length else pos
            tc.sites.add_row(pos, 'A')
        allele = ['A', 'T', 'C', 'G']
        for i in range(len(self.mut.nodeId)):
            node, DS, AS, site, t = self.mut.get_mutation(i)
            tc.mutations.add_row(site=site, node=node, derived_state=allele[DS], time=t0 - t)
        tc.sort()
        return tc.tree_sequence()

    def get_tree(self):  # pyx:1949-1953
        self._need_tree()
        return self.tree, self.times

    def output_tree_mutations(self):  # pyx:1725-1742
        self._need_tree()
        mut = [list(self.mut.nodeId), list(self.mut.AS), list(self.mut.site), list(self.mut.DS), list(self.mut.time)]
        times_dict = {self.events.times[i]: i for i in range(len(self.events.times))}
        populations = {}
        for time in self.times:
            populations[time] = self.events.populations[times_dict[time]]
        return self.tree, self.times, mut, populations

    def export_migrations(self, name_file, file_path):  # pyx:1744-1754
        self._need_tree()
        path = (file_path + '/' + name_file + '.tsv') if file_path is not None else (name_file + '.tsv')
        with open(path, 'w') as f_mig:
            f_mig.write("Node\tTime\tOld_population\tNew_population\n")
            for i in range(len(self.mig.nodeId)):
                f_mig.write(str(self.mig.nodeId[i]) + '\t' + str(self.mig.time[i]) + '\t' + str(self.mig.oldPop[i]) + '\t' +
                            str(self.mig.newPop[i]) + "\n")

    def output_sample_data(self):  # pyx:1756-1763
        ev = self.events
        rows = np.nonzero(ev.types[:ev.ptr] == SAMPLING)[0]
        return list(ev.times[rows]), list(ev.populations[rows]), list(ev.haplotypes[rows])

    # ------------------------------------------------------------------ log replays (pyx:1967-2045)
    def _replay(self, step_num, delta_of, sample_of=None, start=0.0):
        """Shared engine of get_data_infectious / get_data_susceptible: the reference walks the log once per query in
        interpreted loops; here every event's bin is found with one searchsorted and the per-bin sums with bincount.
        Bins the walk never reaches stay 0 exactly as upstream (Data[point+1] is only written when `point` advances)."""
        ev, mv = self.events, self.multievents
        n = ev.ptr
        time_points = [i * self.currentTime / step_num for i in range(step_num + 1)]
        tp = np.asarray(time_points, dtype=float)
        bins = np.minimum(np.searchsorted(tp, ev.times[:n], side="left"), step_num)
        # `point` only ever increases (a later event with a smaller time — a chain continued after a Restart — stays in
        # the bin reached so far)
        bins = np.maximum.accumulate(bins) if n else bins
        cols = {k: getattr(ev, k)[:n] for k in ("types", "haplotypes", "populations", "newHaplotypes", "newPopulations")}
        d = delta_of(cols, np.ones(n, dtype=np.int64)).astype(float)
        data = np.bincount(bins, weights=d, minlength=step_num + 1)
        samp = np.bincount(bins, weights=sample_of(cols, np.ones(n, dtype=np.int64)).astype(float), minlength=step_num + 1) \
            if sample_of else None
        multi = np.nonzero(cols["types"] == MULTITYPE)[0]
        if len(multi) and mv.ptr:
            lo, hi = cols["haplotypes"][multi], cols["populations"][multi]
            cnt = np.maximum(hi - lo, 0)
            rows = np.repeat(lo, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
            rbin = np.repeat(bins[multi], cnt)
            rc = {k: getattr(mv, k)[rows] for k in ("types", "haplotypes", "populations", "newHaplotypes", "newPopulations")}
            rc["_rows"] = True
            num = mv.num[rows]
            data += np.bincount(rbin, weights=delta_of(rc, num).astype(float), minlength=step_num + 1)
            if sample_of:
                samp += np.bincount(rbin, weights=sample_of(rc, num).astype(float), minlength=step_num + 1)
        last = int(bins[-1]) if n else 0
        out = np.zeros(step_num + 1)
        out[:last + 1] = start + np.cumsum(data[:last + 1])
        out_s = None
        if sample_of:
            out_s = np.zeros(step_num + 1)
            out_s[:last + 1] = np.cumsum(samp[:last + 1])
        return out, out_s, time_points

    def _lockdowns_of(self, pop):
        return [[self.loc.states[i], self.loc.times[i]] for i in range(len(self.loc.times)) if self.loc.populationsId[i] == pop]

    def get_data_infectious(self, pop, hap, step_num):
        """pyx:1967-2006, including its operator precedence (pyx:1982: recoveries and samplings of every compartment
        decrement the series)."""
        def delta(c, num):
            t = c["types"]
            here = (c["populations"] == pop) & (c["haplotypes"] == hap)
            a = (t == BIRTH) & here
            b = ~a & ((t == DEATH) | (t == SAMPLING) | ((t == MUTATION) & here))
            c3 = ~a & ~b & (t == MUTATION) & (c["newHaplotypes"] == hap) & (c["populations"] == pop)
            d4 = ~a & ~b & ~c3 & (t == MIGRATION) & (c["newPopulations"] == pop) & (c["haplotypes"] == hap)
            return num * (a.astype(np.int64) - b + c3 + d4)

        def sample(c, num):
            t = c["types"]
            a = (t == BIRTH) & (c["populations"] == pop) & (c["haplotypes"] == hap)
            return num * (~a & (t == SAMPLING))

        data, samp, tp = self._replay(step_num, delta, sample, start=float(self.initial_infectious[pop, hap]))
        return data, samp, tp, self._lockdowns_of(pop)

    def get_data_susceptible(self, pop, sus, step_num):
        """pyx:2008-2045."""
        def delta(c, num):
            t = c["types"]
            a = (t == BIRTH) & (c["populations"] == pop) & (c["newHaplotypes"] == sus)
            b = ~a & ((t == DEATH) | (t == SAMPLING) | (t == SUSCCHANGE)) & (c["populations"] == pop) & (c["newHaplotypes"] == sus)
            c3 = ~a & ~b & (t == SUSCCHANGE) & (c["haplotypes"] == sus) & (c["populations"] == pop)
            # single events test newHaplotypes == sus (pyx:2023), multievent rows haplotypes == sus (pyx:2037)
            grp = c["haplotypes"] if c.get("_rows") else c["newHaplotypes"]
            d4 = ~a & ~b & ~c3 & (t == MIGRATION) & (c["newPopulations"] == pop) & (grp == sus)
            return num * (-a.astype(np.int64) + b - c3 - d4)

        data, _, tp = self._replay(step_num, delta, None, start=float(self.initial_susceptible[pop, sus]))
        return data, tp, self._lockdowns_of(pop)

    def print_mutations(self):  # pyx:1177-1179
        print('nodeId\tDS\tAS\tsite\ttime')
        for i in range(len(self.mut.nodeId)):
            print(self.mut.get_mutation(i))

    def print_migrations(self):  # pyx:1182-1184
        print('nodeId\ttime\tsource population\ttarget population')
        for i in range(len(self.mig.nodeId)):
            print(self.mig.get_migration(i))
