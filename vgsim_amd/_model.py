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

import numpy as np

from ._report import Reporting

BIRTH, DEATH, SAMPLING, MUTATION, SUSCCHANGE, MIGRATION, MULTITYPE = range(7)  # events.pxi:2-8


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


class BirthDeathModel(Reporting):
    COUNTERS = ("bCounter", "dCounter", "sCounter", "mCounter", "iCounter", "swapLockdown", "migPlus",
                "migNonPlus")

    def __init__(self, number_of_sites, populations_number, number_of_susceptible_groups, seed,
                 sampling_probability, memory_optimization, genome_length, recombination_probability):
        # validation order and messages: pyx:70-98
        self.check_amount(seed, 'seed', zero=False)
        self.user_seed = seed
        self.first_simulation = False
        if sampling_probability != True and sampling_probability != False:  # noqa: E712 (reference semantics)
            raise ValueError('Incorrect value of sampling probability. Value of sampling probability should be True or False.')
        self._sampling_probability = sampling_probability
        if memory_optimization != True and memory_optimization != False:  # noqa: E712
            raise ValueError('Incorrect value of memory optimization. Value of memory optimization should be True or False.')
        self._memory_optimization = memory_optimization

        self.check_amount(number_of_sites, 'number of sites', zero=False)
        self.sites = number_of_sites
        self.hapNum = int(4 ** self.sites)
        self.check_amount(number_of_susceptible_groups, 'number of susceptible groups')
        self.susNum = number_of_susceptible_groups
        self.check_amount(populations_number, 'populations number')
        self.popNum = populations_number

        self.check_value(recombination_probability, 'recombination probability', edge=1)
        self.recombination = recombination_probability
        self.check_amount(genome_length, 'genome length')
        self._genome_length = genome_length
        self.sitesPosition = np.zeros(self.sites, dtype=np.int64)
        if self.sites > self._genome_length:
            raise ValueError('Incorrect value of number of sites or genome length. Genome length should be more or equal number of sites.')
        if self.sites > 1:
            for s in range(self.sites):
                self.sitesPosition[s] = int(s * self._genome_length / (self.sites - 1))

        if self._memory_optimization:
            if self.sites > 2:
                self.maxHapNum = int(4 ** (self.sites - 2))
                self.addMemoryNum = int(4 ** (self.sites - 2))
            else:
                self.maxHapNum = 4
                self.addMemoryNum = 4
        else:
            self.maxHapNum = self.hapNum
            self.addMemoryNum = 0

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

    # ------------------------------------------------------------------ haplotype patterns (pyx:1187-1267)
    def calculate_indexes(self, indexes_list, edge):
        if isinstance(indexes_list, list):
            indexes = set()
            for i in indexes_list:
                indexes.update(self.calculate_index(i, edge))
        else:
            indexes = set(self.calculate_index(indexes_list, edge))
        return indexes

    def calculate_index(self, index, edge):
        if isinstance(index, str):
            haplotypes = [index]
            for s in range(self.sites):
                for i in range(len(haplotypes)):
                    old = haplotypes[i]
                    if old[s] == "*":
                        for letter in ('A', 'T', 'C', 'G'):
                            haplotypes.append(old.replace("*", letter, 1))
            haplotypes = [h for h in haplotypes if h.count("*") == 0]
            return [self.calculate_haplotype_from_string(h) for h in haplotypes]
        elif isinstance(index, int):
            return [index]
        else:
            return range(edge)

    def calculate_string_from_haplotype(self, hapNum):
        letters = ["A", "T", "C", "G"]
        string = ""
        for _ in range(self.sites):
            string = letters[hapNum % 4] + string
            hapNum = hapNum // 4
        return string

    def calculate_haplotype_from_string(self, string):
        string = string[::-1]
        haplotype = 0
        for s in range(self.sites):
            haplotype += {"A": 0, "T": 1, "C": 2, "G": 3}.get(string[s], 0) * (4 ** s)
        return haplotype

    def calculate_allele(self, haplotype, site):
        allele = 0
        for _ in range(self.sites - site):
            allele = haplotype % 4
            haplotype = haplotype // 4
        return allele

    # ------------------------------------------------------------------ validation (pyx:1298-1377)
    def check_amount(self, amount, smth, zero=True):
        if isinstance(amount, int) == False:  # noqa: E712
            raise TypeError('Incorrect type of ' + smth + '. Type should be int.')
        elif amount <= 0 and zero:
            raise ValueError('Incorrect value of ' + smth + '. Value should be more 0.')
        elif amount < 0 and zero == False:  # noqa: E712
            raise ValueError('Incorrect value of ' + smth + '. Value should be more or equal 0.')

    def check_value(self, value, smth, edge=None, none=False):
        if none:
            if isinstance(value, (int, float)) == False and value is not None:  # noqa: E712
                raise TypeError('Incorrect type of ' + smth + '. Type should be int or float or None.')
        else:
            if isinstance(value, (int, float)) == False:  # noqa: E712
                raise TypeError('Incorrect type of ' + smth + '. Type should be int or float.')
        if isinstance(value, (int, float)):
            if edge is None:
                if value < 0:
                    raise ValueError('Incorrect value of ' + smth + '. Value should be more or equal 0.')
            elif value < 0 or value > edge:
                raise ValueError('Incorrect value of ' + smth + '. Value should be more or equal 0 and equal or less ' + str(edge) + '.')

    def check_indexes(self, index, edge, smth, hap=False, none=True):
        if isinstance(index, list):
            for i in index:
                self.check_index(i, edge, smth, hap=hap, none=none)
        else:
            self.check_index(index, edge, smth, hap=hap, none=none)

    def check_index(self, index, edge, smth, hap=False, none=True):
        if none == False and index is None:  # noqa: E712
            raise TypeError('Incorrect type of ' + smth + '. Type should be int.')
        elif isinstance(index, int):
            if index < 0 or index >= edge:
                raise IndexError('There are no such ' + smth + '!')
        elif isinstance(index, str) and hap:
            if sum(index.count(ch) for ch in "ATCG*") != self.sites:
                raise ValueError('Incorrect haplotype. Haplotype should contain only \"A\", \"T\", \"C\", \"G\", \"*\" and length of haplotype should be equal number of mutations sites.')
        elif index is not None:
            if hap:
                raise TypeError('Incorrect type of haplotype. Type should be int or str or None.')
            else:
                raise TypeError('Incorrect type of ' + smth + '. Type should be int or None.')

    def check_list(self, data, smth, length):
        if isinstance(data, list):
            if len(data) != length:
                raise ValueError('Incorrect length of ' + smth + '. Length should be equal ' + str(length) + '.')
        else:
            raise TypeError('Incorrect type of ' + smth + '. Type should be list.')

    def check_amount_sus(self, amount, source_type, target_type, population):
        if self.susceptible[population, source_type] - amount < 0:
            raise ValueError('Number of susceptible minus amount should be more or equal 0.')
        if self.susceptible[population, target_type] + amount > self.sizes[population]:
            raise ValueError('Number of susceptible plus amount should be equal or less population size.')

    def check_amount_inf(self, amount, source_type, target_haplotype, population):
        if self.susceptible[population, source_type] - amount < 0:
            raise ValueError('Number of susceptible minus amount should be more or equal 0.')
        if self.infectious[population, target_haplotype] + amount > self.sizes[population]:
            raise ValueError('Number of infectious plus amount should be equal or less population size.')

    def check_mig_rate(self):
        for pn1 in range(self.popNum):
            summa = 0
            self.migrationRates[pn1, pn1] = 1.0
            for pn2 in range(self.popNum):
                if pn1 != pn2:
                    summa += self.migrationRates[pn1, pn2]
                    self.migrationRates[pn1, pn1] -= self.migrationRates[pn1, pn2]
            if summa > 1:
                raise ValueError('Incorrect the sum of migration probabilities. The sum of migration probabilities from each population should be equal or less 1.')
        for pn in range(self.popNum):
            if self.migrationRates[pn, pn] <= 1e-15:
                raise ValueError('Incorrect value of migration probability. Value of migration probability from source population to target population should be more 0.')

    # ------------------------------------------------------------------ read-only properties (pyx:1269-1295)
    @property
    def seed(self):
        return self.user_seed

    @property
    def sampling_probability(self):
        return self._sampling_probability

    @property
    def memory_optimization(self):
        return self._memory_optimization

    @property
    def number_of_sites(self):
        return self.sites

    @property
    def haplotypes_number(self):
        return self.hapNum

    @property
    def populations_number(self):
        return self.popNum

    @property
    def number_of_susceptible_groups(self):
        return self.susNum

    # ------------------------------------------------------------------ setters (pyx:1380-1702)
    @property
    def initial_haplotype(self):
        return self.maxHapNum

    def set_initial_haplotype(self, amount):
        if self._memory_optimization == False:  # noqa: E712
            raise ValueError('Incorrect value of memory optimization. Value should be equal \'True\' for work this function.')
        self.check_amount(amount, 'amount of initial haplotype')
        self.maxHapNum = self.hapNum if amount >= self.hapNum else amount

    @property
    def step_haplotype(self):
        return self.addMemoryNum

    def set_step_haplotype(self, amount):
        if self._memory_optimization == False:  # noqa: E712
            raise ValueError('Incorrect value of memory optimization. Value should be equal \'True\' for work this function.')
        self.check_amount(amount, 'amount of step haplotype')
        self.addMemoryNum = amount

    @property
    def genome_length(self):
        return self._genome_length

    def set_genome_length(self, genome_length):
        self.check_amount(genome_length, 'genome length')
        if self.sites > genome_length:
            raise ValueError('Incorrect value of number of sites or genome length. Genome length should be more or equal number of sites.')
        self._genome_length = genome_length
        for s in range(self.sites):
            self.sitesPosition[s] = int(s * self._genome_length / (self.sites - 1))

    @property
    def coinfection_parameters(self):
        return self.recombination

    def set_coinfection_parameters(self, recombination):
        self.check_value(recombination, 'recombination probability', edge=1)
        self.recombination = recombination

    @property
    def transmission_rate(self):
        return self.bRate

    def set_transmission_rate(self, rate, haplotype):
        self.check_value(rate, 'transmission rate')
        self.check_indexes(haplotype, self.hapNum, 'haplotype', True)
        for hn in self.calculate_indexes(haplotype, self.hapNum):
            self.bRate[hn] = rate

    @property
    def recovery_rate(self):
        return self.dRate

    def set_recovery_rate(self, rate, haplotype):
        self.check_value(rate, 'recovery rate')
        self.check_indexes(haplotype, self.hapNum, 'haplotype', True)
        for hn in self.calculate_indexes(haplotype, self.hapNum):
            self.dRate[hn] = rate

    @property
    def sampling_rate(self):
        return self.sRate

    def set_sampling_rate(self, rate, haplotype):
        self.check_indexes(haplotype, self.hapNum, 'haplotype', True)
        haplotypes = self.calculate_indexes(haplotype, self.hapNum)
        if self._sampling_probability == True:  # noqa: E712
            self.check_value(rate, 'sampling probability', edge=1)
            for hn in haplotypes:
                deathRate = self.dRate[hn] + self.sRate[hn]
                self.dRate[hn] = (1 - rate) * deathRate
                self.sRate[hn] = rate * deathRate
        elif self._sampling_probability == False:  # noqa: E712
            self.check_value(rate, 'sampling rate')
            for hn in haplotypes:
                self.sRate[hn] = rate

    @property
    def mutation_rate(self):
        return self.mRate

    def set_mutation_rate(self, rate, haplotype, mutation):
        self.check_value(rate, 'mutation rate')
        self.check_indexes(haplotype, self.hapNum, 'haplotype', True)
        self.check_indexes(mutation, self.sites, 'mutation site')
        haplotypes = self.calculate_indexes(haplotype, self.hapNum)
        sites = self.calculate_indexes(mutation, self.sites)
        for hn in haplotypes:
            for s in sites:
                self.mRate[hn, s] = rate

    @property
    def mutation_probabilities(self):
        return self.hapMutType

    def set_mutation_probabilities(self, probabilities, haplotype, mutation):
        self.check_list(probabilities, 'probabilities list', 4)
        for i in range(4):
            self.check_value(probabilities[i], 'mutation probabilities')
        self.check_indexes(haplotype, self.hapNum, 'haplotype', True)
        self.check_indexes(mutation, self.sites, 'mutation site')
        haplotypes = self.calculate_indexes(haplotype, self.hapNum)
        sites = self.calculate_indexes(mutation, self.sites)
        for hn in haplotypes:
            for s in sites:
                probabilities_allele = list(probabilities)
                del probabilities_allele[self.calculate_allele(hn, s)]
                if sum(probabilities_allele) == 0:
                    raise ValueError('Incorrect probabilities list. The sum of three elements without mutation allele should be more 0.')
                self.hapMutType[hn, s, 0] = probabilities_allele[0]
                self.hapMutType[hn, s, 1] = probabilities_allele[1]
                self.hapMutType[hn, s, 2] = probabilities_allele[2]

    @property
    def mutation_position(self):
        return self.sitesPosition

    def set_mutation_position(self, mutation, position):
        self.check_index(mutation, self.sites, 'number of site', none=False)
        self.check_index(position, self._genome_length, 'mutation position', none=False)
        for s in range(self.sites):
            if self.sitesPosition[s] == position and s != mutation:
                raise IndexError('Incorrect value of position. Two mutations can\'t have the same position.')
        self.sitesPosition[mutation] = position

    @property
    def susceptibility_type(self):
        return self.suscType

    def set_susceptibility_type(self, susceptibility_type, haplotype):
        if isinstance(susceptibility_type, int) == False:  # noqa: E712
            raise TypeError('Incorrect type of susceptibility type. Type should be int.')
        elif susceptibility_type < 0 or susceptibility_type >= self.susNum:
            raise IndexError('There are no such susceptibility type!')
        self.check_indexes(haplotype, self.hapNum, 'haplotype', True)
        for hn in self.calculate_indexes(haplotype, self.hapNum):
            self.suscType[hn] = susceptibility_type

    def set_susceptibility(self, rate, haplotype, susceptibility_type):
        self.check_value(rate, 'susceptibility rate')
        self.check_indexes(haplotype, self.hapNum, 'haplotype', True)
        self.check_indexes(susceptibility_type, self.susNum, 'susceptibility type')
        haplotypes = self.calculate_indexes(haplotype, self.hapNum)
        sus_types = self.calculate_indexes(susceptibility_type, self.susNum)
        for hn in haplotypes:
            for sn in sus_types:
                self.susceptibility[hn, sn] = rate

    @property
    def immunity_transition(self):
        return self.suscepTransition

    def set_immunity_transition(self, rate, source, target):
        self.check_value(rate, 'immunity transition rate')
        self.check_indexes(source, self.susNum, 'susceptibility type')
        self.check_indexes(target, self.susNum, 'susceptibility type')
        for sn1 in self.calculate_indexes(source, self.susNum):
            for sn2 in self.calculate_indexes(target, self.susNum):
                if sn1 != sn2:
                    self.suscepTransition[sn1, sn2] = rate

    @property
    def population_size(self):
        return self.sizes

    def set_population_size(self, amount, population):
        if self.first_simulation == True:  # noqa: E712
            raise ValueError('Changing population size is available only before first simulation!')
        self.check_amount(amount, 'population size')
        self.check_index(population, self.popNum, 'population')
        for pn in self.calculate_index(population, self.popNum):
            self.sizes[pn] = amount
            self.susceptible[pn, 0] = amount
            for sn in range(1, self.susNum):
                self.susceptible[pn, sn] = 0

    def set_susceptible(self, amount, source_type, target_type, population):
        """Working version of pyx:1593-1608.  The reference's own method always raises TypeError
        (it calls ``check_amount(amount)`` without the required argument, pyx:1596); this engine
        supplies the argument, everything else is as written there."""
        if self.first_simulation:
            raise ValueError('This function is available only before first simulation!')
        self.check_amount(amount, 'amount')
        self.check_index(source_type, self.susNum, 'susceptibility type')
        self.check_index(target_type, self.susNum, 'susceptibility type')
        if source_type == target_type:
            raise ValueError('Source and target susceptibility type shouldn\'t be equal!')
        self.check_indexes(population, self.popNum, 'population')
        for pn in self.calculate_indexes(population, self.popNum):
            self.check_amount_sus(amount, source_type, target_type, pn)
            self.susceptible[pn, source_type] -= amount
            self.susceptible[pn, target_type] += amount

    def set_infectious(self, amount, source_type, target_haplotype, population):
        """Working version of pyx:1614-1627 (same upstream defect as ``set_susceptible``, pyx:1617)."""
        if self.first_simulation:
            raise ValueError('This function is available only before first simulation!')
        self.check_amount(amount, 'amount')
        self.check_index(source_type, self.susNum, 'susceptibility type')
        self.check_index(target_haplotype, self.hapNum, 'haplotype')
        self.check_indexes(population, self.popNum, 'population')
        for pn in self.calculate_indexes(population, self.popNum):
            self.check_amount_inf(amount, source_type, target_haplotype, pn)
            self.susceptible[pn, source_type] -= amount
            self.infectious[pn, target_haplotype] += amount

    @property
    def contact_density(self):
        return self.contactDensity

    def set_contact_density(self, value, population):
        self.check_value(value, 'contact density')
        self.check_indexes(population, self.popNum, 'population')
        for pn in self.calculate_indexes(population, self.popNum):
            self.contactDensity[pn] = value
            self.contactDensityBeforeLockdown[pn] = value

    @property
    def npi(self):
        return [self.contactDensityAfterLockdown, self.startLD, self.endLD]

    def set_npi(self, parameters, population):
        self.check_list(parameters, 'npi parameters', 3)
        self.check_value(parameters[0], 'first npi parameter')
        self.check_value(parameters[1], 'second npi parameter', edge=1)
        self.check_value(parameters[2], 'third npi parameter', edge=1)
        self.check_indexes(population, self.popNum, 'population')
        for pn in self.calculate_indexes(population, self.popNum):
            self.contactDensityAfterLockdown[pn] = parameters[0]
            self.startLD[pn] = parameters[1]
            self.endLD[pn] = parameters[2]

    @property
    def sampling_multiplier(self):
        return self.samplingMultiplier

    def set_sampling_multiplier(self, multiplier, population):
        self.check_value(multiplier, 'sampling multiplier')
        self.check_indexes(population, self.popNum, 'population')
        for pn in self.calculate_indexes(population, self.popNum):
            self.samplingMultiplier[pn] = multiplier

    @property
    def migration_probability(self):
        return self.migrationRates

    def set_migration_probability(self, probability, source, target):
        self.check_value(probability, 'migration probability', edge=1)
        self.check_indexes(source, self.popNum, 'population')
        self.check_indexes(target, self.popNum, 'population')
        for pn1 in self.calculate_indexes(source, self.popNum):
            for pn2 in self.calculate_indexes(target, self.popNum):
                if pn1 != pn2:
                    self.migrationRates[pn1, pn2] = probability
        self.check_mig_rate()

    def set_total_migration_probability(self, total_probability):
        self.check_value(total_probability, 'total migration probability', edge=1)
        source_rate = 1.0 - total_probability
        target_rate = total_probability / (self.popNum - 1)
        self.migrationRates[:, :] = target_rate
        np.fill_diagonal(self.migrationRates, source_rate)
        self.check_mig_rate()

    # ------------------------------------------------------------------ hot-path entry points
    def _check_supported(self):
        if self.recombination != 0 and self.sites < 2:
            # upstream allocates the scratch vector of the recombination branch only for sites > 1 (pyx:98-102) and
            # crashes in Birth otherwise
            raise ValueError('Incorrect value of recombination probability. Recombination needs at least two sites.')

    def _refresh_haplotype_table(self):
        """``memory_optimization=True`` (pyx:105-125, 264-274, 355-377, 651-660).  The engine's state is sparse in the
        haplotype dimension whatever the flag says (ordered occupancy lists), and the reference's table
        keeps program numbers in haplotype order (sorted insert, pyx:365-375), i.e. the scan order of the plain
        layout: the trajectory is that of ``memory_optimization=False``.  What remains of the option is its
        bookkeeping, rebuilt here after every simulate call: ``numToHap`` = the haplotypes seen so far in ascending
        order, ``hapToNum`` its inverse, ``maxHapNum`` grown in ``addMemoryNum`` steps like ``AddMemory``."""
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
        """stdout contract of pyx:456-471."""
        self._compute_actual_sizes()
        check = False
        list_pop = []
        print('Actual sizes: ', end='')
        for pn in range(self.popNum):
            print(self.actualSizes[pn], end=' ')
            if abs(self.actualSizes[pn] / self.sizes[pn] - 1) >= 0.1:
                check = True
                list_pop.append(str(pn))
        print()
        if check:
            print('\033[41m{}\033[0m'.format('WARNING!'), 'Actual population size in deme: ', end='')
            print(", ".join(list_pop))
            print("\tis more than 10% different from the population size. The migration probabilities might be unrealistically high.")
            print("\tWe recommend to check your model with print_populations() method before proceding to simulation.")
            print("\tCheck the documentation file:https://vg-sim.readthedocs.io/en/latest/Migration.html for more details.")

    def _print_termination(self, sample_size, time):
        """pyx:420-429 / pyx:2337-2346."""
        if self.totalRate == 0.0 or self.globalInfectious == 0:
            print('Simulation finished because no infections individuals remain!')
        if self.events.ptr >= self.events.size:
            print("Achieved maximal number of iterations.")
        if self.sCounter > sample_size and sample_size != -1:
            print("Achieved sample size.")
        if self.currentTime > time and time != -1:
            print("Achieved internal time limit.")
        if self.sCounter <= 1:
            print('\033[41m{}\033[0m'.format('WARNING!'), 'Simulated less 2 samples, so genealogy will not work!')

    def _get_engine(self):
        if self._engine is None:
            from ._capi import HipEngine  # raises loudly if libvgx.so or the GPU is missing
            self._engine = HipEngine(self.sites, self.hapNum, self.popNum, self.susNum, n_replicates=1)
        return self._engine

    def SimulatePopulation(self, iterations, sample_size, time, attempts, mode='exact', kernel='auto'):
        """pyx:396-429: direct Gillespie on the GPU.  ``mode``: 'exact' = the reference's floating-point summation
        order (bit-exact log); 'fast' = order-free sums (same random stream and event semantics; include/vgx.h
        vgx_run_opts.mode).  ``kernel``: 'auto', 'wave' (one replicate per wavefront) or 'lane' (one replicate per
        lane, small models; vgx_run_opts.kernel)."""
        self._check_supported()
        if mode not in ('exact', 'fast'):
            raise ValueError("mode must be 'exact' or 'fast'")
        if kernel not in ('auto', 'wave', 'lane'):
            raise ValueError("kernel must be 'auto', 'wave' or 'lane'")
        self.events.CreateEvents(iterations)
        self.CheckSizes()
        time = float(np.float32(time))  # `float time` in the reference signature
        opts = None
        if mode == 'fast' or kernel != 'auto':
            from . import _capi
            opts = _capi.VgxRunOpts()
            opts.record_events = 1
            opts.mode = 1 if mode == 'fast' else 0
            opts.kernel = {'auto': 0, 'wave': 1, 'lane': 2}[kernel]
        eng = self._get_engine()
        eng.simulate_direct(self, iterations, sample_size, time, attempts, opts)
        c = eng.last_counters
        self._rng_position = (int(c.reserved[1]), 2 * int(c.reserved[2])) if c.reserved[1] >= 0 else None
        self._rng_raw = None
        self._refresh_haplotype_table()
        self._print_termination(sample_size, time)

    def SimulatePopulation_tau(self, iterations, sample_size, time, attempts):
        """pyx:2293-2346: Poisson tau-leaping on the GPU."""
        self._check_supported()
        self.events.CreateEvents(iterations)   # via PrepareParameters (pyx:2298 -> pyx:434)
        self.events.CreateEvents(iterations)   # pyx:2306
        self.CheckSizes()
        time = float(np.float32(time))
        self._get_engine().simulate_tau(self, iterations, sample_size, time, attempts)
        self._rng_position, self._rng_raw = None, None
        self._refresh_haplotype_table()
        self._print_termination(sample_size, time)

    # ------------------------------------------------------------------ reporting (pyx:2048-2068, 2284, 2607-2613, 1849-1851)
    def Stats(self, time_simulation):
        print("Number of samples:", self.sCounter)
        print("Total number of iterations:", self.events.ptr)
        print('Success number:', self.good_attempt)
        print("Epidemic time:", self.currentTime)
        print('Simulation time:', time_simulation)
        print('Number of infections:', self.bCounter)
        print('Number of recoveries:', self.dCounter)
        if self.sites >= 1:
            print('Number of mutations:', self.mCounter)
        if self.popNum >= 2:
            print('Number of accepted migrations:', self.migPlus)
            print('Number of rejected migrations:', self.migNonPlus)
        if np.any(self.suscepTransition.sum(axis=1) != 0.0):
            print('Number of immunity transitions:', self.iCounter)
        print('----------------------------------')

    def get_proportion(self):
        return self.migNonPlus / (self.events.ptr - 1)

    def PrintCounters(self):
        print("Birth counter(mutable): ", self.bCounter)
        print("Death counter(mutable): ", self.dCounter)
        print("Sampling counter(mutable): ", self.sCounter)
        print("Mutation counter(mutable): ", self.mCounter)
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
        time, pop, hap = [], [], []
        for i in range(self.events.ptr):
            if self.events.types[i] == SAMPLING:
                time.append(self.events.times[i])
                pop.append(self.events.populations[i])
                hap.append(self.events.haplotypes[i])
        return time, pop, hap

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
