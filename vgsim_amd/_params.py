"""Argument checking and parameter assignment of the host model (the non-hot half of the drop-in boundary).

The reference's ``BirthDeathModel`` has one hand-written method per setter (``src/_BirthDeath.pyx:1381-1702``) over a
family of ``check_*`` helpers (pyx:1298-1377); what users and the reference's ``tests/test_interface.py`` see of them is
the method names, the argument order, the exception TYPES and MESSAGES, and the resulting arrays.  Here that surface is
produced by three pieces:

* ``need_count`` / ``need_number`` / ``need_list`` — scalar checks, one message template each;
* ``ParameterTable._select`` — turns "an index, a list of them, a haplotype pattern such as ``'A*G'`` or ``None``" into a
  sorted ``numpy`` index vector (patterns are expanded arithmetically in base 4, not by string substitution);
* setter factories (``_per_haplotype`` ...) that validate in the pinned order and then assign with one vectorised
  ``numpy`` statement instead of per-element loops.
"""
import numpy as np

LETTERS = "ATCG"       # base-4 digit of a nucleotide, most significant site first (pyx:1243-1261)
_BAD_PATTERN = ('Incorrect haplotype. Haplotype should contain only "A", "T", "C", "G", "*" and length of haplotype '
                'should be equal number of mutations sites.')


# ---------------------------------------------------------------------------------------------- scalar checks
def need_count(value, what, positive=True):
    """An ``int`` that is > 0 (``positive``) or >= 0."""
    if not isinstance(value, int):
        raise TypeError('Incorrect type of %s. Type should be int.' % what)
    if value < (1 if positive else 0):
        raise ValueError('Incorrect value of %s. Value should be more %s0.' % (what, '' if positive else 'or equal '))


def need_number(value, what, upper=None, optional=False):
    """A non-negative ``int``/``float``, at most ``upper`` when given; ``None`` passes when ``optional``."""
    if not isinstance(value, (int, float)):
        if optional and value is None:
            return
        raise TypeError('Incorrect type of %s. Type should be int or float%s.' % (what, ' or None' if optional else ''))
    if upper is None:
        if value < 0:
            raise ValueError('Incorrect value of %s. Value should be more or equal 0.' % what)
    elif value < 0 or value > upper:
        raise ValueError('Incorrect value of %s. Value should be more or equal 0 and equal or less %s.' % (what, upper))


def need_list(value, what, length):
    if not isinstance(value, list):
        raise TypeError('Incorrect type of %s. Type should be list.' % what)
    if len(value) != length:
        raise ValueError('Incorrect length of %s. Length should be equal %d.' % (what, length))


# ---------------------------------------------------------------------------------------------- setter factories
def _per_haplotype(what, *arrays):
    """``set_x(rate, haplotype)``: one rate for the selected haplotypes."""
    def setter(self, rate, haplotype):
        need_number(rate, what)
        rows = self._select(haplotype, self.hapNum, 'haplotype', pattern=True)
        for name in arrays:
            getattr(self, name)[rows] = rate
    return setter


def _per_population(what, *arrays, upper=None):
    """``set_x(value, population)``: one value for the selected populations."""
    def setter(self, value, population):
        need_number(value, what, upper=upper)
        rows = self._select(population, self.popNum, 'population')
        for name in arrays:
            getattr(self, name)[rows] = value
    return setter


def _view(name):
    return property(lambda self: getattr(self, name))


class ParameterTable:
    """Mixin of ``BirthDeathModel``: everything a user can set before ``simulate`` (pyx:1381-1702)."""

    # ------------------------------------------------------------------ selectors
    def _pattern_rows(self, pattern):
        """Haplotype numbers matched by a pattern: sites left to right are base-4 digits, most significant first;
        ``*`` stands for all four letters, anything else that is not T/C/G counts as A like upstream (pyx:1251-1261)."""
        rows = np.zeros(1, dtype=np.int64)
        tail = pattern[len(pattern) - self.sites:] if self.sites else ''
        for pos, ch in enumerate(tail):
            weight = 4 ** (self.sites - 1 - pos)
            if ch == '*':
                rows = (rows[:, None] + weight * np.arange(4, dtype=np.int64)[None, :]).ravel()
            else:
                rows = rows + weight * max(LETTERS.find(ch), 0)
        return rows

    def _check_one(self, item, size, what, pattern, required):
        if item is None:
            if required:
                raise TypeError('Incorrect type of %s. Type should be int.' % what)
        elif isinstance(item, int):
            if not 0 <= item < size:
                raise IndexError('There are no such %s!' % what)
        elif pattern and isinstance(item, str):
            if sum(item.count(ch) for ch in LETTERS + '*') != self.sites:
                raise ValueError(_BAD_PATTERN)
        elif pattern:
            raise TypeError('Incorrect type of haplotype. Type should be int or str or None.')
        else:
            raise TypeError('Incorrect type of %s. Type should be int or None.' % what)

    def _rows_of(self, item, size, pattern):
        if isinstance(item, int):
            return np.array([item], dtype=np.int64)
        if pattern and isinstance(item, str):
            return self._pattern_rows(item)
        return np.arange(size, dtype=np.int64)

    def _select(self, selector, size, what, pattern=False, required=False, listable=True):
        """Validated, sorted, duplicate-free index vector for an int / pattern / None selector or a list of them."""
        items = selector if (listable and isinstance(selector, list)) else [selector]
        for item in items:
            self._check_one(item, size, what, pattern, required)
        if not items:
            return np.zeros(0, dtype=np.int64)
        return np.unique(np.concatenate([self._rows_of(item, size, pattern) for item in items]))

    # names the reference exposes for the same jobs (used by the facade, the printers and the readers)
    def calculate_indexes(self, indexes_list, edge):
        pattern = edge == self.hapNum
        items = indexes_list if isinstance(indexes_list, list) else [indexes_list]
        return set(int(i) for item in items for i in self._rows_of(item, edge, pattern or isinstance(item, str)))

    def calculate_index(self, index, edge):
        return [int(i) for i in self._rows_of(index, edge, isinstance(index, str))]

    def calculate_string_from_haplotype(self, hapNum):
        return ''.join(LETTERS[(hapNum // 4 ** (self.sites - 1 - pos)) % 4] for pos in range(self.sites))

    def calculate_haplotype_from_string(self, string):
        return int(self._pattern_rows(string.replace('*', 'A'))[0])

    def calculate_allele(self, haplotype, site):
        return (haplotype // 4 ** (self.sites - 1 - site)) % 4

    def check_amount(self, amount, smth, zero=True):
        need_count(amount, smth, positive=zero)

    def check_value(self, value, smth, edge=None, none=False):
        need_number(value, smth, upper=edge, optional=none)

    def check_indexes(self, index, edge, smth, hap=False, none=True):
        self._select(index, edge, smth, pattern=hap, required=not none)

    def check_index(self, index, edge, smth, hap=False, none=True):
        self._select(index, edge, smth, pattern=hap, required=not none, listable=False)

    def check_list(self, data, smth, length):
        need_list(data, smth, length)

    def check_mig_rate(self):
        """Diagonal of the migration matrix = 1 - (off-diagonal row sum), subtracted column by column in index order
        (the rounding sequence of pyx:1365-1372); a row may not give away more than everything."""
        m, P = self.migrationRates, self.popNum
        stay, leave = np.ones(P), np.zeros(P)
        rows = np.arange(P)
        for col in range(P):
            step = np.where(rows == col, 0.0, m[:, col])
            leave = leave + step
            stay = stay - step
        m[rows, rows] = stay
        if (leave > 1).any():
            raise ValueError('Incorrect the sum of migration probabilities. The sum of migration probabilities from each '
                             'population should be equal or less 1.')
        if (stay <= 1e-15).any():
            raise ValueError('Incorrect value of migration probability. Value of migration probability from source '
                             'population to target population should be more 0.')

    # ------------------------------------------------------------------ read-only views (pyx:1269-1295)
    seed = _view('user_seed')
    sampling_probability = _view('_sampling_probability')
    memory_optimization = _view('_memory_optimization')
    number_of_sites = _view('sites')
    haplotypes_number = _view('hapNum')
    populations_number = _view('popNum')
    number_of_susceptible_groups = _view('susNum')
    initial_haplotype = _view('maxHapNum')
    step_haplotype = _view('addMemoryNum')
    genome_length = _view('_genome_length')
    coinfection_parameters = _view('recombination')
    transmission_rate = _view('bRate')
    recovery_rate = _view('dRate')
    sampling_rate = _view('sRate')
    mutation_rate = _view('mRate')
    mutation_probabilities = _view('hapMutType')
    mutation_position = _view('sitesPosition')
    susceptibility_type = _view('suscType')
    immunity_transition = _view('suscepTransition')
    population_size = _view('sizes')
    contact_density = _view('contactDensity')
    sampling_multiplier = _view('samplingMultiplier')
    migration_probability = _view('migrationRates')
    npi = property(lambda self: [self.contactDensityAfterLockdown, self.startLD, self.endLD])

    # ------------------------------------------------------------------ haplotype table / genome (pyx:1381-1426)
    def _needs_memory_optimization(self):
        if not self._memory_optimization:
            raise ValueError("Incorrect value of memory optimization. Value should be equal 'True' for work this function.")

    def set_initial_haplotype(self, amount):
        self._needs_memory_optimization()
        need_count(amount, 'amount of initial haplotype')
        self.maxHapNum = min(amount, self.hapNum)

    def set_step_haplotype(self, amount):
        self._needs_memory_optimization()
        need_count(amount, 'amount of step haplotype')
        self.addMemoryNum = amount

    def _spread_sites(self):
        """Default site positions: evenly over ``[0, genome_length]`` (pyx:98-101)."""
        for s in range(self.sites):
            self.sitesPosition[s] = int(s * self._genome_length / (self.sites - 1))

    def set_genome_length(self, genome_length):
        need_count(genome_length, 'genome length')
        if self.sites > genome_length:
            raise ValueError('Incorrect value of number of sites or genome length. Genome length should be more or equal '
                             'number of sites.')
        self._genome_length = genome_length
        self._spread_sites()

    def set_coinfection_parameters(self, recombination):
        need_number(recombination, 'recombination probability', upper=1)
        self.recombination = recombination

    # ------------------------------------------------------------------ per-haplotype rates (pyx:1431-1546)
    set_transmission_rate = _per_haplotype('transmission rate', 'bRate')
    set_recovery_rate = _per_haplotype('recovery rate', 'dRate')

    def set_sampling_rate(self, rate, haplotype):
        rows = self._select(haplotype, self.hapNum, 'haplotype', pattern=True)
        if self._sampling_probability:
            # `rate` is the sampled share of all removals: recovery and sampling split their common total (pyx:1459-1465)
            need_number(rate, 'sampling probability', upper=1)
            removal = self.dRate[rows] + self.sRate[rows]
            self.dRate[rows] = (1 - rate) * removal
            self.sRate[rows] = rate * removal
        else:
            need_number(rate, 'sampling rate')
            self.sRate[rows] = rate

    def set_mutation_rate(self, rate, haplotype, mutation):
        need_number(rate, 'mutation rate')
        rows = self._select(haplotype, self.hapNum, 'haplotype', pattern=True)
        cols = self._select(mutation, self.sites, 'mutation site')
        self.mRate[np.ix_(rows, cols)] = rate

    def set_mutation_probabilities(self, probabilities, haplotype, mutation):
        need_list(probabilities, 'probabilities list', 4)
        for weight in probabilities:
            need_number(weight, 'mutation probabilities')
        rows = self._select(haplotype, self.hapNum, 'haplotype', pattern=True)
        cols = self._select(mutation, self.sites, 'mutation site')
        if not len(rows) or not len(cols):
            return
        # a haplotype cannot mutate into the letter it already carries at the site: drop that entry of the four weights
        carried = (rows[:, None] // 4 ** (self.sites - 1 - cols)[None, :]) % 4
        others = [[w for letter, w in enumerate(probabilities) if letter != own] for own in range(4)]
        for own in np.unique(carried):
            if sum(others[own]) == 0:
                raise ValueError('Incorrect probabilities list. The sum of three elements without mutation allele should '
                                 'be more 0.')
        self.hapMutType[rows[:, None], cols[None, :], :] = np.array(others, dtype=float)[carried]

    def set_mutation_position(self, mutation, position):
        self._select(mutation, self.sites, 'number of site', required=True, listable=False)
        self._select(position, self._genome_length, 'mutation position', required=True, listable=False)
        taken = np.nonzero(self.sitesPosition == position)[0]
        if len(taken) and (len(taken) > 1 or taken[0] != mutation):
            raise IndexError("Incorrect value of position. Two mutations can't have the same position.")
        self.sitesPosition[mutation] = position

    def set_susceptibility_type(self, susceptibility_type, haplotype):
        if not isinstance(susceptibility_type, int):      # no "or None" here, unlike the index arguments (pyx:1532-1533)
            raise TypeError('Incorrect type of susceptibility type. Type should be int.')
        self._select(susceptibility_type, self.susNum, 'susceptibility type', listable=False)
        self.suscType[self._select(haplotype, self.hapNum, 'haplotype', pattern=True)] = susceptibility_type

    def set_susceptibility(self, rate, haplotype, susceptibility_type):
        need_number(rate, 'susceptibility rate')
        rows = self._select(haplotype, self.hapNum, 'haplotype', pattern=True)
        cols = self._select(susceptibility_type, self.susNum, 'susceptibility type')
        self.susceptibility[np.ix_(rows, cols)] = rate

    def set_immunity_transition(self, rate, source, target):
        need_number(rate, 'immunity transition rate')
        src = self._select(source, self.susNum, 'susceptibility type')
        dst = self._select(target, self.susNum, 'susceptibility type')
        block = self.suscepTransition[np.ix_(src, dst)]
        self.suscepTransition[np.ix_(src, dst)] = np.where(src[:, None] == dst[None, :], block, rate)   # no self-transition

    # ------------------------------------------------------------------ populations (pyx:1555-1702)
    def set_population_size(self, amount, population):
        if self.first_simulation:
            raise ValueError('Changing population size is available only before first simulation!')
        need_count(amount, 'population size')
        rows = self._select(population, self.popNum, 'population', listable=False)
        self.sizes[rows] = amount
        self.susceptible[rows] = 0          # everybody starts in susceptibility group 0
        self.susceptible[rows, 0] = amount

    def _move_hosts(self, amount, source_type, population, counts, column, full):
        """Shared body of set_susceptible / set_infectious: ``amount`` hosts leave susceptible group ``source_type`` of
        every selected population and enter ``counts[:, column]``."""
        rows = self._select(population, self.popNum, 'population')
        for pn in rows:   # bounds are checked population by population, each after the previous one's move
            if self.susceptible[pn, source_type] - amount < 0:
                raise ValueError('Number of susceptible minus amount should be more or equal 0.')
            if counts[pn, column] + amount > self.sizes[pn]:
                raise ValueError('Number of %s plus amount should be equal or less population size.' % full)
            self.susceptible[pn, source_type] -= amount
            counts[pn, column] += amount

    def set_susceptible(self, amount, source_type, target_type, population):
        """pyx:1593-1608 made to work.  The reference's own method always raises ``TypeError`` (it calls
        ``check_amount(amount)`` without the required argument, pyx:1596); the intended behaviour is implemented."""
        if self.first_simulation:
            raise ValueError('This function is available only before first simulation!')
        need_count(amount, 'amount')
        self._select(source_type, self.susNum, 'susceptibility type', listable=False)
        self._select(target_type, self.susNum, 'susceptibility type', listable=False)
        if source_type == target_type:
            raise ValueError("Source and target susceptibility type shouldn't be equal!")
        self._move_hosts(amount, source_type, population, self.susceptible, target_type, 'susceptible')

    def set_infectious(self, amount, source_type, target_haplotype, population):
        """pyx:1614-1627 made to work (same upstream defect as ``set_susceptible``, pyx:1617)."""
        if self.first_simulation:
            raise ValueError('This function is available only before first simulation!')
        need_count(amount, 'amount')
        self._select(source_type, self.susNum, 'susceptibility type', listable=False)
        self._select(target_haplotype, self.hapNum, 'haplotype', listable=False)
        self._move_hosts(amount, source_type, population, self.infectious, target_haplotype, 'infectious')

    set_contact_density = _per_population('contact density', 'contactDensity', 'contactDensityBeforeLockdown')
    set_sampling_multiplier = _per_population('sampling multiplier', 'samplingMultiplier')

    def set_npi(self, parameters, population):
        need_list(parameters, 'npi parameters', 3)
        need_number(parameters[0], 'first npi parameter')
        need_number(parameters[1], 'second npi parameter', upper=1)
        need_number(parameters[2], 'third npi parameter', upper=1)
        rows = self._select(population, self.popNum, 'population')
        for name, value in zip(('contactDensityAfterLockdown', 'startLD', 'endLD'), parameters):
            getattr(self, name)[rows] = value

    def set_migration_probability(self, probability, source, target):
        need_number(probability, 'migration probability', upper=1)
        src = self._select(source, self.popNum, 'population')
        dst = self._select(target, self.popNum, 'population')
        block = self.migrationRates[np.ix_(src, dst)]
        self.migrationRates[np.ix_(src, dst)] = np.where(src[:, None] == dst[None, :], block, probability)
        self.check_mig_rate()

    def set_total_migration_probability(self, total_probability):
        need_number(total_probability, 'total migration probability', upper=1)
        self.migrationRates[:] = total_probability / (self.popNum - 1)    # spread evenly over the other populations
        np.fill_diagonal(self.migrationRates, 1.0 - total_probability)
        self.check_mig_rate()
