"""Host-side mirror of the reference's engine object: constructor validation, setters (values written, error
types and messages) and event-log bookkeeping.  The (argument, expected) tuples restate the behaviours the
reference pins in its tests/test_interface.py and events.pxi; they are checked on this repository's own classes."""
import contextlib
import io

import numpy as np
import pytest

from vgsim_amd import Simulator
from vgsim_amd._model import Events


def make(**kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return Simulator(**kw)


@pytest.mark.parametrize("kw,attr,expect", [
    (dict(number_of_sites=2), "number_of_sites", 2), (dict(number_of_sites=0), "number_of_sites", 0),
    (dict(number_of_sites=3), "haplotypes_number", 64), (dict(populations_number=2), "populations_number", 2),
    (dict(number_of_susceptible_groups=2), "number_of_susceptible_groups", 2), (dict(seed=15), "seed", 15),
    (dict(seed=0), "seed", 0), (dict(sampling_probability=True), "sampling_probability", True),
    (dict(memory_optimization=True), "memory_optimization", True), (dict(genome_length=1000), "genome_length", 1000),
    (dict(recombination_probability=0.5), "coinfection_parameters", 0.5),
])
def test_constructor_values(kw, attr, expect):
    assert getattr(make(**kw), attr) == expect


@pytest.mark.parametrize("kw,err,text", [
    (dict(number_of_sites=None), TypeError, 'Incorrect type of number of sites. Type should be int.'),
    (dict(number_of_sites=-2), ValueError, 'Incorrect value of number of sites. Value should be more or equal 0.'),
    (dict(populations_number='str'), TypeError, 'Incorrect type of populations number. Type should be int.'),
    (dict(populations_number=0), ValueError, 'Incorrect value of populations number. Value should be more 0.'),
    (dict(number_of_susceptible_groups=0), ValueError, 'Incorrect value of number of susceptible groups. Value should be more 0.'),
    (dict(seed='str'), TypeError, 'Incorrect type of seed. Type should be int.'),
    (dict(seed=-15), ValueError, 'Incorrect value of seed. Value should be more or equal 0.'),
    (dict(sampling_probability=None), ValueError, 'Incorrect value of sampling probability. Value of sampling probability should be True or False.'),
    (dict(memory_optimization='str'), ValueError, 'Incorrect value of memory optimization. Value of memory optimization should be True or False.'),
    (dict(genome_length=None), TypeError, 'Incorrect type of genome length. Type should be int.'),
    (dict(number_of_sites=4, genome_length=2), ValueError, 'Genome length should be more or equal number of sites.'),
    (dict(recombination_probability=1.1), ValueError, 'Value should be more or equal 0 and equal or less 1.'),
    (dict(recombination_probability='str'), TypeError, 'Incorrect type of recombination probability. Type should be int or float.'),
])
def test_constructor_errors(kw, err, text):
    with pytest.raises(err, match=text):
        make(**kw)


def test_defaults():
    m = make(number_of_sites=1, populations_number=2, number_of_susceptible_groups=2, seed=1).simulation
    assert np.array_equal(m.bRate, [2.0] * 4) and np.array_equal(m.dRate, [1.0] * 4) and np.array_equal(m.sRate, [0.01] * 4)
    assert np.array_equal(m.mRate, np.full((4, 1), 0.01)) and np.array_equal(m.hapMutType, np.ones((4, 1, 3)))
    assert np.array_equal(m.susceptibility, [[1.0, 0.0]] * 4) and np.array_equal(m.suscType, [0] * 4)
    assert np.array_equal(m.sizes, [10 ** 6] * 2) and np.array_equal(m.susceptible, [[10 ** 6, 0]] * 2)
    assert np.array_equal(m.contactDensity, [1, 1]) and np.array_equal(m.contactDensityAfterLockdown, [0, 0])
    assert np.array_equal(m.startLD, [1, 1]) and np.array_equal(m.endLD, [1, 1]) and np.array_equal(m.samplingMultiplier, [1, 1])


@pytest.mark.parametrize("hap,expect", [(None, [7] * 16), (3, [2, 2, 2, 7] + [2] * 12), ("AT", [2, 7] + [2] * 14),
                                        ("T*", [2] * 4 + [7] * 4 + [2] * 8), ([0, "GG"], [7] + [2] * 14 + [7])])
def test_haplotype_patterns(hap, expect):
    s = make(number_of_sites=2)
    s.set_transmission_rate(7, haplotype=hap)
    assert np.array_equal(s.transmission_rate, expect)


@pytest.mark.parametrize("call,err,text", [
    (lambda s: s.set_transmission_rate(None), TypeError, 'Incorrect type of transmission rate. Type should be int or float.'),
    (lambda s: s.set_transmission_rate(-1), ValueError, 'Incorrect value of transmission rate. Value should be more or equal 0.'),
    (lambda s: s.set_transmission_rate(1, haplotype=16), IndexError, 'There are no such haplotype!'),
    (lambda s: s.set_transmission_rate(1, haplotype='AAA'), ValueError, 'Incorrect haplotype.'),
    (lambda s: s.set_transmission_rate(1, haplotype=1.5), TypeError, 'Incorrect type of haplotype. Type should be int or str or None.'),
    (lambda s: s.set_recovery_rate('x'), TypeError, 'Incorrect type of recovery rate.'),
    (lambda s: s.set_sampling_rate(-0.1), ValueError, 'Incorrect value of sampling rate.'),
    (lambda s: s.set_mutation_rate(1, mutation=2), IndexError, 'There are no such mutation site!'),
    (lambda s: s.set_mutation_probabilities([1, 2, 3]), ValueError, 'Incorrect length of probabilities list. Length should be equal 4.'),
    (lambda s: s.set_mutation_probabilities('x'), TypeError, 'Incorrect type of probabilities list. Type should be list.'),
    (lambda s: s.set_mutation_probabilities([1, 0, 0, 0], haplotype='AA', mutation=0), ValueError, 'The sum of three elements without mutation allele should be more 0.'),
    (lambda s: s.set_susceptibility_type(3), IndexError, 'There are no such susceptibility type!'),
    (lambda s: s.set_susceptibility_type(1.0), TypeError, 'Incorrect type of susceptibility type. Type should be int.'),
    (lambda s: s.set_susceptibility(-1), ValueError, 'Incorrect value of susceptibility rate.'),
    (lambda s: s.set_immunity_transition(1, source=3), IndexError, 'There are no such susceptibility type!'),
    (lambda s: s.set_population_size(0), ValueError, 'Incorrect value of population size. Value should be more 0.'),
    (lambda s: s.set_population_size(10, population=5), IndexError, 'There are no such population!'),
    (lambda s: s.set_contact_density(-1), ValueError, 'Incorrect value of contact density.'),
    (lambda s: s.set_npi([1, 2]), ValueError, 'Incorrect length of npi parameters. Length should be equal 3.'),
    (lambda s: s.set_npi([1, 2, 0.5]), ValueError, 'Incorrect value of second npi parameter.'),
    (lambda s: s.set_sampling_multiplier('a'), TypeError, 'Incorrect type of sampling multiplier.'),
    (lambda s: s.set_migration_probability(1.5), ValueError, 'Incorrect value of migration probability.'),
    (lambda s: s.set_migration_probability(0.6), ValueError, 'The sum of migration probabilities from each population should be equal or less 1.'),
    (lambda s: s.set_initial_haplotype(3), ValueError, "Value should be equal 'True' for work this function."),
])
def test_setter_errors(call, err, text):
    s = make(number_of_sites=2, populations_number=3, number_of_susceptible_groups=3)
    with pytest.raises(err, match=text):
        call(s)


def test_setters_write_the_reference_arrays():
    s = make(number_of_sites=2, populations_number=3, number_of_susceptible_groups=3, sampling_probability=True)
    m = s.simulation
    s.set_recovery_rate(2.0)
    s.set_sampling_rate(0.25, haplotype=1)          # sampling_probability: splits d+s (pyx:1459-1465)
    assert m.dRate[1] == pytest.approx(0.75 * 2.01) and m.sRate[1] == pytest.approx(0.25 * 2.01) and m.sRate[0] == 0.01
    s.set_mutation_rate(0.5, haplotype="A*", mutation=1)
    assert np.array_equal(np.nonzero(m.mRate[:, 1] == 0.5)[0], [0, 1, 2, 3]) and (m.mRate[:, 0] == 0.01).all()
    s.set_mutation_probabilities([1, 2, 3, 4], haplotype=0, mutation=0)     # allele A removed
    assert np.array_equal(m.hapMutType[0, 0], [2, 3, 4])
    s.set_mutation_probabilities([1, 2, 3, 4], haplotype="GT")               # site 0 allele G, site 1 allele T
    assert np.array_equal(m.hapMutType[13, 0], [1, 2, 3]) and np.array_equal(m.hapMutType[13, 1], [1, 3, 4])
    s.set_susceptibility(0.3, haplotype=2, susceptibility_type=1)
    assert m.susceptibility[2, 1] == 0.3
    s.set_immunity_transition(0.1)
    assert np.array_equal(m.suscepTransition, 0.1 * (1 - np.eye(3)))
    s.set_population_size(500, population=1)
    assert np.array_equal(m.sizes, [10 ** 6, 500, 10 ** 6]) and np.array_equal(m.susceptible[1], [500, 0, 0])
    s.set_contact_density(0.5, population=2)
    assert m.contactDensity[2] == 0.5 and m.contactDensityBeforeLockdown[2] == 0.5
    s.set_npi([0.1, 0.2, 0.05], population=0)
    assert (m.contactDensityAfterLockdown[0], m.startLD[0], m.endLD[0]) == (0.1, 0.2, 0.05)
    s.set_migration_probability(0.1, source=0, target=1)
    assert m.migrationRates[0, 1] == 0.1 and m.migrationRates[0, 0] == pytest.approx(0.9) and m.migrationRates[1, 1] == 1.0
    s.set_total_migration_probability(0.2)
    assert np.allclose(m.migrationRates, np.full((3, 3), 0.1) + np.eye(3) * 0.7)
    s.set_infectious(5, source_type=0, target_haplotype=3, population=2)   # works here (upstream defect, pyx:1617)
    assert m.infectious[2, 3] == 5 and m.susceptible[2, 0] == 10 ** 6 - 5
    s.set_susceptible(7, source_type=0, target_type=2, population=0)
    assert np.array_equal(m.susceptible[0], [10 ** 6 - 7, 0, 7])


def test_events_capacity_rule():
    """events.pxi:52-68: size accumulates while ptr == 0, otherwise grows to ptr + iterations."""
    ev = Events()
    ev.CreateEvents(10)
    assert (ev.size, len(ev.times)) == (10, 10)
    ev.CreateEvents(5)
    assert ev.size == 15                      # ptr still 0: size += iterations
    ev.ptr = 12
    ev.types[:12] = 1
    ev.CreateEvents(2)
    assert ev.size == 15 and ev.types[:12].sum() == 12     # 2 + 12 - 15 <= 0: unchanged
    ev.CreateEvents(10)
    assert ev.size == 22 and len(ev.haplotypes) == 22 and ev.types[:12].sum() == 12
    assert ev.as_array().shape == (6, 22)


def test_memory_optimization_table_starts_empty():
    m = make(number_of_sites=3, memory_optimization=True).simulation   # pyx:105-121
    assert (m.maxHapNum, m.addMemoryNum, m.currentHapNum, len(m.numToHap), len(m.hapToNum)) == (4, 4, 0, 4, 64)
    m = make(number_of_sites=3).simulation
    assert m.currentHapNum == 64 and list(m.numToHap[:3]) == [0, 1, 2]


def test_unsupported_paths_raise():
    s = make(number_of_sites=1, recombination_probability=0.5)   # upstream has no scratch vector for sites < 2 (pyx:98-102)
    with pytest.raises(ValueError):
        s.simulate(10)
    with pytest.raises(SystemExit):   # "Less than two cases were sampled..." (pyx:762-765)
        make().genealogy()
