"""``Simulator``: the caller-facing facade, same constructor, setters and ``simulate`` signature as the
reference's ``src/_interface.py:9-883`` (delegation to ``self.simulation``; ``simulate`` = if:799-829).

Every method of the reference's facade is present with its name, argument order and defaults; ``simulate`` takes two
extra keywords (``mode``, ``kernel``).  The matplotlib helpers (if:639-797) live in ``_plots.py`` and import matplotlib
only when called.
"""
import sys
import time
from random import randrange

from ._model import BirthDeathModel
from ._plots import PlotMixin


class Simulator(PlotMixin):
    def __init__(self, number_of_sites=0, populations_number=1, number_of_susceptible_groups=1, seed=None,
                 sampling_probability=False, memory_optimization=False, genome_length=int(1e6),
                 recombination_probability=0.0):
        if seed == None:  # noqa: E711 (reference semantics, if:40)
            seed = int(randrange(sys.maxsize))
        print('User seed:', seed)
        self.fig = None
        self.simulation = BirthDeathModel(
            number_of_sites=number_of_sites, populations_number=populations_number,
            number_of_susceptible_groups=number_of_susceptible_groups, seed=seed,
            sampling_probability=sampling_probability, memory_optimization=memory_optimization,
            genome_length=genome_length, recombination_probability=recombination_probability)

    # read-only properties (if:130-156)
    seed = property(lambda self: self.simulation.seed)
    sampling_probability = property(lambda self: self.simulation.sampling_probability)
    memory_optimization = property(lambda self: self.simulation.memory_optimization)
    number_of_sites = property(lambda self: self.simulation.number_of_sites)
    haplotypes_number = property(lambda self: self.simulation.haplotypes_number)
    populations_number = property(lambda self: self.simulation.populations_number)
    number_of_susceptible_groups = property(lambda self: self.simulation.number_of_susceptible_groups)
    initial_haplotype = property(lambda self: self.simulation.initial_haplotype)
    step_haplotype = property(lambda self: self.simulation.step_haplotype)
    genome_length = property(lambda self: self.simulation.genome_length)
    coinfection_parameters = property(lambda self: self.simulation.coinfection_parameters)
    transmission_rate = property(lambda self: self.simulation.transmission_rate)
    recovery_rate = property(lambda self: self.simulation.recovery_rate)
    sampling_rate = property(lambda self: self.simulation.sampling_rate)
    mutation_rate = property(lambda self: self.simulation.mutation_rate)
    mutation_probabilities = property(lambda self: self.simulation.mutation_probabilities)
    mutation_position = property(lambda self: self.simulation.mutation_position)
    susceptibility_type = property(lambda self: self.simulation.susceptibility_type)
    susceptibility = property(lambda self: self.simulation.susceptibility)
    immunity_transition = property(lambda self: self.simulation.immunity_transition)
    population_size = property(lambda self: self.simulation.population_size)
    contact_density = property(lambda self: self.simulation.contact_density)
    npi = property(lambda self: self.simulation.npi)
    sampling_multiplier = property(lambda self: self.simulation.sampling_multiplier)
    migration_probability = property(lambda self: self.simulation.migration_probability)
    susceptible = property(lambda self: self.simulation.susceptible)
    infectious = property(lambda self: self.simulation.infectious)

    # setters (if:158-470): same names, argument order and defaults
    def set_initial_haplotype(self, amount):
        self.simulation.set_initial_haplotype(amount)

    def set_step_haplotype(self, amount):
        self.simulation.set_step_haplotype(amount)

    def set_genome_length(self, genome_length):
        self.simulation.set_genome_length(genome_length)

    def set_coinfection_parameters(self, recombination):
        self.simulation.set_coinfection_parameters(recombination)

    def set_transmission_rate(self, rate, haplotype=None):
        self.simulation.set_transmission_rate(rate, haplotype)

    def set_recovery_rate(self, rate, haplotype=None):
        self.simulation.set_recovery_rate(rate, haplotype)

    def set_sampling_rate(self, rate, haplotype=None):
        self.simulation.set_sampling_rate(rate, haplotype)

    def set_mutation_rate(self, rate, haplotype=None, mutation=None):
        self.simulation.set_mutation_rate(rate, haplotype, mutation)

    def set_mutation_probabilities(self, probabilities, haplotype=None, mutation=None):
        self.simulation.set_mutation_probabilities(probabilities, haplotype, mutation)

    def set_mutation_position(self, mutation, position):
        self.simulation.set_mutation_position(mutation, position)

    def set_susceptibility_type(self, susceptibility_type, haplotype=None):
        self.simulation.set_susceptibility_type(susceptibility_type, haplotype)

    def set_susceptibility(self, rate, haplotype=None, susceptibility_type=None):
        self.simulation.set_susceptibility(rate, haplotype, susceptibility_type)

    def set_immunity_transition(self, rate, source=None, target=None):
        self.simulation.set_immunity_transition(rate, source, target)

    def set_population_size(self, size, population=None):
        self.simulation.set_population_size(size, population)

    def set_contact_density(self, value, population=None):
        self.simulation.set_contact_density(value, population)

    def set_npi(self, parameters, population=None):
        self.simulation.set_npi(parameters, population)

    def set_sampling_multiplier(self, multiplier, population=None):
        self.simulation.set_sampling_multiplier(multiplier, population)

    def set_migration_probability(self, probability, source=None, target=None):
        self.simulation.set_migration_probability(probability, source, target)

    def set_total_migration_probability(self, total_probability):
        self.simulation.set_total_migration_probability(total_probability)

    def set_susceptible(self, amount, source_type, target_type, population=None):
        self.simulation.set_susceptible(amount, source_type, target_type, population)

    def set_infectious(self, amount, source_type, target_haplotype, population=None):
        self.simulation.set_infectious(amount, source_type, target_haplotype, population)

    # the drop-in boundary (if:799-829)
    def simulate(self, iterations=1000, sample_size=None, epidemic_time=-1, method='direct', attempts=200, mode='exact',
                 kernel='auto', record_multievents=True):
        if sample_size is None:
            sample_size = iterations
        if epidemic_time is None:
            epidemic_time = -1
        start_time = time.time()
        if method == 'direct':
            self.simulation.SimulatePopulation(iterations, sample_size, epidemic_time, attempts, mode=mode, kernel=kernel)
            self.simulation.Stats(time.time() - start_time)
        elif method == 'tau':
            self.simulation.SimulatePopulation_tau(iterations, sample_size, epidemic_time, attempts,
                                                   record_multievents=record_multievents)
            self.simulation.Stats(time.time() - start_time)
        else:
            print("Unknown method. Choose between 'direct' and 'tau'.")

    def ensemble(self, n_replicates, seeds=None, device=0):
        """Many independent seeded trajectories of THIS model on one GPU (``vgsim_amd.ensemble.Ensemble``): what the engine
        is built for.  One ``simulate()`` call on a small model is a single sequential event loop and runs at the speed of
        one wavefront (about 1e5 events/s, slower than the reference's CPU loop); replicates run concurrently, four per
        wavefront, at up to 1e9 events/s in aggregate.  Every replicate starts from this simulator's current state and
        parameters and is bit for bit the trajectory a ``Simulator`` with its seed would produce::

            ens = simulator.ensemble(4096)                        # seeds seed, seed+1, ...
            res = ens.simulate(100000, record_events=True)        # direct Gillespie for every replicate
            chain = ens.replicate_events(17)                      # (6, n) chain of one replicate, as export_chain_events
            state = ens.replicate_state(17)                       # compartments, counters, epidemic time
        """
        from .ensemble import Ensemble
        return Ensemble(self, n_replicates, seeds=seeds, device=device)

    def simulate_ensemble(self, n_replicates, iterations=1000, sample_size=None, epidemic_time=-1, method='direct', attempts=200,
                          seeds=None, **kw):
        """``simulate`` for ``n_replicates`` seeded copies of this model in one launch; returns ``(ensemble, result)``
        (``result.events`` = events.ptr of every replicate; chains and states through the ensemble's ``replicate_*``)."""
        ens = self.ensemble(n_replicates, seeds=seeds)
        if method == 'direct':
            res = ens.simulate(iterations, sample_size=sample_size, epidemic_time=epidemic_time, attempts=attempts,
                               record_events=kw.pop('record_events', True), **kw)
        elif method == 'tau':
            res = ens.simulate_tau(iterations, sample_size=sample_size, epidemic_time=epidemic_time, attempts=attempts, **kw)
        else:
            raise ValueError("Unknown method. Choose between 'direct' and 'tau'.")
        return ens, res

    def genealogy(self, seed=None):
        start_time = time.time()
        self.simulation.GetGenealogy(seed)
        print(f"Getting genealogy time: {time.time() - start_time}")

    def get_tree(self):
        return self.simulation.get_tree()

    def get_data_susceptible(self, population, susceptibility_type, step_num):  # if:599-617
        return self.simulation.get_data_susceptible(population, susceptibility_type, step_num)

    def get_data_infectious(self, population, haplotype, step_num):  # if:619-637
        return self.simulation.get_data_infectious(population, haplotype, step_num)

    def output_sample_data(self, output_print=False):  # if:536-553 (prints when output_print is False, as upstream)
        time, pop, hap = self.simulation.output_sample_data()
        if output_print:
            return time, pop, hap
        else:
            print(time)
            print(pop)
            print(hap)

    def export_newick(self, file_template=None, file_path=None):  # if:498-509
        from ._writers import write_newick
        pruferSeq, times, mut, populations = self.simulation.output_tree_mutations()
        write_newick(pruferSeq, times, populations, file_template, file_path)

    def export_mutations(self, file_template=None, file_path=None):  # if:511-522
        from ._writers import write_mutations
        pruferSeq, times, mut, populations = self.simulation.output_tree_mutations()
        write_mutations(mut, len(pruferSeq), file_template, file_path)

    def export_migrations(self, file_template=None, file_path=None):  # if:524-534
        self.simulation.export_migrations(file_template, file_path)

    def print_basic_parameters(self):  # if:49-53
        self.simulation.print_basic_parameters()

    def print_populations(self, population=True, susceptibles=True, infectious=True, migration=True):  # if:55-72
        self.simulation.print_populations(population=population, susceptibles=susceptibles, infectious=infectious,
                                          migration=migration)

    def print_immunity_model(self, immunity=True, transition=True):  # if:74-84
        self.simulation.print_immunity_model(immunity, transition)

    def print_all(self, basic_parameters=False, population=False, susceptible=False, infectious=False, migration=False,
                  immunity_model=False, immunity=False, transition=False):
        """if:92-126 (upstream passes an undefined name ``susceptibles`` to print_populations; ``susceptible`` is meant)."""
        if basic_parameters:
            self.simulation.print_basic_parameters()
        if population or susceptible or infectious or migration:
            self.simulation.print_populations(population=population, susceptibles=susceptible, infectious=infectious,
                                              migration=migration)
        if immunity or transition:
            self.simulation.print_immunity_model(immunity=immunity, transition=transition)

    def get_indexes_from_haplotype(self, haplotype):
        """if:129-130 (upstream calls an undefined ``create_list_for_cycles``): haplotype numbers matching a pattern
        such as ``'A*'``, an int, a list of those, or all for ``None``."""
        import numpy as np
        return np.array(sorted(self.simulation.calculate_indexes(haplotype, self.simulation.hapNum)))

    def print_mutations(self):
        self.simulation.print_mutations()

    def print_migrations(self):
        self.simulation.print_migrations()

    def export_chain_events(self, file_name="chain_events"):
        self.simulation.export_chain_events(file_name)

    def set_chain_events(self, file_name):
        self.simulation.set_chain_events(file_name)

    def export_settings(self, file_template="parameters"):  # if:569-574
        self.simulation.export_settings(file_template)

    def set_settings(self, file_template):  # if:478-482
        self.simulation.set_settings(file_template)

    def export_state(self, file_template="parameters"):
        """if:576-581 (upstream calls undefined ``output_chain_events``/``output_settings``): the event chain as
        ``<template>.npy`` plus the settings directory ``<template>/``."""
        self.export_chain_events(file_template)
        self.export_settings(file_template)

    def set_state(self, file_template):  # if:484-489
        self.set_chain_events(file_template)
        self.set_settings(file_template)

    def output_epidemiology_timelines(self, step=1000, output_file=False):  # if:555-567
        if output_file:
            self.simulation.output_epidemiology_timelines(step, output_file)
        else:
            return self.simulation.output_epidemiology_timelines(step, output_file)

    def export_ts(self):  # if:583-584
        return self.simulation.export_ts()

    def print_recomb(self, left, right):
        self.simulation.print_recomb(left, right)

    def print_chain(self):
        self.simulation.print_chain()

    def print_tree(self):
        self.simulation.print_tree()

    def debug(self):
        self.simulation.Debug()

    def print_propensities(self):
        self.simulation.PrintPropensities()

    def get_proportion(self):
        return self.simulation.get_proportion()

    def print_counters(self):
        self.simulation.PrintCounters()

    def citation(self):
        print("VGsim: scalable viral genealogy simulator for global pandemic")
        print("Vladimir Shchur, Vadim Spirin, Dmitry Sirotkin, EvgeniBurovski, Nicola De Maio, Russell Corbett-Detig")
        print("medRxiv 2021.04.21.21255891; doi: https://doi.org/10.1101/2021.04.21.21255891")
