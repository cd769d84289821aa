#!/usr/bin/env python3
"""Command line front end with the reference's flags (``VGsim_cmd.py``): settings files in, simulate on the GPU,
genealogy, optional Newick / mutation / migration / chain files out.  ``python -m vgsim_amd.cmd -h``."""
import argparse
import math
import sys
from random import randrange

from . import IO
from ._interface import Simulator
from ._report import apply_settings


def parser():
    p = argparse.ArgumentParser(description='VGsim-compatible epidemic simulation on MI355X.')
    p.add_argument('--iterations', '-it', type=int, default=1000, help='number of iterations (default is 1000)')
    p.add_argument('--sampleSize', '-s', type=int, default=None, help='number of sample (default is None)')
    p.add_argument('--time', '-t', type=float, default=None, help='time for stopping simulation (default is None)')
    p.add_argument('--seed', '-seed', type=float, default=None, help='random seed')
    p.add_argument('--rates', '-rt', default=None, help='rate: a file with rates for each haplotype')
    p.add_argument('--populationModel', '-pm', nargs=2, default=None,
                   help='population model: a file with population sizes etc, and a file with migration rate matrix')
    p.add_argument('--susceptibility', '-su', default=None, help='susceptibility file')
    p.add_argument('--suscepTransition', '-st', default=None, help='susceptibility transition file')
    p.add_argument('--sampling_probability', action="store_true",
                   help='the S column of the rates file is a sampling probability')
    p.add_argument('--method', default='direct', choices=('direct', 'tau'), help='simulation algorithm')
    p.add_argument("--createNewick", '-nwk', default=False, help="Create a newick file of tree *.nwk ")
    p.add_argument("--writeMutations", '-tsv', default=False, help="Create a mutation file *.tsv ")
    p.add_argument("--writeMigrations", default=False, help="Create a migration file *.tsv ")
    p.add_argument("--output_chain_events", default=False, help="Save the chain of events as *.npy")
    p.add_argument("-citation", '-c', action="store_true", help="Information for citation.")
    return p


def build_simulator(args):
    """Defaults and setter sequence of VGsim_cmd.py:79-142."""
    if args.rates is None:
        bRate, dRate, sRate, mRate = [2], [1], [0.1], [[]]
    else:
        bRate, dRate, sRate, mRate = IO.read_rates(args.rates)
    if args.populationModel is None:
        pops = ([1000000], [1], [1], [1], [1], [1])
        migrationRates = [[0.0]]
    else:
        pops = IO.read_populations(args.populationModel[0])
        migrationRates = IO.read_matrix(args.populationModel[1])
    if args.susceptibility is None:
        susceptible, susType = [[1.0] for _ in bRate], [0 for _ in bRate]
    else:
        susceptible, susType = IO.read_susceptibility(args.susceptibility)
    suscepTransition = [[0.0]] if args.suscepTransition is None else IO.read_matrix(args.suscepTransition)
    seed = randrange(sys.maxsize) if args.seed is None else args.seed
    sim = Simulator(number_of_sites=int(math.log(len(bRate), 4)), populations_number=len(pops[0]),
                    number_of_susceptible_groups=len(susceptible[0]), seed=int(seed),
                    sampling_probability=args.sampling_probability)
    apply_settings(sim, bRate, dRate, sRate, mRate, *pops, migrationRates, susceptible, susType, suscepTransition)
    return sim, int(seed)


def main(argv=None):
    args = parser().parse_args(argv)
    if args.citation:
        Simulator.citation(None)
        return 0
    sim, seed = build_simulator(args)
    sim.simulate(args.iterations, args.iterations if args.sampleSize is None else args.sampleSize,
                 -1 if args.time is None else args.time, method=args.method)
    sim.genealogy(seed)
    if args.createNewick:
        sim.export_newick(args.createNewick)
    if args.writeMutations:
        sim.export_mutations(args.writeMutations)
    if args.writeMigrations:
        sim.export_migrations(args.writeMigrations)
    if args.output_chain_events:
        sim.export_chain_events(args.output_chain_events)
    return 0


if __name__ == "__main__":
    sys.exit(main())
