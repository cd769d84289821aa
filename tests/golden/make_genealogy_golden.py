#!/usr/bin/env python3
"""Golden vectors for the backward pass (GetGenealogy, reference src/_BirthDeath.pyx:743-1000), recorded from the
REFERENCE itself (development container only).

    tests/golden/build_reference.sh /tmp/vgsim_ref_build
    PYTHONPATH=/tmp/vgsim_ref_build:/tmp/vgsim_ref_build/stubs python3 tests/golden/make_genealogy_golden.py

For every (case, genealogy seed) the reference runs the forward phases of tests/models.py, then
``Simulator.genealogy(seed)``; recorded: the genealogy's INPUTS exactly as the reference held them (event chain,
final infectious counts — the multievent rows of tau cases are re-derived by the test from the forward golden) and
its OUTPUTS: tree, times (get_tree), the mutation records (output_tree_mutations) and the migration records
(export_migrations, parsed back from the TSV it writes; str(float) round-trips exactly).  Only data is written.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import models  # noqa: E402

import VGsim  # noqa: E402  (the reference build)

# (case of tests/models.py, genealogy seed or None = continue the simulation's stream)
GENEALOGY_CASES = [("g1_short", 11), ("g7_short", 12), ("g9_short", 13), ("g9_short", None), ("stress_h64", 14),
                   ("c3_s5_p16", 15), ("continuation", 16), ("p70", 17), ("tau_a", 18), ("tau_b", 19), ("tau_c", 20)]
# Chains with recombinant births (pyx:575-596): the BIRTH record names the parent haplotype while the new host carries
# the recombinant one (pyx:595-596), so upstream's backward pass loses lineages and returns a FOREST (node times left
# at 0, several roots); its writers then fail (KeyError in output_tree_mutations).  Recorded: that forest as get_tree
# returns it, and the number of mutation records print_mutations lists.
FOREST_CASES = [("recomb_a", 21), ("recomb_pos", 22)]


def run_forest(name, gseed):
    with contextlib.redirect_stdout(io.StringIO()):
        sim, phases = models.build(VGsim.Simulator, name)
        for setup, kw in phases:
            setup(sim)
            sim.simulate(**kw)
        sim.genealogy(gseed)
        tree, times = sim.simulation.get_tree()
        tree, times = np.asarray(tree).astype(np.int64).copy(), np.asarray(times).astype(np.float64).copy()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sim.print_mutations()
    n_mut = len(buf.getvalue().splitlines()) - 1
    meta = dict(case=name, genealogy_seed=gseed, mutations=n_mut, unset_times=int((times == 0).sum()))
    np.savez_compressed(os.path.join(HERE, "forest_%s_seed%d.npz" % (name, gseed)), meta=json.dumps(meta), tree=tree, times=times)
    print("%-28s nodes=%d unset=%d mutations=%d" % (name, len(tree), meta["unset_times"], n_mut))



def run(name, gseed):
    with contextlib.redirect_stdout(io.StringIO()):
        sim, phases = models.build(VGsim.Simulator, name)
        for setup, kw in phases:
            setup(sim)
            sim.simulate(**kw)
        m = sim.simulation
        tmp = "/tmp/_gen_chain_%s" % name
        sim.export_chain_events(tmp)
        chain = np.load(tmp + ".npy")
        os.remove(tmp + ".npy")
        inf_before = np.asarray(m.infectious).astype(np.int64).copy()
        sim.genealogy(gseed)
        tree, times = m.get_tree()
        tree = np.asarray(tree).astype(np.int64).copy()
        times = np.asarray(times).astype(np.float64).copy()
        _, _, mut, _ = m.output_tree_mutations()
        m.export_migrations("_gen_mig_%s" % name, "/tmp")
        # the reference's text writers (src/IO.py:144-255) on the same genealogy
        sim.export_newick("_gen_%s" % name, "/tmp")
        sim.export_mutations("_gen_mut_%s" % name, "/tmp")
    texts = {}
    for key, fn in (("newick", "/tmp/_gen_%s_tree.nwk" % name), ("sample_population", "/tmp/_gen_%s_sample_population.tsv" % name),
                    ("mutations_tsv", "/tmp/_gen_mut_%s.tsv" % name)):
        texts[key] = open(fn).read()
        os.remove(fn)
    rows = [l.split("\t") for l in open("/tmp/_gen_mig_%s.tsv" % name).read().splitlines()[1:]]
    os.remove("/tmp/_gen_mig_%s.tsv" % name)
    mig_node = np.array([int(r[0]) for r in rows], dtype=np.int64)
    mig_time = np.array([float(r[1]) for r in rows], dtype=np.float64)
    mig_old = np.array([int(r[2]) for r in rows], dtype=np.int64)
    mig_new = np.array([int(r[3]) for r in rows], dtype=np.int64)
    inf_after = np.asarray(m.infectious).astype(np.int64)
    tag = "%s_seed%s" % (name, "None" if gseed is None else gseed)
    nz = np.argwhere(inf_before != 0)
    nz2 = np.argwhere(inf_after != 0)
    meta = dict(case=name, genealogy_seed=gseed, sCounter=int((len(tree) + 1) // 2), events=int(chain.shape[1]), **texts)
    np.savez_compressed(
        os.path.join(HERE, "genealogy_" + tag + ".npz"), meta=json.dumps(meta), tree=tree, times=times,
        mut_node=np.array(mut[0], dtype=np.int64), mut_AS=np.array(mut[1], dtype=np.int64),
        mut_site=np.array(mut[2], dtype=np.int64), mut_DS=np.array(mut[3], dtype=np.int64),
        mut_time=np.array(mut[4], dtype=np.float64), mig_node=mig_node, mig_time=mig_time, mig_old=mig_old, mig_new=mig_new,
        infectious_before_nz=np.concatenate([nz, inf_before[inf_before != 0][:, None]], axis=1),
        infectious_after_nz=np.concatenate([nz2, inf_after[inf_after != 0][:, None]], axis=1))
    print("%-28s samples=%d nodes=%d mutations=%d migrations=%d" % (tag, meta["sCounter"], len(tree), len(mut[0]), len(mig_node)))


if __name__ == "__main__":
    want = sys.argv[1:]
    for name, gseed in GENEALOGY_CASES:
        if want and name not in want:
            continue
        run(name, gseed)
    for name, gseed in FOREST_CASES:
        if want and name not in want:
            continue
        run_forest(name, gseed)
