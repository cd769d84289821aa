"""Backward pass on chains produced by the HIP engine (GPU): forward simulation on the device, genealogy in libvgx's
host code, against the CPU oracle running the same two stages.  Direct chains are bit-identical to the oracle's
(portable log), so trees, node times, mutation and migration records must be identical too — including
``genealogy(None)``, which continues the simulation's random stream from the position the kernel reports
(last attempt, loop iterations).  Tau chains (Philox on the device: other chains than the oracle's own) are walked back by
both passes and the trees compared, plus structure and reproducibility."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

KEYS = ("tree", "times", "tree_pop")


def _product(sim):
    m = sim.simulation
    return {"tree": m.tree, "times": m.times, "tree_pop": m.tree_pop, "mut_node": np.array(m.mut.nodeId), "mut_AS": np.array(m.mut.AS),
            "mut_DS": np.array(m.mut.DS), "mut_site": np.array(m.mut.site), "mut_time": np.array(m.mut.time),
            "mig_node": np.array(m.mig.nodeId), "mig_time": np.array(m.mig.time), "mig_old": np.array(m.mig.oldPop),
            "mig_new": np.array(m.mig.newPop)}


@pytest.mark.parametrize("name", ["g9_short", "stress_h64", "c3_s5_p16", "p70", "continuation", "extinct_restart"])
@pytest.mark.parametrize("gseed", [None, 4711])
def test_direct_pipeline_matches_oracle(oracle_mod, name, gseed):
    hip = helpers.run_case_hip(name)
    ref = helpers.run_case_oracle(oracle_mod, name)
    if ref.simulation.sCounter < 2:
        pytest.skip("fewer than two samples")
    with helpers.quiet():
        hip.genealogy(gseed)
    want = oracle_mod.run_genealogy(ref.simulation, gseed)
    assert want["rc"] == 0
    got = _product(hip)
    for k in got:
        assert np.array_equal(got[k], want[k]), "%s %s" % (name, k)
    assert np.array_equal(hip.simulation.infectious, ref.simulation.infectious)


def _tree_is_valid(m):
    s = m.sCounter
    tree, times = m.tree, m.times
    assert len(tree) == 2 * s - 1
    roots = np.nonzero(tree == -1)[0]
    assert len(roots) == 1
    child = np.nonzero(tree >= 0)[0]
    assert (times[tree[child]] <= times[child]).all()          # a parent is older than its children
    assert np.bincount(tree[child], minlength=len(tree)).max() == 2 and (np.bincount(tree[child]) != 1).all()


@pytest.mark.parametrize("name", ["tau_b", "tau_c", "tau_d"])
def test_tau_pipeline_tree_equals_the_oracles_backward_pass(oracle_mod, name):
    """A tau chain produced on the device (events + multievent rows brought into the reference's order and granularity by
    the host layer) walked back by libvgx's host pass and, on a second identical run, by the oracle's backward pass
    (pinned on trees recorded from the reference): the same tree, node times, mutation and migration records."""
    hip = helpers.run_case_hip(name)
    if hip.simulation.sCounter < 2:
        pytest.skip("fewer than two samples")
    with helpers.quiet():
        hip.genealogy(99)
    twin = helpers.run_case_hip(name)
    want = oracle_mod.run_genealogy(twin.simulation, 99, multievents=twin.simulation.multievents)
    assert want["rc"] == 0
    got = _product(hip)
    for k in got:
        assert np.array_equal(got[k], want[k]), "%s %s" % (name, k)


@pytest.mark.parametrize("name", ["tau_b", "tau_c"])
def test_tau_pipeline_structure_and_reproducibility(name):
    runs = []
    for _ in range(2):
        sim = helpers.run_case_hip(name)
        m = sim.simulation
        # canonical multievent rows: per step sorted like UpdateCompartmentCounts_tau writes them, equal channels merged
        mv = m.multievents
        assert (mv.num[:mv.ptr] > 0).all()
        with helpers.quiet():
            sim.genealogy(99)
        _tree_is_valid(m)
        runs.append((m.tree.copy(), m.times.copy(), list(m.mut.nodeId), list(m.mig.nodeId), mv.num[:mv.ptr].copy()))
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)


def test_ensemble_replicate_genealogy_equals_single_run():
    """Genealogy of replicate r of an ensemble == genealogy of a single run with that replicate's seed."""
    from vgsim_amd import Simulator
    from vgsim_amd.ensemble import Ensemble
    import models
    with helpers.quiet():
        sim, phases = models.build(Simulator, "c3_s5_p16")
        phases[0][0](sim)
    R = 8
    ens = Ensemble(sim, R, seeds=np.arange(500, 500 + R))
    ens.simulate(6000, sample_size=10 ** 9, record_events=True)
    for r in (0, 5):
        with helpers.quiet():
            one, ph = models.build(Simulator, "c3_s5_p16")
            ph[0][0](one)
            one.simulation.user_seed = 500 + r
            one.simulate(6000, sample_size=10 ** 9)
        if one.simulation.sCounter < 2:
            continue
        for gseed in (None, 31):
            got = ens.genealogy(r, gseed)
            with helpers.quiet():
                one.genealogy(gseed)
            assert np.array_equal(got["tree"], one.simulation.tree) and np.array_equal(got["times"], one.simulation.times)
            break   # the single-run model's infectious array has been walked back: one pass per model
    ens.close()
