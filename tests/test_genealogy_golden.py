"""Backward pass (GetGenealogy, reference src/_BirthDeath.pyx:743-1000): the CPU oracle against golden vectors
recorded from the reference itself (tests/golden/make_genealogy_golden.py) — tree, node times, mutation and
migration records and the walked-back infectious counts, bit for bit, for direct chains, for a genealogy that
continues the simulation's random stream (seed=None) and for tau chains (MULTITYPE events: numpy's
random_hypergeometric thinning)."""
import glob
import json
import os

import numpy as np
import pytest

import helpers

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "genealogy_*.npz")))


def load(path):
    z = np.load(path, allow_pickle=False)
    return json.loads(str(z["meta"])), z


def dense(nz, shape):
    a = np.zeros(shape, dtype=np.int64)
    if len(nz):
        a[nz[:, 0], nz[:, 1]] = nz[:, 2]
    return a


def assert_genealogy_equal(out, z, what):
    assert np.array_equal(out["tree"], z["tree"]), what + " tree"
    assert np.array_equal(out["times"], z["times"]), what + " times"
    for k in ("mut_node", "mut_AS", "mut_DS", "mut_site", "mut_time", "mig_node", "mig_time", "mig_old", "mig_new"):
        assert np.array_equal(out[k], z[k]), "%s %s: %r != %r" % (what, k, out[k][:8], z[k][:8])


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[10:-4] for p in GOLD])
def test_oracle_genealogy_matches_reference(oracle_mod, path):
    meta, z = load(path)
    sim = helpers.run_case_oracle(oracle_mod, meta["case"], record_multievents=True)
    m = sim.simulation
    assert m.sCounter == meta["sCounter"]
    assert np.array_equal(m.infectious, dense(z["infectious_before_nz"], m.infectious.shape))
    out = oracle_mod.run_genealogy(m, meta["genealogy_seed"])
    assert out["rc"] == 0
    assert_genealogy_equal(out, z, meta["case"])
    assert np.array_equal(m.infectious, dense(z["infectious_after_nz"], m.infectious.shape)), "walked-back infectious"


def _product_genealogy(m, seed, rng_raw=None):
    from vgsim_amd import _capi
    return _capi.get_genealogy(m, seed, rng_raw=rng_raw)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[10:-4] for p in GOLD])
def test_product_genealogy_matches_reference(oracle_mod, path):
    """The shipped backward pass (libvgx host code, called through the C ABI; needs no GPU) on the same chains."""
    meta, z = load(path)
    sim = helpers.run_case_oracle(oracle_mod, meta["case"], record_multievents=True)
    m = sim.simulation
    st = oracle_mod.get_state(m)
    helpers.sparse_multievents(m, st)
    raw = None
    if meta["genealogy_seed"] is None:   # continue where the forward run's generator stands
        raw = tuple(st.rng_final) + (0, 0)
    out = _product_genealogy(m, meta["genealogy_seed"], raw)
    assert_genealogy_equal(out, z, meta["case"])
    assert np.array_equal(m.infectious, dense(z["infectious_after_nz"], m.infectious.shape)), "walked-back infectious"


def test_product_genealogy_through_the_model_api(oracle_mod):
    """Simulator.genealogy / get_tree / export_migrations on a chain handed over with the reference's own names."""
    meta, z = load([p for p in GOLD if "g9_short_seed13" in p][0])
    sim = helpers.run_case_oracle(oracle_mod, meta["case"])
    sim.genealogy(13)
    tree, times = sim.get_tree()
    assert np.array_equal(tree, z["tree"]) and np.array_equal(times, z["times"])
    t, tm, mut, pops = sim.simulation.output_tree_mutations()
    assert mut[0] == z["mut_node"].tolist() and mut[1] == z["mut_AS"].tolist() and mut[2] == z["mut_site"].tolist()
    assert mut[3] == z["mut_DS"].tolist() and mut[4] == z["mut_time"].tolist()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        sim.export_migrations("mig", d)
        rows = [l.split("\t") for l in open(os.path.join(d, "mig.tsv")).read().splitlines()[1:]]
    assert [int(r[0]) for r in rows] == z["mig_node"].tolist()
    assert [float(r[1]) for r in rows] == z["mig_time"].tolist()
    assert [int(r[2]) for r in rows] == z["mig_old"].tolist() and [int(r[3]) for r in rows] == z["mig_new"].tolist()


def test_hypergeometric_matches_numpy(oracle_mod):
    """numpy's random_hypergeometric restated in the oracle: same draws as numpy.random.Generator(PCG64) on both
    branches (direct sampling below 10 draws, HRUA above) from the same stream position."""
    import ctypes as C
    rs = np.random.RandomState(5)
    lib = oracle_mod.lib()
    for trial in range(300):
        good, bad = int(rs.randint(0, 400)), int(rs.randint(0, 400))
        if good + bad == 0:
            continue
        sample = int(rs.randint(0, good + bad + 1))
        seed = int(rs.randint(1, 2 ** 31))
        gen = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed, spawn_key=(0,))))
        want = [int(gen.hypergeometric(good, bad, sample)) for _ in range(5)] if sample > 0 else [0] * 5
        r = oracle_mod.VgoGenRng()
        lib.vgo_pcg64_seed(C.byref(r.g), seed, 0)
        got = [int(lib.vgo_hypergeometric(C.byref(r), good, bad, sample)) for _ in range(5)] if sample > 0 else [0] * 5
        assert got == want, (good, bad, sample, seed, got, want)


def test_rng_position_matches_numpy():
    """vgx_rng_position(seed, attempt, draws): the state numpy's PCG64(SeedSequence(seed, spawn_key=(attempt,))) has
    after `draws` outputs (how GetGenealogy(seed=None) finds the simulation's stream on the host)."""
    import ctypes as C
    from vgsim_amd import _capi
    lib = _capi.load_library()
    for seed, att, draws in ((2020, 0, 0), (2020, 3, 7), (1234, 1, 10 ** 6 + 1), (2 ** 40 + 5, 2, 123456789)):
        bg = np.random.PCG64(np.random.SeedSequence(seed, spawn_key=(att,)))
        bg.advance(draws)
        st = bg.state["state"]
        out = (C.c_uint64 * 4)()
        lib.vgx_rng_position(seed, att, draws, C.byref(out))
        assert (int(out[0]) << 64 | int(out[1])) == st["state"] and (int(out[2]) << 64 | int(out[3])) == st["inc"]


def test_chain_round_trip_feeds_the_backward_pass(oracle_mod, tmp_path):
    """export_chain_events -> set_chain_events -> genealogy gives the tree of the original log."""
    meta, z = load([p for p in GOLD if "c3_s5_p16_seed15" in p][0])
    sim = helpers.run_case_oracle(oracle_mod, meta["case"])
    sim.export_chain_events(str(tmp_path / "chain"))
    final_inf = sim.simulation.infectious.copy()
    sim.simulation.events.ptr = 0
    sim.set_chain_events(str(tmp_path / "chain"))
    assert np.array_equal(sim.simulation.infectious, final_inf)
    sim.genealogy(15)
    tree, times = sim.get_tree()
    assert np.array_equal(tree, z["tree"]) and np.array_equal(times, z["times"])


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[10:-4] for p in GOLD])
def test_text_writers_match_reference(oracle_mod, path, tmp_path):
    """export_newick / export_mutations (src/IO.py:144-255) on the genealogy of every fixture: same bytes as the
    reference wrote.  (Tau chains: the sample-population table is skipped — upstream looks the population up in the
    MULTITYPE event's row-index field, which depends on the multievent layout.)"""
    meta, z = load(path)
    sim = helpers.run_case_oracle(oracle_mod, meta["case"], record_multievents=True)
    m = sim.simulation
    tau = bool((m.events.types[:m.events.ptr] == 6).any())
    out = oracle_mod.run_genealogy(m, meta["genealogy_seed"])
    m.tree, m.times, m.tree_pop = out["tree"], out["times"], out["tree_pop"]
    m.mut.nodeId, m.mut.AS, m.mut.DS = out["mut_node"].tolist(), out["mut_AS"].tolist(), out["mut_DS"].tolist()
    m.mut.site, m.mut.time = out["mut_site"].tolist(), out["mut_time"].tolist()
    sim.export_newick("t", str(tmp_path))
    sim.export_mutations("m", str(tmp_path))
    assert open(tmp_path / "t_tree.nwk").read() == meta["newick"]
    assert open(tmp_path / "m.tsv").read() == meta["mutations_tsv"]
    if not tau:
        assert open(tmp_path / "t_sample_population.tsv").read() == meta["sample_population"]


@pytest.mark.parametrize("name,gseed", [("recomb_a", 21), ("recomb_pos", 22)])
def test_recombinant_chain_gives_upstreams_forest_and_the_product_says_so(oracle_mod, name, gseed):
    """A recombinant BIRTH is logged under the parent haplotype although the new host carries the recombinant one
    (pyx:595-596), so the backward pass loses lineages: the reference returns a forest with unset node times
    (fixture recorded from it), the literal oracle reproduces that forest bit for bit and reports the condition,
    and the shipped pass raises instead of handing back a broken tree."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "forest_%s_seed%d.npz" % (name, gseed)))
    meta = json.loads(str(z["meta"]))
    m = helpers.run_case_oracle(oracle_mod, name).simulation
    out = oracle_mod.run_genealogy(m, gseed)
    assert out["rc"] != 0
    assert np.array_equal(out["tree"], z["tree"]) and np.array_equal(out["times"], z["times"])
    assert len(out["mut_node"]) == meta["mutations"] and int((out["times"] == 0).sum()) == meta["unset_times"] > 0
    m = helpers.run_case_oracle(oracle_mod, name).simulation
    with pytest.raises(RuntimeError, match="never coalesced"):
        _product_genealogy(m, gseed)
