"""The workflow of the reference's example (testing/example.py) end to end on the GPU: the example model of
``tests/models.py`` (``_example_1`` / ``_example_2`` restate its parameters), a direct phase up to an epidemic time,
parameter changes, a tau-leaping call, the backward pass and the three exports.  Plus the command line."""
import os

import pytest

import helpers
import models

pytestmark = pytest.mark.gpu


def test_example_workflow_direct_tau_genealogy_exports(tmp_path, monkeypatch):
    from vgsim_amd import Simulator
    monkeypatch.chdir(tmp_path)
    with helpers.quiet():
        sim, phases = models.build(Simulator, "example")
        setup, kw = phases[0]
        setup(sim)
        sim.simulate(**kw)                                   # direct, stops at epidemic time 110
        m = sim.simulation
        direct_events, direct_samples = m.events.ptr, m.sCounter
        helpers.check_against_golden(m, "example", exact_time=helpers.libm_matches_fixture_host(), leftovers=False)
        models._example_2(sim)                               # parameter changes between the calls
        sim.simulate(1000, epidemic_time=210, method='tau')  # a no-op upstream too: sample_size defaults to 1000 < sCounter
        sim.genealogy()
        sim.export_newick()
        sim.export_mutations('mutations')
        sim.export_migrations('migrations')
    assert (m.events.ptr, m.sCounter) == (direct_events, direct_samples)
    samples = int(m.sCounter)
    nwk = (tmp_path / "tree.nwk").read_text()
    assert nwk.endswith(";") and nwk.count(",") == samples - 1          # a binary tree over all samples
    assert len((tmp_path / "sample_population.tsv").read_text().splitlines()) == 2 * samples - 1
    assert (tmp_path / "mutations.tsv").exists() and (tmp_path / "migrations.tsv").read_text().startswith("Node\tTime")


def test_command_line_end_to_end(tmp_path, monkeypatch):
    """`python -m vgsim_amd.cmd` with the reference's example settings files: simulate on the device, genealogy, all four
    output files (VGsim_cmd.py:144-154)."""
    import contextlib
    import io
    import os

    import numpy as np

    from vgsim_amd import cmd
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cmd_example")
    monkeypatch.chdir(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()) as out:
        rc = cmd.main(["-it", "12000", "-seed", "17", "-rt", os.path.join(d, "example.rt"), "-pm", os.path.join(d, "example.pp"),
                       os.path.join(d, "example.mg"), "-su", os.path.join(d, "example.su"), "-st", os.path.join(d, "example.st"),
                       "-nwk", "run", "-tsv", "run_mut", "--writeMigrations", "run_mig", "--output_chain_events", "run_chain"])
    assert rc == 0 and "Success number: 4" in out.getvalue()        # the same run as the cmd_example golden (seed 17)
    for f in ("run_tree.nwk", "run_sample_population.tsv", "run_mig.tsv", "run_chain.npy"):
        assert os.path.getsize(f) > 0, f
    assert os.path.exists("run_mut.tsv")     # no mutation happens at the example's rates (3e-6 per site)
    chain = np.load("run_chain.npy")
    assert chain.shape == (6, 12000) and open("run_tree.nwk").read().endswith(";")
