"""The reference's own example script (testing/example.py: parameters, direct phase, parameter changes, tau phase,
genealogy, Newick / mutation / migration export) run unchanged except for its import line — `import vgsim_amd as VGsim`
— on the GPU, in a scratch directory."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import vgsim_amd as VGsim
    import os
    import os.path

    number_of_sites = 2
    populations_number = 3
    number_of_susceptible_groups = 3
    simulator = VGsim.Simulator(number_of_sites, populations_number, number_of_susceptible_groups, seed=1234)
    simulator.set_transmission_rate(0.25)
    simulator.set_transmission_rate(0.5, haplotype="GG")
    simulator.set_recovery_rate(0.099)
    simulator.set_sampling_rate(0.001)
    mutation_rate=0.00003
    substitution_weights=[1,1,1,2]#ATCG
    simulator.set_mutation_rate(mutation_rate)
    simulator.set_mutation_probabilities(substitution_weights)
    simulator.set_mutation_rate(3*mutation_rate, haplotype='G*', mutation=1)
    simulator.set_susceptibility_type(1)
    simulator.set_susceptibility_type(2, haplotype='G*')
    simulator.set_susceptibility(0.1, susceptibility_type=1)
    simulator.set_susceptibility(0.5, susceptibility_type=1, haplotype='G*')
    simulator.set_susceptibility(0.0, susceptibility_type=2)
    simulator.set_immunity_transition(1/90, source=1, target=0)
    simulator.set_immunity_transition(1/180, source=2, target=0)
    simulator.set_population_size(10000000, population=0)
    simulator.set_population_size(5000000, population=1)
    simulator.set_population_size(1000000, population=2)
    simulator.set_migration_probability(10/365/2)
    simulator.set_sampling_multiplier(3, population=1)
    simulator.set_sampling_multiplier(0, population=2)
    simulator.set_npi([0.1, 0.01, 0.002])
    simulator.simulate(10000000, epidemic_time=110)
    simulator.set_immunity_transition(0.05, source=0, target=1)
    simulator.set_immunity_transition(0.05, source=0, target=2)
    simulator.set_contact_density(0.7, population=0)
    simulator.set_contact_density(0.7, population=1)
    simulator.set_migration_probability(2/365/2, source=0, target=2)
    simulator.set_migration_probability(2/365/2, source=1, target=2)
    simulator.simulate(1000, epidemic_time=210, method='tau')
    simulator.genealogy()
    os.chdir('testing')
    if os.path.exists('example_output') == False:
        os.mkdir('example_output')
    os.chdir('example_output')
    simulator.export_newick()
    simulator.export_mutations('mutations')
    simulator.export_migrations('migrations')
    print("EXAMPLE_DONE", simulator.simulation.sCounter, simulator.simulation.events.ptr)
""")


def test_reference_example_script_runs(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "testing").mkdir()
    script = tmp_path / "example.py"
    script.write_text(SCRIPT % root)
    p = subprocess.run([sys.executable, str(script)], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and "EXAMPLE_DONE" in out, out[-3000:]
    samples = int(out.split("EXAMPLE_DONE")[1].split()[0])
    d = tmp_path / "testing" / "example_output"
    nwk = (d / "tree.nwk").read_text()
    assert nwk.endswith(";") and nwk.count(",") == samples - 1          # a binary tree over all samples
    assert len((d / "sample_population.tsv").read_text().splitlines()) == 2 * samples - 1
    assert (d / "mutations.tsv").exists() and (d / "migrations.tsv").read_text().startswith("Node\tTime")


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
