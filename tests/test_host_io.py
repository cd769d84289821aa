"""Host-side formats next to the hot path: settings-file readers pinned on the reference's own
example data files, settings round trip, timelines, printouts.  No GPU."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from vgsim_amd import IO, Simulator
from vgsim_amd._report import host_propensities, host_rates

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "golden", "cmd_example")


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r = fn(*a, **k)
    return r, buf.getvalue()


def as_json(x):
    return json.loads(json.dumps(x))


def test_readers_match_reference_on_its_example_files():
    """tests/golden/io_readers.json = what the reference's IO.py returned for the same files (make_io_golden.py)."""
    with open(os.path.join(HERE, "golden", "io_readers.json")) as f:
        want = json.load(f)
    assert as_json(IO.read_rates(os.path.join(D, "example.rt"))) == want["read_rates"]
    assert as_json(IO.read_populations(os.path.join(D, "example.pp"))) == want["read_populations"]
    assert as_json(IO.read_matrix(os.path.join(D, "example.mg"))) == want["read_matrix_mg"]
    assert as_json(IO.read_susceptibility(os.path.join(D, "example.su"))) == want["read_susceptibility"]
    assert as_json(IO.read_matrix(os.path.join(D, "example.st"))) == want["read_matrix_st"]


def test_rates_sampling_probability_header(tmp_path):
    p = tmp_path / "r.rt"
    p.write_text("#v\nB D SP\n2.0 1.0 0.25\n")
    b, d, s, m = IO.read_rates(str(p))
    assert (b, d, s, m) == ([2.0], [0.75], [0.25], [[]])


def build_model():
    sim, _ = quiet(Simulator, 2, 3, 2, seed=7)
    sim.set_transmission_rate(3.5, 'A*')
    sim.set_recovery_rate(0.7, 3)
    sim.set_sampling_rate(0.05)
    sim.set_mutation_rate(0.02, None, 1)
    sim.set_mutation_probabilities([1, 2, 4, 3], 0, 0)
    sim.set_susceptibility_type(1, 'T*')
    sim.set_susceptibility(0.4, None, 1)
    sim.set_immunity_transition(0.03, 1, 0)
    sim.set_population_size(2000000, 1)
    sim.set_contact_density(0.8, 2)
    sim.set_npi([0.1, 0.01, 0.002], 0)
    sim.set_sampling_multiplier(2.5, 1)
    sim.set_migration_probability(0.004, 0, 2)
    sim.set_migration_probability(0.001, 2, 1)
    return sim


PARAMS = ("bRate", "dRate", "sRate", "mRate", "hapMutType", "suscType", "susceptibility", "suscepTransition", "sizes",
          "contactDensity", "contactDensityAfterLockdown", "startLD", "endLD", "samplingMultiplier", "migrationRates")


def test_settings_round_trip(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    a = build_model()
    _, out = quiet(a.export_settings, "params")
    assert out.strip() == ("Command line command: params/params.rt -pm params/params.pp params/params.mg "
                           "-su params/params.su -st params/params.st")
    assert sorted(os.listdir("params")) == ["params.mg", "params.pp", "params.rt", "params.st", "params.su"]
    # layout of pyx:1860-1872: version line, header, one row per haplotype with "rate,w0,w1,w2" mutation columns
    rt = open("params/params.rt").read().splitlines()
    assert rt[0] == "#Rates_format_version 0.0.1" and rt[1] == "H B D S M0 M1"
    assert rt[2] == "AA 3.5 1.0 0.05 0.01,2.0,4.0,3.0 0.02,1.0,1.0,1.0 "
    pp = open("params/params.pp").read().splitlines()
    assert pp[2] == "0 1000000 1.0 0.1,0.01,0.002 1.0"
    b, _ = quiet(Simulator, 2, 3, 2, seed=7)
    b.set_settings("params")
    for k in PARAMS:
        assert np.array_equal(getattr(a.simulation, k), getattr(b.simulation, k)), k
    # the files the writer produced load through the same readers the reference's command line uses
    wrong, _ = quiet(Simulator, 1, 3, 2, seed=7)
    with pytest.raises(ValueError):
        wrong.set_settings("params")


def test_reference_example_files_apply_through_the_setters():
    """VGsim_cmd.py:113-142 applied to the reference's example files (2 sites, 3 populations, 3 groups)."""
    from vgsim_amd._report import apply_settings
    b, d, s, m = IO.read_rates(os.path.join(D, "example.rt"))
    pops = IO.read_populations(os.path.join(D, "example.pp"))
    mg = IO.read_matrix(os.path.join(D, "example.mg"))
    su, st_type = IO.read_susceptibility(os.path.join(D, "example.su"))
    st = IO.read_matrix(os.path.join(D, "example.st"))
    sim, _ = quiet(Simulator, 2, len(pops[0]), len(su[0]), seed=1)
    apply_settings(sim, b, d, s, m, *pops, mg, su, st_type, st)
    mdl = sim.simulation
    assert np.array_equal(mdl.bRate, b) and np.array_equal(mdl.dRate, d) and np.array_equal(mdl.sRate, s)
    assert np.array_equal(mdl.sizes, pops[0]) and np.array_equal(mdl.samplingMultiplier, pops[5])
    assert np.array_equal(mdl.susceptibility, np.array(su, dtype=float)) and np.array_equal(mdl.suscType, st_type)
    off = ~np.eye(len(mg), dtype=bool)
    assert np.array_equal(mdl.migrationRates[off], np.array(mg)[off])
    # haplotype AA, site 0: weights [0 (own allele A), 1, 1, 2] -> the three derived states T, C, G
    assert np.array_equal(mdl.hapMutType[0, 0], [1.0, 1.0, 2.0])


def fill_chain(sim):
    """A hand-made log: birth, mutation, migration, immunity change, recovery, sampling."""
    m = sim.simulation
    rows = [(0.5, 0, 0, 0, 0, 16), (1.0, 3, 0, 0, 5, 0), (1.5, 5, 5, 0, 0, 1), (2.0, 4, 0, 1, 1, 0),
            (2.5, 1, 0, 0, 0, 0), (3.0, 2, 5, 1, 1, 0)]
    m.events.CreateEvents(len(rows))
    for k, r in enumerate(rows):
        m.events.times[k] = r[0]
        for c, v in zip(("types", "haplotypes", "populations", "newHaplotypes", "newPopulations"), r[1:]):
            getattr(m.events, c)[k] = v
    m.events.ptr = len(rows)
    m.currentTime = 3.0
    m.initial_susceptible[:, 0] = m.sizes
    m.initial_susceptible[0, 0] -= 1
    m.initial_infectious[0, 0] = 1


def test_epidemiology_timelines(tmp_path, monkeypatch):
    sim, _ = quiet(Simulator, 2, 2, 2, seed=3)
    fill_chain(sim)
    log = sim.output_epidemiology_timelines(step=3)
    assert log["time"] == [0.0, 1.0, 2.0, 3.0]
    # rows are emitted after events 0 (t=.5 >= 0), 1 (1.0 >= 1.0), 3 (2.0 >= 2.0), 5 (3.0 >= 3.0)
    assert log["P0"]["H0"] == [2, 1, 1, 0] and log["P0"]["H5"] == [0, 1, 1, 1]
    assert log["P1"]["H5"] == [0, 0, 1, 0]
    assert log["P0"]["S0"] == [999998, 999998, 999998, 999999]
    assert log["P1"]["S0"] == [1000000, 1000000, 999998, 999998] and log["P1"]["S1"] == [0, 0, 1, 2]
    monkeypatch.chdir(tmp_path)
    assert sim.output_epidemiology_timelines(step=3, output_file=True) is None
    lines = open("logs/PID1.log").read().splitlines()
    assert lines[0] == "time S0 S1 " + " ".join("H%d" % h for h in range(16))
    assert lines[3].split()[:3] == ["2.0", "999998", "1"] and len(lines) == 5
    # the replay used by print_populations agrees with the end of the timeline
    sus, inf = sim.simulation.GetCurrentIndividuals()
    assert [sus[1][0], sus[1][1], inf[1][5], inf[0][0]] == [999998, 2, 0, 0]


def test_printouts_and_facade_surface():
    sim = build_model()
    fill_chain(sim)
    for call in (sim.print_basic_parameters, sim.print_populations, sim.print_immunity_model, sim.debug,
                 sim.print_propensities, sim.print_chain, sim.print_counters, sim.citation,
                 lambda: sim.print_all(True, True, True, True, True, True, True, True)):
        _, out = quiet(call)
        assert out
    _, out = quiet(sim.print_populations, True, False, False, False)
    assert "Actual size" in out and "| 1  | 2000000 |" in out
    assert list(sim.get_indexes_from_haplotype('A*')) == [0, 1, 2, 3]
    assert list(sim.get_indexes_from_haplotype([5, 'GG'])) == [5, 15]
    # every public method of the reference's facade (src/_interface.py) exists here
    names = """add_legend add_plot_infectious add_plot_susceptible add_title citation debug export_chain_events
        export_migrations export_mutations export_newick export_settings export_state export_ts genealogy
        get_data_infectious get_data_susceptible get_indexes_from_haplotype get_proportion get_tree
        output_epidemiology_timelines output_sample_data plot plot_infectious print_all print_basic_parameters
        print_chain print_counters print_immunity_model print_migrations print_mutations print_populations
        print_propensities print_recomb print_tree set_chain_events set_coinfection_parameters set_contact_density
        set_genome_length set_immunity_transition set_infectious set_initial_haplotype set_migration_probability
        set_mutation_position set_mutation_probabilities set_mutation_rate set_npi set_population_size set_recovery_rate
        set_sampling_multiplier set_sampling_rate set_settings set_state set_step_haplotype set_susceptibility
        set_susceptibility_type set_susceptible set_total_migration_probability set_transmission_rate simulate""".split()
    for n in names:
        assert callable(getattr(sim, n)), n


def test_host_rates_match_hand_values():
    """UpdateAllRates (pyx:279-351) on a two-population model worked by hand."""
    sim, _ = quiet(Simulator, 0, 2, 1, seed=1)
    sim.set_migration_probability(0.1)
    m = sim.simulation
    m.infectious[0, 0] = 10
    m.susceptible[0, 0] -= 10
    r = host_rates(m)
    assert np.allclose(r["migrationRates"], [[0.9, 0.1], [0.1, 0.9]])
    assert np.allclose(r["actualSizes"], [1e6, 1e6])
    coef = (0.81 + 0.01) / 1e6
    assert np.isclose(r["eventHapPopRate"][0, 0, 0], 2.0 * (1e6 - 10) * coef)
    assert np.isclose(r["effectiveMigration"][0, 1], 2 * 0.09 / 1e6) and r["effectiveMigration"][0, 0] == 0
    assert np.isclose(r["migPopRate"][1], r["effectiveMigration"][0, 1] * 2.0 * 1e6 * 10)
    p = host_propensities(m)
    assert np.isclose(p["PropensitiesTransmission"][0, 0, 0], 2.0 * coef * (1e6 - 10) * 10)
    assert np.isclose(p["PropensitiesMigr"][0, 1, 0, 0], r["effectiveMigration"][1, 0] * 1e6 * 10 * 2.0 * 0.9)
    assert p["PropensitiesRecovery"][0, 0] == 10.0


def test_command_line_builds_the_model_from_the_example_files():
    from vgsim_amd import cmd
    args = cmd.parser().parse_args(["-it", "500", "-seed", "11", "-rt", os.path.join(D, "example.rt"),
                                    "-pm", os.path.join(D, "example.pp"), os.path.join(D, "example.mg"),
                                    "-su", os.path.join(D, "example.su"), "-st", os.path.join(D, "example.st")])
    (sim, seed), _ = quiet(cmd.build_simulator, args)
    m = sim.simulation
    assert (seed, m.sites, m.popNum, m.susNum, args.iterations) == (11, 2, 3, 3, 500)
    assert m.bRate[0] == 0.25 and m.sizes[1] == 5000000 and m.suscType[0] == 1
    (sim, _), _ = quiet(cmd.build_simulator, cmd.parser().parse_args(["-seed", "3"]))
    assert (sim.simulation.hapNum, sim.simulation.popNum, sim.simulation.sRate[0]) == (1, 1, 0.1)
    _, out = quiet(cmd.main, ["-c"])
    assert "VGsim" in out


def test_plot_helpers(tmp_path):
    matplotlib = pytest.importorskip("matplotlib")
    matplotlib.use("Agg")
    sim, _ = quiet(Simulator, 2, 2, 2, seed=3)
    fill_chain(sim)
    sim.simulation.loc.AddLockdown(True, 0, 1.2)
    sim.simulation.loc.AddLockdown(False, 0, 2.2)
    sim.add_plot_infectious(0, 0, step_num=3)
    sim.add_plot_infectious(0, 'A*', step_num=3, label_infectious="inf", label_samples="smp")
    sim.add_plot_susceptible(1, 1, step_num=3)
    sim.add_legend()
    sim.add_title("trajectories")
    out = tmp_path / "p.png"
    sim.plot(str(out))
    assert out.stat().st_size > 1000 and sim.fig is None


def test_readers_and_the_setter_sequence_build_the_golden_model():
    """The `cmd_example` parity case (tests/models.py) parses the example files by hand; the shipped readers + the
    command line's setter sequence must build the identical model."""
    import models
    from vgsim_amd import cmd
    (a, _), _ = quiet(cmd.build_simulator, cmd.parser().parse_args(
        ["-seed", "17", "-rt", os.path.join(D, "example.rt"), "-pm", os.path.join(D, "example.pp"), os.path.join(D, "example.mg"),
         "-su", os.path.join(D, "example.su"), "-st", os.path.join(D, "example.st")]))
    (b, phases), _ = quiet(models.build, Simulator, "cmd_example")
    phases[0][0](b)
    for k in PARAMS:
        assert np.array_equal(getattr(a.simulation, k), getattr(b.simulation, k)), k
