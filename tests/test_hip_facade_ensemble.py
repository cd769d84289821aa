"""The ensemble entry of the facade (``Simulator.ensemble`` / ``simulate_ensemble``): replicate r of the ensemble is the run a
single ``Simulator`` with that seed makes (bit for bit), for the direct path on both kernels (64 replicates: one per wavefront;
2048: four per wavefront) — the intended way to use the engine for small models."""
import numpy as np
import pytest

import helpers
import models

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rep", [64, 2048])
def test_simulate_ensemble_replicates_equal_single_simulators(n_rep):
    from vgsim_amd import Simulator
    with helpers.quiet():
        sim, phases = models.build(Simulator, "g8_short")
        phases[0][0](sim)
        ens, res = sim.simulate_ensemble(n_rep, 2000, sample_size=10 ** 9)
    base = sim.simulation.user_seed
    assert res.events.shape == (n_rep,) and (res.events == 2000).all()
    for r in (0, 1, n_rep // 2 + 3, n_rep - 1):
        with helpers.quiet():
            one, ph = models.build(Simulator, "g8_short")
            one.simulation.user_seed = base + r
            ph[0][0](one)
            one.simulate(2000, sample_size=10 ** 9)
        m = one.simulation
        assert np.array_equal(ens.replicate_events(r), m.events.as_array()[:, :m.events.ptr]), "replicate %d" % r
        st = ens.replicate_state(r)
        assert np.array_equal(st.infectious, m.infectious) and st.currentTime == m.currentTime
    ens.close()


def test_limits_are_reported_not_hidden():
    """More rate classes than the direct kernel's tables hold, and tau population sizes >= 2^31: error codes, no crash."""
    from vgsim_amd import Simulator
    from vgsim_amd._capi import VgxError
    with helpers.quiet():
        s = Simulator(number_of_sites=6, seed=1)                 # 4096 haplotypes, each with its own transmission rate
        for h in range(0, 4096, 3):
            s.set_transmission_rate(2.0 + h * 1e-4, haplotype=h)
    with pytest.raises(VgxError) as ei, helpers.quiet():
        s.simulate(500)
    assert ei.value.code == 6                                    # VGX_ERR_CLASSES
    with helpers.quiet():
        t = Simulator(number_of_sites=1, seed=1)
        t.set_population_size(2 ** 31)
    with pytest.raises(VgxError) as ei, helpers.quiet():
        t.simulate(300, method='tau', sample_size=10 ** 12)
    assert ei.value.code == 1 and "2^31" in str(ei.value)        # VGX_ERR_ARG
    with helpers.quiet():
        u = Simulator(number_of_sites=4, populations_number=96, number_of_susceptible_groups=4, seed=1)   # 256 transmission classes x 96 x 4
        for h in range(256):
            u.set_transmission_rate(2.0 + h * 1e-3, haplotype=h)
    with pytest.raises(VgxError) as ei, helpers.quiet():
        u.simulate(500)
    assert ei.value.code == 1 and "LDS" in str(ei.value)
