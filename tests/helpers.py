"""Shared helpers for the parity tests (test infrastructure)."""
import contextlib
import hashlib
import io
import json
import math
import os

import numpy as np

import models

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, z


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def libm_matches_fixture_host():
    """The time row of the fixtures was produced with glibc 2.35's log() on the development container.
    glibc's log is not correctly rounded and has CPU-dispatched variants, so another host can differ in the
    last bit.  The bit-exact time assertion is made only where this known-answer probe (4096 recorded
    (u, log u) pairs, tests/golden/libm_probe.json) holds; elsewhere the documented tolerance applies."""
    global _LIBM_OK
    if _LIBM_OK is None:
        probe = json.load(open(os.path.join(GOLDEN, "libm_probe.json")))
        _LIBM_OK = all(math.log(float.fromhex(u)) == float.fromhex(v) for u, v in probe)
    return _LIBM_OK


_LIBM_OK = None


def run_case_oracle(oracle, name, sparse=False, log_mode=0, record_multievents=False):
    """Drive a case of tests/models.py through this repo's Simulator object + the CPU oracle."""
    from vgsim_amd import Simulator
    with quiet():
        sim, phases = models.build(Simulator, name)
    m = sim.simulation
    for setup, kw in phases:
        setup(sim)
        kw = dict(kw)
        it = kw.pop("iterations")
        ss = kw.pop("sample_size", None)
        ss = it if ss is None else ss
        tm = kw.pop("epidemic_time", -1)
        at = kw.pop("attempts", 200)
        method = kw.pop("method", "direct")
        if method == "direct":
            rc = oracle.run_direct(m, it, ss, tm, at, sparse=sparse, log_mode=log_mode)
        else:
            rc = oracle.run_tau(m, it, ss, tm, at, record_multievents=record_multievents, log_mode=log_mode)
        assert rc == 0, "oracle error code %d" % rc
    return sim


def chain_of(model):
    return model.events.as_array()


def check_against_golden(model, name, exact_time=True, rtol_time=1e-12, leftovers=True):
    """Compare a finished host model with the fixture recorded from the reference.  ``leftovers``: the model's log still
    holds, beyond ``events.ptr``, the rows of failed attempts exactly as the reference leaves them (true for the oracle,
    which writes into the host arrays; the engine copies out rows below ``ptr`` only)."""
    meta, z = load_golden(name)
    st = meta["stats"]
    chain = chain_of(model)
    ptr = st["ptr"]
    assert model.events.ptr == ptr
    assert chain.shape[1] == meta["size"]
    for k in ("sCounter", "bCounter", "dCounter", "mCounter", "migPlus", "migNonPlus", "iCounter", "good_attempt"):
        if k in st:
            assert getattr(model, k) == st[k], k
    # rows at and beyond events.ptr are leftovers of failed attempts (Restart only rewinds ptr, pyx:715);
    # they are compared only in the exact (sha256) branch below
    if "times" in z:
        ints = z["ints"].astype(np.int64)
        assert np.array_equal(chain[1:, :ptr].astype(np.int64), ints[:, :ptr]), "integer rows differ"
        ref_t = z["times"]
    else:
        head, tail = z["head"], z["tail"]
        assert np.array_equal(chain[1:, :256], head[1:]) and np.array_equal(chain[1:, ptr - tail.shape[1]:ptr], tail[1:])
        ref_t = None
    if "sha256_ints" in meta:   # all five integer rows of the log, full length (rows 1-5 do not depend on the host's libm)
        if leftovers or ptr == chain.shape[1]:
            assert hashlib.sha256(np.ascontiguousarray(chain[1:]).tobytes()).hexdigest() == meta["sha256_ints"], "integer rows (sha256)"
    if exact_time:
        if leftovers or ptr == chain.shape[1]:
            assert hashlib.sha256(np.ascontiguousarray(chain).tobytes()).hexdigest() == meta["sha256_chain"]
        if ref_t is not None:
            assert np.array_equal(chain[0, :ptr], ref_t[:ptr]), "time row differs"
        assert model.currentTime == st["currentTime"]
    else:
        # documented tolerance for the float row when log() is not the fixture host's libm
        if ref_t is not None:
            np.testing.assert_allclose(chain[0, :ptr], ref_t[:ptr], rtol=rtol_time, atol=0)
        else:
            np.testing.assert_allclose(chain[0, :256], z["head"][0], rtol=rtol_time, atol=0)
            np.testing.assert_allclose(chain[0, ptr - z["tail"].shape[1]:ptr], z["tail"][0], rtol=rtol_time, atol=0)
        assert abs(model.currentTime - st["currentTime"]) <= rtol_time * abs(st["currentTime"])
    assert np.array_equal(model.susceptible, z["susceptible"])
    nz = z["infectious_nz"]
    inf = np.zeros_like(model.infectious)
    if len(nz):
        inf[nz[:, 0], nz[:, 1]] = nz[:, 2]
    assert np.array_equal(model.infectious, inf)


def run_case_hip(name, phases_limit=None, mode=None, kernel=None):
    """Drive a case through the product path: Simulator -> BirthDeathModel -> ctypes -> libvgx.so -> HIP.
    ``mode='fast'`` runs the direct phases in the engine's FAST mode."""
    from vgsim_amd import Simulator
    with quiet():
        sim, phases = models.build(Simulator, name)
        for setup, kw in phases[:phases_limit]:
            setup(sim)
            kw = dict(kw)
            if mode is not None and kw.get("method", "direct") == "direct":
                kw["mode"] = mode
            if kernel is not None and kw.get("method", "direct") == "direct":
                kw["kernel"] = kernel
            sim.simulate(**kw)
    return sim


def describe_first_diff(a, b, ptr):
    """Human-readable report of the first differing log column between two (6, N) chains."""
    n = min(a.shape[1], b.shape[1], ptr)
    neq = np.nonzero((a[:, :n] != b[:, :n]).any(axis=0))[0]
    if len(neq) == 0:
        return "chains equal on the first %d columns (shapes %s vs %s)" % (n, a.shape, b.shape)
    i = int(neq[0])
    lo = max(0, i - 2)
    return "first difference at event %d of %d (%d differing):\n  got  %s\n  want %s\n context got:\n%s\n context want:\n%s" % (
        i, n, len(neq), a[:, i].tolist(), b[:, i].tolist(), a[:, lo:i + 2].T, b[:, lo:i + 2].T)


def assert_models_equal(got, want, what=""):
    """Bit-exact comparison of two finished host models (event chain incl. times, counters, compartments)."""
    assert got.events.ptr == want.events.ptr, "%s events.ptr %d != %d" % (what, got.events.ptr, want.events.ptr)
    a, b = chain_of(got), chain_of(want)
    ptr = want.events.ptr
    assert a.shape == b.shape
    assert np.array_equal(a[:, :ptr], b[:, :ptr]), what + " " + describe_first_diff(a, b, ptr)
    for k in got.COUNTERS + ("good_attempt", "globalInfectious"):
        assert getattr(got, k) == getattr(want, k), "%s %s: %r != %r" % (what, k, getattr(got, k), getattr(want, k))
    assert got.currentTime == want.currentTime, "%s currentTime %r != %r" % (what, got.currentTime, want.currentTime)
    assert np.array_equal(got.susceptible, want.susceptible), what + " susceptible"
    assert np.array_equal(got.infectious, want.infectious), what + " infectious"
    assert np.array_equal(got.lockdownON, want.lockdownON), what + " lockdownON"
    assert np.array_equal(got.contactDensity, want.contactDensity), what + " contactDensity"
    assert got.loc.states == want.loc.states and got.loc.populationsId == want.loc.populationsId, what + " lockdown log"
    assert got.loc.times == want.loc.times, what + " lockdown times"
    for k in ("idevents", "his", "hi2s", "nhis", "posRecombs"):   # forward recombination records (models.pxi:69-89)
        assert getattr(got.rec, k) == getattr(want.rec, k), "%s rec.%s" % (what, k)


def sparse_multievents(m, st):
    """Give a host model driven by the oracle the engine's multievent layout (rows with num > 0 only; MULTITYPE
    events carry [start, end) of their rows) derived from the oracle's dense reference layout."""
    if st.mev is None:
        return
    mv, n = st.mev, st.mev_ptr
    keep = np.nonzero(mv["num"][:n] > 0)[0]
    newpos = np.concatenate(([0], np.cumsum(mv["num"][:n] > 0)))
    ev = m.events
    multi = np.nonzero(ev.types[:ev.ptr] == 6)[0]
    ev.haplotypes[multi] = newpos[ev.haplotypes[multi]]
    ev.populations[multi] = newpos[ev.populations[multi]]
    m.multievents.ptr = 0
    m.multievents.extend(mv["times"][keep], **{k: mv[k][keep] for k in m.multievents.COLUMNS})
