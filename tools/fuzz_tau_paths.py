"""Development aid: the accelerated paths of a tau try (front pass, lists of occupied compartments, front pass alone) against the plain
ones, and the byte drift pass against the two-pass form, on random filled models — same leaps, same states.  python tools/fuzz_tau_paths.py [cases] [seed]
Each case runs in fresh subprocesses (the switches are read from the environment when a call sets its kernels up)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, hashlib
import test_hip_tau as T, helpers
cfg = json.loads(sys.argv[1])
rng = np.random.default_rng(cfg["seed"])
def fill(r, shape):
    a = np.zeros(shape, dtype=np.int64)
    n = max(1, int(a.size * cfg["occ"]))
    idx = rng.choice(a.size, size=n, replace=False)
    a.reshape(-1)[idx] = rng.choice(cfg["vals"], size=n)
    return a
s = T._filled(cfg["sites"], cfg["P"], cfg["S"], cfg["seed"], fill, cfg["mig"], classes=cfg["classes"])
with helpers.quiet():
    s.simulate(cfg["steps"], sample_size=10 ** 12, method="tau", record_multievents=False)
m = s.simulation
h = hashlib.sha256()
for x in (m.infectious, m.susceptible, m.events.times[:m.events.ptr]):
    h.update(np.ascontiguousarray(x).tobytes())
hs = hashlib.sha256()
for x in (m.infectious, m.susceptible):
    hs.update(np.ascontiguousarray(x).tobytes())
print(json.dumps({"ptr": int(m.events.ptr), "sha": h.hexdigest(), "state": hs.hexdigest(), "b": int(m.bCounter), "d": int(m.dCounter), "mut": int(m.mCounter)}))
''' % (ROOT, ROOT)


def run(cfg, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", CHILD, json.dumps(cfg)], env=env, capture_output=True, text=True)
    if out.returncode != 0:      # (upstream's dead end: every path must end there, whatever its message)
        return {"error": "halving" if "halving" in out.stderr else out.stderr[-400:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    import numpy as np
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for i in range(n):
        cfg = {"seed": int(rng.integers(1, 10 ** 6)), "sites": int(rng.integers(7, 11)), "P": int(rng.integers(1, 5)), "S": int(rng.integers(1, 3)),
               "mig": bool(rng.integers(0, 2)), "classes": int(os.environ.get("FUZZ_CLASSES") or rng.choice([1, 3])), "steps": int(rng.integers(3, 9)),   # (FUZZ_CLASSES=1: the byte drift pass's models only)
               "occ": float(rng.choice([0.002, 0.01, 0.02, 0.2, 1.0])),
               "vals": [int(v) for v in rng.choice([1, 1, 2, 3, 5, 9, 40, 200, 254, 255, 256, 900, 30000], size=5)]}
        if cfg["P"] == 1:
            cfg["mig"] = False
        ref = run(cfg, {"VGX_TAU_NO_FRONT": "1", "VGX_TAU_NO_OCCLIST": "1"})
        res = {"all": run(cfg, {}), "no lists": run(cfg, {"VGX_TAU_NO_OCCLIST": "1"}), "front in the try": run(cfg, {"VGX_TAU_NO_FRONT_ALONE": "1"})}
        ok = all(r == ref for r in res.values())     # (a model that runs into upstream's dead end must do so on every path)
        # the byte drift pass (its single-precision screen included) against the two-pass form: leap lengths agree to the last bits
        # only, so everything but the times
        two = run(cfg, {"VGX_TAU_NO_BYTE_DRIFT": "1"})
        res["two-pass drift"] = two
        ok = ok and {k: v for k, v in two.items() if k != "sha"} == {k: v for k, v in ref.items() if k != "sha"}
        # the drift pass that takes only the occupied dwords of a sparse state against the dense one: the same, up to the susceptible
        # compartments' sums (another order)
        dense = run(cfg, {"VGX_TAU_DENSE_DRIFT": "1"})
        res["dense drift"] = dense
        ok = ok and {k: v for k, v in dense.items() if k != "sha"} == {k: v for k, v in res["all"].items() if k != "sha"}
        bad += not ok
        print("%s %s -> %s" % ("ok  " if ok else "DIFF", json.dumps(cfg), json.dumps(ref if ok else {"plain": ref, **res})), flush=True)
    print("%d of %d cases differ" % (bad, n))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
