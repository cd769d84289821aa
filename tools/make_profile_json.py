"""Turns gpurun_out/prof_r02/ (tools/collect_profiles.sh) into the committed artefacts under profiles/: kernel-stats CSVs,
the raw PMC rows, and the JSON files bench.py reads for its `traffic` fields."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02")
DST = os.path.join(ROOT, "profiles")
TAG = "r02"


def one(pattern):
    f = glob.glob(os.path.join(SRC, pattern))
    assert f, pattern
    return f[0]


def pmc(leg, counter):
    """counter KiB and dispatch count per kernel name"""
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(one("%s_%s/*/*counter_collection.csv" % (leg, "fetch" if counter == "FETCH_SIZE" else "write")))):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg


CORR = ("gfx950: FETCH_SIZE counts 64 B per 128-B read request (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE taken as is; "
        "both in KiB; separate rocprofv3 --pmc passes with --kernel-trace only")
for leg in ("headline", "spread_occupancy", "spread_occupancy_fast", "tau_leap"):
    shutil.copy(one("%s_stats/*/*kernel_stats.csv" % leg), os.path.join(DST, "%s_%s_kernel_stats.csv" % (TAG, leg)))
    for c in ("fetch", "write"):
        shutil.copy(one("%s_%s/*/*counter_collection.csv" % (leg, c)), os.path.join(DST, "%s_%s_pmc_%s.csv" % (TAG, leg, c.upper())))
    # the JSON line each profiled run printed
    log = open(os.path.join(SRC, "%s_stats.log" % leg)).read().splitlines()
    js = [ln for ln in log if ln.startswith("{")]
    if js:
        open(os.path.join(DST, "%s_%s_under_rocprof.json" % (TAG, leg)), "w").write(js[-1] + "\n")

direct = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, round 2; raw rows in profiles/r02_*_pmc_*.csv; round-1 values in git history",
          "correction": CORR}
for leg, kernel, cmd, cfg in (
        ("headline", "vgx_quad_kernel", "python3 bench.py --no-cpu-baseline --no-tau --no-extra --steps 3 --warmup 1",
         {"replicates_per_gpu": 16384, "events_per_replicate": 100000, "trajectory_points": 1001}),
        ("spread_occupancy", "vgx_quad_long_kernel", "python3 bench.py --only spread_occupancy",
         {"replicates_per_gpu": 8192, "events_per_replicate": 2500, "occupied": 4096, "mode": "exact"}),
        ("spread_occupancy_fast", "vgx_direct_fast_kernel_p64s1", "python3 bench.py --only spread_occupancy_fast",
         {"replicates_per_gpu": 4096, "events_per_replicate": 20000, "occupied": 4096, "mode": "fast"})):
    f, w = pmc(leg, "FETCH_SIZE")[kernel], pmc(leg, "WRITE_SIZE")[kernel]
    assert f[1] == w[1] and f[1] > 0
    direct[leg] = {"command": cmd, "config": cfg, "kernel": kernel, "launches": f[1], "FETCH_SIZE_KiB_per_launch": f[0] / f[1],
                   "WRITE_SIZE_KiB_per_launch": w[0] / w[1], "hbm_bytes_per_launch": (2.0 * f[0] / f[1] + w[0] / w[1]) * 1024}
json.dump(direct, open(os.path.join(DST, "pmc_direct_c3.json"), "w"), indent=1)

steps = 20
fk, wk = pmc("tau_leap", "FETCH_SIZE"), pmc("tau_leap", "WRITE_SIZE")
kern = {}
tot = 0.0
for k in sorted(set(fk) | set(wk)):
    if "vgx_tau" not in k:   # (template instances are listed as "void vgx_tau_...<...>(VgxTauArgs)")
        continue
    rb, wb = 2.0 * fk[k][0] * 1024 / steps, wk[k][0] * 1024 / steps
    kern[k] = {"launches": fk[k][1], "read_bytes_per_step": rb, "write_bytes_per_step": wb}
    tot += rb + wb
old = json.load(open(os.path.join(DST, "pmc_tau_c4.json")))
hist = old.get("history", {})
if isinstance(hist, dict):
    hist = dict(hist)
    prev = old.get("hbm_bytes_per_step")
    if prev and abs(prev - tot) > 1e6 and prev not in hist.values():
        hist["before this collection (%s)" % TAG] = prev   # rename by hand to what that kernel set was
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, csv), command: python3 bench.py --only tau_leap "
                     "(config 4, 20 steps), round 2; raw rows in profiles/r02_tau_leap_pmc_*.csv",
           "correction": CORR, "config": {"steps": steps, "per_cell": 3}, "steps": steps, "kernels": kern,
           "hbm_bytes_per_step": tot, "history": hist}, open(os.path.join(DST, "pmc_tau_c4.json"), "w"), indent=1)
print("headline %.3g B/launch, spread %.3g, spread_fast %.3g, tau %.3g B/step" % (
    direct["headline"]["hbm_bytes_per_launch"], direct["spread_occupancy"]["hbm_bytes_per_launch"],
    direct["spread_occupancy_fast"]["hbm_bytes_per_launch"], tot))
for k, v in sorted(kern.items(), key=lambda kv: -(kv[1]["read_bytes_per_step"] + kv[1]["write_bytes_per_step"]))[:8]:
    print("  %-30s %6.2f GB/step" % (k, (v["read_bytes_per_step"] + v["write_bytes_per_step"]) / 1e9))
