"""Per-kernel SQ counters of a rocprofv3 --pmc pass (instructions, wave cycles and where they went), per launch:
python tools/sq_summary.py <dir with */*counter_collection.csv> [kernel substring ...]"""
import collections, csv, glob, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
seen = set()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); launches[k] += 1
out = {}
for k, c in agg.items():
    n = launches[k]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    d = {"launches": n, "waves_per_launch": c.get("SQ_WAVES", 0) / n,
         "valu_per_wave": c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1),
         "salu_per_wave": c.get("SQ_INSTS_SALU", 0) / max(c.get("SQ_WAVES", 1), 1),
         "lds_per_wave": c.get("SQ_INSTS_LDS", 0) / max(c.get("SQ_WAVES", 1), 1)}
    if wc:
        d.update({"wait_any_frac": c.get("SQ_WAIT_ANY", 0) / wc, "wait_inst_any_frac": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                  "active_inst_any_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                  # SQ_*_CYCLES count quad-cycles (MI355X_MICROARCH.md): 4 shader cycles each
                  "wave_cycles_per_wave": 4.0 * wc / max(c.get("SQ_WAVES", 1), 1)})
    out[k] = d
print(json.dumps(out, indent=1))
