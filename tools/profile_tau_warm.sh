#!/usr/bin/env bash
# Per-kernel time of bench.tau_warm_start's TIMED steps (natural occupancy): rocprofv3 kernel trace, the last 100 steps' launches summed.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_tau_warm
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/probe_tau_warm.py ${WARM:-6000} ${TIMED:-100} > $O.log 2>&1
tail -1 $O.log
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" ${TIMED:-100} <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("vgx_tau_finish")]
start = idx[-n - 1] + 1 if len(idx) > n else 0
tot, cnt = {}, {}
for r in rows[start:]:
    k = r["Kernel_Name"][:60]
    tot[k] = tot.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[k] = cnt.get(k, 0) + 1
span = (int(rows[-1]["End_Timestamp"]) - int(rows[start]["Start_Timestamp"])) / 1e3
print("last %d steps: %.1f us per step between first launch and last end; kernels %.1f us per step" % (n, span / n, sum(tot.values()) / n))
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:22]:
    print("%-62s %8.1f us/step  %6.2f launches/step  %8.1f us each" % (k, v / n, cnt[k] / n, v / cnt[k]))
PY
find $O -name "*kernel_trace.csv" -delete
