#!/usr/bin/env bash
# The launches of bench.tau_warm_start's last steps in order with their durations (which tries cost what at natural occupancy).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_tau_warm_seq
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/probe_tau_warm.py ${WARM:-6000} ${TIMED:-100} > $O.log 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" ${NLAST:-6} <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("vgx_tau_finish")]
start = idx[-n - 1] + 1
short = {"vgx_tau_drift8_kernel": "D", "vgx_tau_apply_kernel": "A", "vgx_tau_sync8_kernel": "Y", "vgx_tau_decide_kernel": "d", "vgx_tau_finish_kernel": "F",
         "vgx_tau_colsum8_kernel": "C", "vgx_tau_listscan_kernel": "L", "vgx_tau_front_kernel": "f"}
line, t_prev, gaps = [], None, 0.0
for r in rows[start:]:
    k = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if t_prev is not None: gaps += (int(r["Start_Timestamp"]) - t_prev) / 1e3
    t_prev = int(r["End_Timestamp"])
    c = short.get(k) or ("E" if "events_kernel<2, false" in k or "events_kernel<1, false" in k else "e" if "events_kernel" in k else None)
    if c: line.append("%s%.0f" % (c, dur))
    if c == "F":
        print(" ".join(line) + "   | gaps %.0f" % gaps); line = []; gaps = 0.0
PY
find $O -name "*kernel_trace.csv" -delete
