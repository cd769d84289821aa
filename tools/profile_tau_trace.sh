#!/usr/bin/env bash
# Per-launch durations of the tau step kernels at config 4, in launch order (which tries cost what): bash tools/profile_tau_trace.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_tau/trace
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 ${PROBE:-tools/probe_tau_wall.py 20} > $O.log 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = {"vgx_tau_drift8_kernel": "D", "vgx_tau_apply_kernel": "A", "vgx_tau_sync8_kernel": "Y", "vgx_tau_decide_kernel": "d", "vgx_tau_finish_kernel": "F"}
out, t_prev = [], None
for r in rows:
    n = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = 0.0 if t_prev is None else (int(r["Start_Timestamp"]) - t_prev) / 1e3
    t_prev = int(r["End_Timestamp"])
    k = "L" if "listscan" in n else "S" if "scan_fast" in n else "E" if "events_kernel" in n else "f" if "front_kernel" in n else short.get(n)
    if k is None:
        k = "."
    out.append((k, dur, gap, n))
# the last call's steps only
idx = [i for i, o in enumerate(out) if o[0] == "F"]
NLAST = int(__import__("os").environ.get("NLAST", "20"))
start = idx[-NLAST - 1] + 1 if len(idx) > NLAST else 0
line = []
for k, dur, gap, n in out[start:]:
    if k in "SEAYDfL":
        line.append("%s%.0f" % (k, dur))
    if k == "F":
        print(" ".join(line)); line = []
tot = {}
gaps = 0.0
for k, dur, gap, n in out[start:]:
    tot[n] = tot.get(n, 0.0) + dur
    gaps += gap
print("gaps between kernels in the last call: %.1f us total" % gaps)
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:14]:
    print("%-62s %9.1f us" % (n[:62], v))
PY
find $O -name "*kernel_trace.csv" -delete
