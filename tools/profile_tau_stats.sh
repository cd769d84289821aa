cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_tau
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tau/stats -- python3 tools/probe_tau_wall.py 20 > gpurun_out/prof_tau/stats.log 2>&1
f=$(find gpurun_out/prof_tau/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-60s calls %6s  total %9.3f ms  avg %8.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
find gpurun_out/prof_tau/stats -name "*kernel_trace.csv" -delete
