"""Development aid: per-kernel totals from a rocprofv3 kernel_trace.csv for the dispatches after the last
vgx_direct_kernel (i.e. the last workload of tools/probe_tau.py)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
last_direct = max(int(r["End_Timestamp"]) for r in rows if r["Kernel_Name"] == "vgx_direct_kernel")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    if r["Kernel_Name"].startswith("vgx_tau") and int(r["Start_Timestamp"]) > last_direct:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-32s calls %3d total %9.2f ms max %8.2f ms" % (k, v[0], v[1], v[2]))
