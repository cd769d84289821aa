#!/usr/bin/env bash
# SQ counters of the on-device tau step loop (vgx_taus_kernel) on the small models of tools/probe_tau_small.py; into gpurun_out/prof_taus/
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_taus
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 tools/probe_tau_small.py > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES --output-format csv -d $O/vm -- python3 tools/probe_tau_small.py > $O/vm.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 tools/sq_summary.py $O/sq vgx_taus
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/prof_taus/vm/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vgx_taus" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for d, c in agg.items(): print(d, dict(c))
PY
grep "steps/s" $O/sq.log
