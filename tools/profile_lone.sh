#!/usr/bin/env bash
# SQ counters of config-3 trajectories on the latency kernel for large haplotype spaces: instructions and cycles per event.
# N (events), R (replicates), T (trajectory points) from the environment.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_lone
mkdir -p $O
N=${N:-100000}; R=${R:-1}; T=${T:-0}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 tools/run_lone_one.py $N lone $R $T > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/sq2 -- python3 tools/run_lone_one.py $N lone $R $T > $O/sq2.log 2>&1
tail -1 $O/sq.log
python3 tools/sq_summary.py $O/sq vgx_lone
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$O/sq2/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vgx_lone" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(dict(agg))
PY
