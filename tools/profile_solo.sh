#!/usr/bin/env bash
# SQ counters of ONE trajectory on the latency kernel: instructions and cycles per event.  K M EVENTS from the environment.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_solo
mkdir -p $O
K=${K:-2}; M=${M:-0.001}; N=${N:-200000}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 tools/run_solo_one.py $K $M $N solo > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq2 -- python3 tools/run_solo_one.py $K $M $N solo > $O/sq2.log 2>&1
tail -1 $O/sq.log
python3 tools/sq_summary.py $O/sq vgx_solo
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$O/sq2/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vgx_solo" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(dict(agg))
PY
