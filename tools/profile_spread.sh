#!/usr/bin/env bash
# rocprofv3 SQ / memory passes of the spread-occupancy FAST probe; into gpurun_out/prof_sp/
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_sp
mkdir -p $O
SPEC=${SPEC:-12288:10000:fast:1:4}
run() { name=$1; shift; echo "== $name" ; "$@" > $O/$name.log 2>&1; echo "   rc=$?"; }
run sq rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 tools/probe_spread.py $SPEC
run vm rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/vm -- python3 tools/probe_spread.py $SPEC
run fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/probe_spread.py $SPEC
run write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/probe_spread.py $SPEC
rocprofv3 -L > $O/counters.txt 2>&1
python3 tools/sq_summary.py $O/sq vgx_quadf > $O/sq_summary.json 2>&1
cat $O/sq_summary.json
python3 - <<'PY'
import csv, glob, collections
for d in ("vm", "fetch", "write"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("gpurun_out/prof_sp/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "vgx_quadf_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(d, {k: (v, n[k]) for k, v in agg.items()})
PY
echo done
