cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_k100
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 tools/probe_k100.py one > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -- python3 tools/probe_k100.py one > $O/sq2.log 2>&1
python3 tools/sq_summary.py $O/sq vgx_quadg
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float)
for f in glob.glob("$O/sq2/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vgx_quadg" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
print(dict(agg))
PY
