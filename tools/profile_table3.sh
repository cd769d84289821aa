#!/usr/bin/env bash
# rocprofv3 passes of the table3 K=10 ensemble (general row kernel): stats, SQ counters, FETCH/WRITE; into gpurun_out/prof_t3/
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_t3
mkdir -p $O
ARGS="--only table3 --no-cpu-baseline --table3-cells ${CELLS:-10:0.001}"
run() { name=$1; shift; echo "== $name" ; "$@" > $O/$name.log 2>&1; echo "   rc=$?"; }
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $ARGS
run sq rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 bench.py $ARGS
if [ -z "${SKIP_MEM:-}" ]; then
run fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py $ARGS
run write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py $ARGS
fi
python3 tools/sq_summary.py $O/sq vgx_quadg > $O/sq_summary.json 2>&1
cat $O/sq_summary.json
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -5
echo done
