#!/usr/bin/env bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/prof_$TAG/ (copied to profiles/ afterwards by
# tools/make_profile_json.py).  rocprofv3 --stats and --pmc passes are separate runs (PMC passes with --kernel-trace only).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${TAG:-r04}
O=gpurun_out/prof_$TAG
mkdir -p $O
run() { name=$1; shift; echo "== $name" ; "$@" > $O/$name.log 2>&1; echo "   rc=$?"; find $O/$name -name "*kernel_trace.csv" -delete 2>/dev/null; }   # (gpurun copies back at most 64 MiB)
HEAD="--no-cpu-baseline --no-tau --no-extra --steps 3 --warmup 1"
LEGS=${LEGS:-"table3 single_trajectory tau_leap config3_general propensity_scan spread_occupancy tau_small"}
if [ -z "${SKIP_HEAD:-}" ]; then
run headline_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline_stats -- python3 bench.py $HEAD
run headline_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/headline_fetch -- python3 bench.py $HEAD
run headline_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/headline_write -- python3 bench.py $HEAD
fi
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
if [ -z "${SKIP_HEAD:-}" ]; then
run headline_sq rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/headline_sq -- python3 bench.py $HEAD
fi
for leg in $LEGS; do
  EXTRA=""
  if [ "$leg" = table3 ]; then EXTRA="--no-cpu-baseline --table3-cells ${TABLE3_CELLS:-2:0.001,10:0.001}"; fi
  if [ "$leg" = tau_leap ]; then EXTRA="--no-cpu-baseline --no-tau-warmup-start"; fi
  if [ "$leg" = tau_small ]; then EXTRA="--no-cpu-baseline"; fi
  run ${leg}_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/${leg}_stats -- python3 bench.py --only $leg $EXTRA
  if [ "$leg" = tau_small ]; then continue; fi      # (tens of thousands of launches: the stats pass only)
  run ${leg}_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${leg}_fetch -- python3 bench.py --only $leg $EXTRA
  run ${leg}_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${leg}_write -- python3 bench.py --only $leg $EXTRA
  run ${leg}_sq rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/${leg}_sq -- python3 bench.py --only $leg $EXTRA
done
du -sh $O
echo done
