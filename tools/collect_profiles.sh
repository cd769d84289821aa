#!/usr/bin/env bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/prof_r02/ (copied to profiles/ afterwards by
# tools/make_profile_json.py).  rocprofv3 --stats and --pmc passes are separate runs (PMC passes with --kernel-trace only).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02
mkdir -p $O
run() { name=$1; shift; echo "== $name" ; "$@" > $O/$name.log 2>&1; echo "   rc=$?"; }
HEAD="--no-cpu-baseline --no-tau --no-extra --steps 3 --warmup 1"
LEGS=${LEGS:-"spread_occupancy spread_occupancy_fast tau_leap"}
if [ -z "${SKIP_HEAD:-}" ]; then
run headline_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline_stats -- python3 bench.py $HEAD
run headline_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/headline_fetch -- python3 bench.py $HEAD
run headline_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/headline_write -- python3 bench.py $HEAD
fi
for leg in $LEGS; do
  run ${leg}_stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/${leg}_stats -- python3 bench.py --only $leg
  run ${leg}_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${leg}_fetch -- python3 bench.py --only $leg
  run ${leg}_write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${leg}_write -- python3 bench.py --only $leg
done
echo done
