#!/usr/bin/env bash
# rocprofv3 kernel stats (and optionally PMC traffic) of the config-4 tau leg into gpurun_out/prof_tau/
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_tau
rm -rf $O; mkdir -p $O
run() { name=$1; shift; echo "== $name" ; "$@" > $O/$name.log 2>&1; echo "   rc=$?"; }
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --only tau_leap
if [ -n "${MEM:-}" ]; then
run fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --only tau_leap
run write rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --only tau_leap
fi
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-150
tail -c 600 $O/stats.log
echo done
