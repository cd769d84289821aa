#!/usr/bin/env bash
# Builds libvgx.so and says BUILD_OK only if the library is newer than every source (a failed build must not go unnoticed before a
# GPU run).  PROF=1 also builds the diagnostic library with in-kernel phase stamps (vgsim_amd/libvgx_prof.so).
set -euo pipefail
cd "$(dirname "$0")/.."
make -j8 -s -C vgsim_amd/csrc 2>&1 | grep -v "argument unused" || true
if [ -n "${PROF:-}" ]; then make -j8 -s -C vgsim_amd/csrc prof 2>&1 | grep -v "argument unused" || true; fi
for f in vgsim_amd/csrc/*.hip vgsim_amd/csrc/*.h vgsim_amd/csrc/*.cpp include/vgx.h; do
  test vgsim_amd/libvgx.so -nt "$f" || { echo "BUILD FAILED (libvgx.so older than $f)"; exit 1; }
done
echo BUILD_OK
