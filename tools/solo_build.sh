#!/usr/bin/env bash
# build both libraries; fail loudly
set -euo pipefail
cd "$(dirname "$0")/.."
make -j8 -s -C vgsim_amd/csrc 2>&1 | grep -v "argument unused" || true
make -j8 -s -C vgsim_amd/csrc prof 2>&1 | grep -v "argument unused" || true
test vgsim_amd/libvgx.so -nt vgsim_amd/csrc/vgx_solo.hip && test vgsim_amd/libvgx_prof.so -nt vgsim_amd/csrc/vgx_solo.hip && echo BUILD_OK
