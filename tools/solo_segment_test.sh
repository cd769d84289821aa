#!/usr/bin/env bash
# The single-trajectory kernel counts a segment of at most 2^30 iterations on 32-bit countdowns and does its 64-bit bookkeeping at
# the segment ends.  No test runs 10^9 iterations, so this builds the library with segments of 777 iterations
# (vgsim_amd/libvgx_seg.so) and runs tests/test_hip_solo.py against it: every case then crosses dozens of segment ends.
#   here:        bash tools/solo_segment_test.sh build
#   on the GPU:  bash tools/solo_segment_test.sh        (gpurun -- 'bash tools/solo_segment_test.sh')
set -euo pipefail
cd "$(dirname "$0")/.."
if [ "${1:-}" = build ]; then
  make -s -j8 -C vgsim_amd/csrc OBJDIR=build_seg OUT=../libvgx_seg.so EXTRA=-DVGX_SOLO_SEG=777
  exit 0
fi
VGX_LIBRARY=vgsim_amd/libvgx_seg.so python -m pytest tests/test_hip_solo.py -x -q
