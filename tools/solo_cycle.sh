#!/usr/bin/env bash
# development loop of the single-trajectory kernel on the GPU box: parity tests, phase stamps, rates
set -uo pipefail
python -m pytest tests/test_hip_solo.py -x -q > gpurun_out/solo_t.log 2>&1; tail -4 gpurun_out/solo_t.log
VGX_LIBRARY=vgsim_amd/libvgx_prof.so python tools/profile_solo.py 2 0.001 200000
VGX_LIBRARY=vgsim_amd/libvgx_prof.so python tools/profile_solo.py 0 0 200000 | head -1
python tools/probe_solo.py solo 2>&1 | tail -1
