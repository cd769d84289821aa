#!/usr/bin/env bash
# development loop of the single-trajectory kernel on the GPU box: parity tests, phase stamps, rates
set -uo pipefail
python -m pytest tests/test_hip_solo.py -x -q > gpurun_out/solo_t.log 2>&1; tail -4 gpurun_out/solo_t.log
VGX_LIBRARY=vgsim_amd/libvgx_prof.so python tools/profile_solo.py ${PK:-2} 0.001 200000 > gpurun_out/solo_prof.log 2>&1; cat gpurun_out/solo_prof.log
python tools/probe_solo.py solo > gpurun_out/solo_probe.log 2>&1; tail -1 gpurun_out/solo_probe.log
