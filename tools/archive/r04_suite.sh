#!/bin/bash
# the whole GPU suite as the driver runs it (-x), with durations; log into gpurun_out/r04
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu --durations=12 -s 2>&1 | grep -v "^\[Gloo\]\|^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/r04/gpu_tests.log
tail -40 gpurun_out/r04/gpu_tests.log
