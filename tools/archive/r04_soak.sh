#!/bin/bash
# soak: the GPU suite twice more and a 100-step bench line (flakiness check of the round's final state)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
for i in 1 2; do
  python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^\[Gloo\]\|^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -3 | tee -a gpurun_out/r04/soak.txt
done
python bench.py --steps 100 --warmup 10 --cpu-baseline off 2>/dev/null | tail -1 | cut -c1-400 | tee -a gpurun_out/r04/soak.txt
