#!/usr/bin/env python3
"""torch.profiler table of one headline train step (after warm-up): which `aten::` ops still launch kernels, with input shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import workloads, tuning
tuning.enable()
wl = workloads.build("arxiv", "cuda", scale=1.0)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:40]:
    print(f"{e.key:32s} calls/step {e.count / 3:5.1f}  self device us/step {e.self_device_time_total / 3:8.1f}  shapes {str(e.input_shapes)[:110]}")
