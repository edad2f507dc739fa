"""Measured HBM peak (torch allocator) of build + two steps of every workload on one GPU, beside bot_amd.workloads.hbm_budget's estimate."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import workloads  # noqa: E402

out = {}
for name in sys.argv[1:] or workloads.NAMES:
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    wl = workloads.build(name, "cuda", seed=0)
    b = torch.cuda.max_memory_allocated()
    for _ in range(2):
        wl.step()
    torch.cuda.synchronize()
    p = torch.cuda.max_memory_allocated()
    est = workloads.hbm_budget(name)
    out[name] = {"n": wl.n_nodes, "E": wl.n_edges, "build_peak_GiB": round(b / 2**30, 2), "step_peak_GiB": round(p / 2**30, 2),
                 "estimate_GiB": {k: round(v / 2**30, 2) for k, v in est.items()}}
    print(name, json.dumps(out[name]), flush=True)
    del wl
