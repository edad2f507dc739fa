import torch, sys
sys.path.insert(0, "/root/repo")
from bot_amd import workloads, gemm
for name, scale in (("products", 0.02), ("arxiv", 1.0)):
    wl = workloads.build(name, torch.device("cuda", 0), scale=scale)
    for _ in range(2): wl.step()
    print(name, gemm.STATS); gemm.STATS.update(stashed=0, taken=0, split=0)
