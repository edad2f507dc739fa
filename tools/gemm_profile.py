"""Per-launch table of the halves GEMMs of one workload step (HIP events around each launch, side stream off): shape key, time, TFLOP/s of fp16
MFMA work, GB/s of fp32-sized operand + result bytes, and which roofline bounds it.    python tools/gemm_profile.py products"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import _C, side, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
wl = workloads.build(name, "cuda", seed=0)
side.ENABLED = False
for _ in range(2):
    wl.step()
_C.PROFILE, _C.PROFILE_SKIP = [], ()
wl.step()
torch.cuda.synchronize()
recs = [r for r in _C.PROFILE if r[0] == "gemm_halves"]
_C.PROFILE = None
tot = 0.0
print(f"# {name}: {len(recs)} halves-GEMM launches in one step")
print("# kernel                                   m        n       k(x3)   ms      TFLOP/s  GB/s   bound  t_roof/t")
for r in recs:
    m, n, k3, bt = r[1]
    ms = r[2].elapsed_time(r[3])
    fl = 2.0 * m * n * k3 * bt
    by = 4.0 * bt * (m * (k3 / 3.0) + n * (k3 / 3.0) + m * n) if k3 > 1 else 0.0
    tm, th = fl / 2.5e15, by / 8e12
    tot += ms
    print(f"{r[4][-38:]:38s} {m:9d} {n:8d} {k3:7d} {ms:7.3f} {fl / ms / 1e9:8.0f} {by / ms / 1e6:6.0f}   {'hbm' if th > tm else 'mfma'}  {max(tm, th) * 1e3 / ms:5.2f}")
print(f"# total {tot:.2f} ms")
