"""Experiment: fused BatchNorm kernels at F = 750 (8-byte lanes) vs F = 752 (16-byte lanes), N = 169 343."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import _C
n = 169343
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for F in (750, 752):
    x = torch.randn(n, F, device="cuda"); dy = torch.randn(n, F, device="cuda")
    w = torch.rand(F, device="cuda") + 0.5; b = torch.randn(F, device="cuda")
    mean, m2 = _C.colstats(x); invstd = torch.rsqrt(m2 / n + 1e-5)
    sg, sgx = _C.bn_act_bwd_reduce(dy, x, mean, invstd, w, b, True, 0.75, 123)
    r = {"colstats": t(lambda: _C.colstats(x)),
         "fwd": t(lambda: _C.bn_act_fwd(x, mean, invstd, w, b, True, 0.75, 123)),
         "bwd_reduce": t(lambda: _C.bn_act_bwd_reduce(dy, x, mean, invstd, w, b, True, 0.75, 123)),
         "bwd_apply": t(lambda: _C.bn_act_bwd_apply(dy, x, mean, invstd, w, b, True, 0.75, 123, sg, sgx, float(n)))}
    gb = n * F * 4 / 1e9
    print(F, {k: round(v, 4) for k, v in r.items()}, "TB/s:", {k: round(gb * m / v, 2) for (k, v), m in zip(r.items(), (1, 2, 2, 3))})
