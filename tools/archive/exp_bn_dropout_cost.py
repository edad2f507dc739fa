"""What the regenerated Philox dropout mask costs the fused BatchNorm kernels at [169 343, 752]: p = 0 vs p = 0.75."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import _C
n, F = 169343, 752
def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
x = torch.randn(n, F, device="cuda"); dy = torch.randn(n, F, device="cuda")
w = torch.rand(F, device="cuda") + 0.5; b = torch.randn(F, device="cuda")
mean, m2 = _C.colstats(x); invstd = torch.rsqrt(m2 / n + 1e-5)
hscale = torch.tensor([16.0, 1 / 16.0], device="cuda")
for p in (0.0, 0.75):
    sg, sgx = _C.bn_act_bwd_reduce(dy, x, mean, invstd, w, b, True, p, 123)
    r = {"fwd": t(lambda: _C.bn_act_fwd(x, mean, invstd, w, b, True, p, 123)),
         "fwd_halves_only": t(lambda: _C.bn_act_fwd(x, mean, invstd, w, b, True, p, 123, halves=(hscale, 768), want_y=False)),
         "bwd_reduce": t(lambda: _C.bn_act_bwd_reduce(dy, x, mean, invstd, w, b, True, p, 123)),
         "bwd_apply": t(lambda: _C.bn_act_bwd_apply(dy, x, mean, invstd, w, b, True, p, 123, sg, sgx, float(n)))}
    print("p =", p, {k: round(v, 4) for k, v in r.items()})
