"""Experiment: bot_skinny_gemm_f32 on the GEMMs of config-2 layer 0 against the fp32 library GEMMs (TunableOp selections loaded).
    python tools/exp_skinny.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import _C, tuning
tuning.enable()
dev = torch.device("cuda", 0)
N, H, Fin, D, P2 = 169343, 3, 168, 250, 768
def timed(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
h = torch.randn(N, Fin, device=dev); Wr = torch.randn(Fin, P2, device=dev) * 0.1
z = torch.randn(H, N, Fin, device=dev); W = torch.randn(H, D, Fin, device=dev) * 0.1
out2 = torch.empty(N, P2, device=dev); dz = torch.empty(H, N, Fin, device=dev)
dx = torch.randn(N, P2, device=dev)
print("h @ Wr  [N,168]x[168,768]:   fp32 %.3f ms   skinny %.3f ms" % (
    timed(lambda: torch.mm(h, Wr, out=out2)), timed(lambda: _C.skinny_gemm(h, Wr, b_is_kn=True, out=out2))))
def fwd32():
    for i in range(H): out2[:, i * D:(i + 1) * D].addmm_(z[i], W[i].t())
print("per-head fwd (+= into out2): fp32 %.3f ms   skinny %.3f ms" % (
    timed(fwd32), timed(lambda: _C.skinny_gemm(z, W, b_is_kn=False, out=out2, accumulate=True, batch=H, strides=(N * Fin, D * Fin, D), m=N, n=D, k=Fin))))
def dz32():
    for i in range(H): torch.mm(dx[:, i * D:(i + 1) * D], W[i], out=dz[i])
print("per-head dz:                 fp32 %.3f ms   skinny %.3f ms" % (
    timed(dz32), timed(lambda: _C.skinny_gemm(dx, W, b_is_kn=True, out=dz, batch=H, strides=(D, D * Fin, N * Fin), m=N, n=Fin, k=D))))
# what a fused [z_i | h] product could save: the per-head product without the read-modify-write, h @ Wr for the score columns only
print("per-head fwd, plain stores:   skinny %.3f ms" % timed(lambda: _C.skinny_gemm(z, W, b_is_kn=False, out=out2, accumulate=False, batch=H, strides=(N * Fin, D * Fin, D), m=N, n=D, k=Fin)))
Ws = Wr[:, 750:].contiguous(); outs = torch.empty(N, 18, device=dev)
print("h @ Wr[:, 750:] (18 columns): skinny %.3f ms   fp32 %.3f ms" % (timed(lambda: _C.skinny_gemm(h, Ws, b_is_kn=True, out=outs)), timed(lambda: torch.mm(h, Ws, out=outs))))
