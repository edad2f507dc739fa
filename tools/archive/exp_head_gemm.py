"""Experiment: the aggregate-first layer's per-head projections (3 x [N,168]x[168,250], config 2 layer 0) as three 2-D GEMMs on
column slices (what fused.py does) vs one strided-batched GEMM (baddbmm / bmm on permuted views)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import tuning
tuning.enable()
N, H, D, Fin, P2 = 169343, 3, 250, 168, 768
dev = "cuda"
z = torch.randn(H, N, Fin, device=dev)
W = torch.randn(H * D, Fin, device=dev)
Wh = W.view(H, D, Fin)
out2 = torch.randn(N, P2, device=dev)
dout2 = torch.randn(N, P2, device=dev)

def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3

def fwd_loop():
    for i in range(H):
        out2[:, i * D:(i + 1) * D].addmm_(z[i], Wh[i].t())
def fwd_bmm():
    ov = out2[:, :H * D].unflatten(1, (H, D)).transpose(0, 1)   # [H, N, D] strides (D, P2, 1)
    ov.baddbmm_(z, Wh.transpose(1, 2))
dz = torch.empty(H, N, Fin, device=dev)
def bwd_dz_loop():
    for i in range(H):
        torch.mm(dout2[:, i * D:(i + 1) * D], Wh[i], out=dz[i])
def bwd_dz_bmm():
    dv = dout2[:, :H * D].unflatten(1, (H, D)).transpose(0, 1)
    torch.bmm(dv, Wh, out=dz)
dW3 = torch.empty(H, D, Fin, device=dev)
def bwd_dw_loop():
    for i in range(H):
        torch.mm(dout2[:, i * D:(i + 1) * D].t(), z[i], out=dW3[i])
def bwd_dw_bmm():
    dv = dout2[:, :H * D].unflatten(1, (H, D)).transpose(0, 1)
    torch.bmm(dv.transpose(1, 2), z, out=dW3)
a = out2.clone(); fwd_loop(); r1 = out2.clone(); out2.copy_(a); fwd_bmm(); print("fwd equal", torch.allclose(r1, out2, atol=1e-3))
for name, f in (("fwd loop", fwd_loop), ("fwd baddbmm", fwd_bmm), ("dz loop", bwd_dz_loop), ("dz bmm", bwd_dz_bmm), ("dW loop", bwd_dw_loop), ("dW bmm", bwd_dw_bmm)):
    print("%-12s %.3f ms" % (name, t(f)))
