#!/usr/bin/env python3
"""Would the layer-0 small-K products of config 2 pay on the fp16-halves GEMMs if their operands' halves came for free (written by the
producers)?  Times the hipBLASLt halves GEMMs of the five products on random halves buffers of the right shapes (best of the 16
heuristic candidates, BOT_GEMM_TUNE=1) next to what runs today (skinny_gemm / tn_gemm / library fp32)."""
import os, sys, time
os.environ["BOT_GEMM_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import _C
dev = "cuda"
N, H, Fin, D = 169343, 3, 168, 250
def timed(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
hf = lambda *s: (torch.randn(*s, device=dev) * 0.1).half()
one = torch.ones(1, device=dev)
KF, KD = 192, 256                      # piece widths: Fin -> 192, D -> 256
hh = hf(N, 3 * KF)                     # h halves [h1|h1|h2]
Wr = hf(768, 3 * KF)                   # right operand [n=768, K]
out2 = torch.empty(N, 768, device=dev)
t1 = timed(lambda: _C.gemm_halves(hh, Wr, one, trans_b=True, out=out2))
zh = hf(H, N, 3 * KF)
Wh = hf(H, D, 3 * KF)
t2 = timed(lambda: _C.gemm_halves(zh, Wh, one, trans_b=True, out=out2, batch=H, strides=(N * 3 * KF, D * 3 * KF, D), m=N, n=D, k=3 * KF, beta=1.0, ldc=768))
dxh = hf(N, H * 3 * KD)                # head-major [h1_i|h1_i|h2_i] pieces of 256
Wt = hf(H, Fin, 3 * KD)                # per head [n=168, K=768]
dz = torch.empty(H, N, Fin, device=dev)
t3 = timed(lambda: _C.gemm_halves(dxh, Wt, one, trans_b=True, out=dz, batch=H, strides=(3 * KD, Fin * 3 * KD, N * Fin), m=N, n=Fin, k=3 * KD))
# weight gradients: per head, chunked over rows (S chunks as batch entries): x1^T [d1|d2] and x2^T d1
R = 8192; S = N // R
def dw(xh_, ldx, x_off1, x_off2, kx, dh_, ldd, d_off, kd):
    # x1^T[d1|d2]: m=kx, n=2*kd, k=R, batch S ; x2^T d1: m=kx, n=kd
    a = _C.gemm_halves(xh_[:, x_off1:], dh_[:, d_off + kd:], one.expand(2 * kd).contiguous(), trans_a=True, m=kx, n=2 * kd, k=R, batch=S, strides=(R * ldx, R * ldd, 0))
    b = _C.gemm_halves(xh_[:, x_off2:], dh_[:, d_off:], one.expand(kd).contiguous(), trans_a=True, m=kx, n=kd, k=R, batch=S, strides=(R * ldx, R * ldd, 0))
    return a.sum(0), b.sum(0)
zh2 = hf(N, H * 3 * KF)                # z halves row-major per node for the reduction over rows: [N, H*3*192]
def dW3():
    for i in range(H):
        dw(dxh, H * 3 * KD, i * 3 * KD, i * 3 * KD + 2 * KD, D, zh2, H * 3 * KF, i * 3 * KF, KF)
def dWres():
    for i in range(H):
        dw(dxh, H * 3 * KD, i * 3 * KD, i * 3 * KD + 2 * KD, D, hh, 3 * KF, 0, KF)
t4, t5 = timed(dW3, 10), timed(dWres, 10)
print(f"halves GEMMs: out2 {t1:.3f}  heads(+=) {t2:.3f}  dz {t3:.3f}  dW3 {t4:.3f}  dWres {t5:.3f}  total {t1 + t2 + t3 + t4 + t5:.3f} ms")
print("today: skinny 0.305 + 0.420 + 0.339, tn_gemm 0.56 + 0.11, fp32 dWr 0.49 = 2.23 ms (profiles/r03_skinny_ablation.txt, kernel stats)")
