"""Experiment: bot_tn_gemm_f32 on the two weight-gradient shapes of config-2 layer 0 against the fp32 library GEMMs.
    python tools/exp_tn.py"""
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
h = torch.randn(N, Fin, device=dev); dout2 = torch.randn(N, P2, device=dev); z = torch.randn(H, N, Fin, device=dev)
dW3 = torch.empty(H, D, Fin, device=dev)
def dw3_32():
    for i in range(H): torch.mm(dout2[:, i * D:(i + 1) * D].t(), z[i], out=dW3[i])
print("dWr = h^T dout2 [168,N]x[N,768]:      fp32 %.3f ms   tn_gemm %.3f ms" % (timed(lambda: torch.mm(h.t(), dout2)), timed(lambda: _C.tn_gemm(dout2, h, transpose_out=True))))
print("dW3 = dx_i^T z_i, 3 x [250,N]x[N,168]: fp32 %.3f ms   tn_gemm %.3f ms" % (
    timed(dw3_32), timed(lambda: _C.tn_gemm(dout2, z, out=dW3, batch=H, strides=(D, N * Fin, 0), n=N, kx=D, ky=Fin))))
