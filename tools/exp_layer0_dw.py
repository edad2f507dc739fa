"""Experiment: the two weight gradients of config-2 layer 0 (reductions over N = 169 343 rows with small outputs) as fp32 GEMMs:
one GEMM each (what runs now, TunableOp selections loaded) vs row-chunked batched products summed afterwards.
    python tools/exp_layer0_dw.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import tuning
tuning.enable()
dev = torch.device("cuda", 0)
N, H, Fin, D, P2 = 169343, 3, 168, 250, 768
def timed(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
z = torch.randn(H, N, Fin, device=dev); dout2 = torch.randn(N, P2, device=dev); h = torch.randn(N, Fin, device=dev)
dx = dout2[:, :H * D]
print("dW3 now: 3 x [250,N]x[N,168] (strided dx)   %.3f ms" % timed(lambda: [torch.mm(dx[:, i * D:(i + 1) * D].t(), z[i]) for i in range(H)]))
print("dWr now: [168,N]x[N,768]                    %.3f ms" % timed(lambda: torch.mm(h.t(), dout2)))
for S in (8, 16, 32, 64):
    R = N // S
    def dw3():
        out = []
        for i in range(H):
            a = dx[:S * R, i * D:(i + 1) * D].reshape(S, R, D).transpose(1, 2)        # copies (strided columns)
            out.append(torch.bmm(a, z[i, :S * R].view(S, R, Fin)).sum(0))
        return out
    def dwr():
        return torch.bmm(h[:S * R].view(S, R, Fin).transpose(1, 2), dout2[:S * R].view(S, R, P2)).sum(0)
    print("S=%2d chunks: dW3 %.3f ms   dWr %.3f ms" % (S, timed(dw3), timed(dwr)))
# dW3 for all heads at once from the [N, 768] gradient: [768, N] x [N, 3*168] is 3x the flops but ONE product
zc = z.permute(1, 0, 2).reshape(N, H * Fin).contiguous()
print("dW3 as one dense [768,N]x[N,504] (3x flops) %.3f ms" % timed(lambda: torch.mm(dout2.t(), zc)))
for S in (16, 32):
    R = N // S
    print("  chunked S=%d                              %.3f ms" % (S, timed(lambda: torch.bmm(dout2[:S * R].view(S, R, P2).transpose(1, 2), zc[:S * R].view(S, R, H * Fin)).sum(0))))
