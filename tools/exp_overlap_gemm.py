#!/usr/bin/env python3
"""Can a weight-gradient GEMM (MFMA-bound, off the critical path of the backward) run in the shadow of the memory-bound kernels of
the next layer's backward?  Times, at config-2 sizes: the layer-1 dW halves GEMM alone, a chain of memory-bound kernels alone
(BatchNorm backward reduce + apply, the fused sparse backward sweep), and both on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import _C, gemm, synth
dev = "cuda"
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
g = ds.graph.to(dev); g.create_formats_()
N, E, H, D = g.number_of_nodes(), g.number_of_edges(), 3, 250
x = torch.randn(N, 750, device=dev); dout = torch.randn(N, 1536, device=dev) * 1e-3
xh, dh = gemm.split(x, 0), gemm.split(dout, 0)
def gemm_dw():
    return gemm.tn(xh, dh)
dy = torch.randn(N, 750, device=dev); xb = torch.randn(N, 752, device=dev)[:, :750]
mean, invstd = torch.zeros(750, device=dev), torch.ones(750, device=dev)
big = torch.randn(N, 1536, device=dev); a = torch.rand(E, H, device=dev)
def chain():
    sg, sgx = _C.bn_act_bwd_reduce(dy, xb, mean, invstd, None, None, True, 0.75, 123)
    _C.bn_act_bwd_apply(dy, xb, mean, invstd, None, None, True, 0.75, 123, sg, sgx, float(N), out=big[:, 752:1502])
    _C.spmm_dot(g.csr, big[:, 752:1502].unflatten(1, (H, D)), a, g.csr2csc, big[:, :750].unflatten(1, (H, D)))
def timed(f, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
side = torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        gemm_dw()
    chain()
    cur.wait_stream(side)
tg, tc, tb = timed(gemm_dw), timed(chain), timed(both)
print(f"dW GEMM alone {tg:.3f} ms   memory-bound chain alone {tc:.3f} ms   sum {tg + tc:.3f}   both on two streams {tb:.3f} ms")
