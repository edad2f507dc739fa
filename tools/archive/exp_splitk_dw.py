"""Experiment: fp32 reduction noise and time of a Linear weight gradient gy^T x over N = 2.45 M rows (layer 1 of the S-products
GAT, real operands of a train step): one GEMM vs S partial GEMMs over row chunks (batched) summed afterwards, both against fp64.
    python tools/exp_splitk_dw.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import workloads, tuning
tuning.enable()
dev = torch.device("cuda", 0)
wl = workloads.build("products", dev, drop=False)
model, g, ds = wl.model.train(), wl.graph, wl.dataset
keep = {}
def fwd_hook(m, a, out):
    keep["x"] = a[1].detach()
    out.register_hook(lambda gr: keep.__setitem__("g", gr.detach().flatten(1)))
model.convs[1].register_forward_hook(fwd_hook)
pred = model(g)
workloads._loge(pred[ds.train_idx], ds.labels[ds.train_idx]).mean().backward()
x, gy = keep["x"], keep["g"].contiguous()
del pred
print("x", tuple(x.shape), "gy", tuple(gy.shape))
ref = gy.double().t() @ x.double()
scale = ref.abs().max()
def timed(f, k=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): out = f()
    torch.cuda.synchronize(); return out, (time.perf_counter() - t0) / k * 1e3
out, ms = timed(lambda: gy.t() @ x)
print("one GEMM              : err/max %.3e  %.2f ms" % (((out.double() - ref).abs().max() / scale).item(), ms))
n = x.shape[0]
for S in (8, 32, 128, 512):
    R = n // S
    def f():
        part = torch.bmm(gy[:S * R].view(S, R, -1).transpose(1, 2), x[:S * R].view(S, R, -1)).sum(0)
        if S * R < n:
            part = part + gy[S * R:].t() @ x[S * R:]
        return part
    out, ms = timed(f)
    print("S=%3d batched partials : err/max %.3e  %.2f ms" % (S, ((out.double() - ref).abs().max() / scale).item(), ms))
