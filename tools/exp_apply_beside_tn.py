"""Round 6: what the BatchNorm-backward apply pass (HBM-bound, regenerates its Philox dropout mask) pays BESIDE the side stream's weight-gradient
product (MFMA + LDS-DMA), at the config-2 hidden-layer shapes: each kernel alone, then the pair (TN launched first on a second stream), for
dropout p = 0.75 / 0 and ReLU on / off.  HIP events on each stream.    python tools/exp_apply_beside_tn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import _C  # noqa: E402

dev = torch.device("cuda")
N, F, HD, B, P, K = 169343, 750, 750, 768, 1536, 768
g = torch.Generator(device=dev).manual_seed(0)
dy = torch.randn(N, 752, device=dev, generator=g)[:, :F]
x = torch.randn(N, 752, device=dev, generator=g)[:, :F]
mean, invstd = torch.zeros(F, device=dev), torch.ones(F, device=dev)
w, b = torch.ones(F, device=dev), torch.zeros(F, device=dev)
sg, sgx = torch.randn(F, device=dev, generator=g), torch.randn(F, device=dev, generator=g)
buf = torch.zeros(N, 2 * P, dtype=torch.float16, device=dev)
dxb = torch.empty(N, B, device=dev)
s1 = torch.tensor([64.0, 1 / 64.0], device=dev)
xh = (torch.randn(N, 2 * K, device=dev, generator=g)).half()
sx = torch.tensor([4.0, 0.25], device=dev)
side = _C.stream_create(dev)


def apply(p, relu):
    _C.bn_act_bwd_apply_halves(dy, x, mean, invstd, w, b, relu, p, 1234, sg, sgx, float(N), s1, buf[:, B:], HD, HD, out=dxb[:, :HD], h2_off=P)


def tn():
    return _C.gemm_halves3_tn(xh, buf, sx, s1, K, P, 750, P, x2_off=K, d2_off=P)


def timed(fn, stream=None, reps=10):
    st = stream or torch.cuda.current_stream()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for _ in range(3):
    apply(0.75, True), tn()
torch.cuda.synchronize()
t_tn = timed(tn)
print(f"TN [768 x 1536] over {N} rows alone: {t_tn * 1e3:.0f} us")
for p, relu in ((0.75, True), (0.0, True), (0.0, False)):
    t_a = timed(lambda: apply(p, relu))
    pa, pt, wall = [], [], []
    for _ in range(10):
        a0, a1, t0, t1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            t0.record()
            tn()
            t1.record()
        a0.record()
        apply(p, relu)
        a1.record()
        torch.cuda.synchronize()
        pa.append(a0.elapsed_time(a1)), pt.append(t0.elapsed_time(t1)), wall.append(max(t0.elapsed_time(t1), t0.elapsed_time(a1)))
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"apply p={p} relu={relu}: alone {t_a * 1e3:.0f} us; beside TN: apply {med(pa) * 1e3:.0f} us, TN {med(pt) * 1e3:.0f} us, both done after {med(wall) * 1e3:.0f} us "
          f"(serial {t_a * 1e3 + t_tn * 1e3:.0f})")
