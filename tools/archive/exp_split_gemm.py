"""Experiment: fp32 GEMMs of the config-2 layer-1 shapes computed from fp16 halves on the fp16 MFMA path.
x = s (h1 + h2) with h1 = fp16(x/s), h2 = fp16(x/s - h1)  (22 of the 24 mantissa bits; s a power of two putting max|x| at 2^14),
x w = s_x s_w (h1 g1 + h1 g2 + h2 g1) as ONE fp16 GEMM over the concatenated reduction axis with fp32 accumulation/output.
Reports time and error against an fp64 product next to the native fp32 GEMM.     python tools/exp_split_gemm.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
N, K, P = 169343, 750, 1506

def timed(f, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): out = f()
    torch.cuda.synchronize(); return out, (time.perf_counter() - t0) / k * 1e3

def scale_of(x):
    return 2.0 ** (14 - torch.ceil(torch.log2(x.abs().max())).item())

def split(x, dt=torch.float16, pieces=2):
    s = scale_of(x) if dt == torch.float16 else 1.0
    r = x * s
    out = []
    for _ in range(pieces):
        h = r.to(dt)
        out.append(h)
        r = r - h.float()
    return out, s

def report(name, got, ref, ms, flops):
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{name:44s} {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s(fp32-equivalent)  max err / max |ref| = {err:.2e}")

x = torch.relu(torch.randn(N, K, device=dev, generator=gen)) * (torch.rand(N, K, device=dev, generator=gen) > 0.75) * 4
w = torch.randn(P, K, device=dev, generator=gen) * 0.05
dy = torch.randn(N, P, device=dev, generator=gen) * 1e-7
for tag, a, b in (("fwd  x[N,K] w^T[K,P]", x, w.t()), ("dX   dy[N,P] w[P,K]", dy, w), ("dW   dy^T[P,N] x[N,K]", dy.t(), x)):
    flops = 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
    ref = a.double() @ b.double()
    out, ms = timed(lambda: a @ b)
    report(tag + "  native fp32", out, ref, ms, flops)
    # fp16 halves, 3 terms, one GEMM over the concatenated reduction axis
    (a1, a2), sa = split(a.contiguous())
    (b1, b2), sb = split(b.contiguous())
    A = torch.cat([a1, a1, a2], 1).contiguous()
    B = torch.cat([b1, b2, b1], 0).contiguous()
    out, ms = timed(lambda: torch.mm(A, B, out_dtype=torch.float32))
    report(tag + "  fp16 halves, 3 terms, 1 GEMM", out / (sa * sb), ref, ms, flops)
    def three():
        o = torch.mm(a1, b1, out_dtype=torch.float32)
        o += torch.mm(a1, b2, out_dtype=torch.float32)
        o += torch.mm(a2, b1, out_dtype=torch.float32)
        return o
    out, ms = timed(three)
    report(tag + "  fp16 halves, 3 GEMMs + adds", out / (sa * sb), ref, ms, flops)
    out, ms = timed(lambda: torch.mm(a1, b1, out_dtype=torch.float32))
    report(tag + "  fp16 single term (reference point)", out / (sa * sb), ref, ms, flops)
    # bf16 thirds, 6 terms
    (p1, p2, p3), _ = split(a.contiguous(), torch.bfloat16, 3)
    (q1, q2, q3), _ = split(b.contiguous(), torch.bfloat16, 3)
    A = torch.cat([p1, p1, p2, p1, p2, p3], 1).contiguous()
    B = torch.cat([q1, q2, q1, q3, q2, q1], 0).contiguous()
    out, ms = timed(lambda: torch.mm(A, B, out_dtype=torch.float32))
    report(tag + "  bf16 thirds, 6 terms, 1 GEMM", out, ref, ms, flops)
    t0 = timed(lambda: split(a.contiguous()))[1]
    print(f"{'':44s} split of the left operand with torch ops: {t0:.3f} ms")
