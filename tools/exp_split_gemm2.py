"""Experiment 2: layouts for the fp16-halves GEMMs of config-2 layer 1 (see exp_split_gemm.py): padded reduction axes,
the weight gradient as row-chunked batched products.   python tools/exp_split_gemm2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(0)
N = 169343

def timed(f, k=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): out = f()
    torch.cuda.synchronize(); return out, (time.perf_counter() - t0) / k * 1e3

def halves(x):
    s = 2.0 ** (14 - torch.ceil(torch.log2(x.abs().max())).item())
    r = x * s
    h1 = r.half(); h2 = (r - h1.float()).half()
    return h1, h2, s

def padcols(t, w):
    out = t.new_zeros(t.shape[0], w); out[:, :t.shape[1]] = t; return out

def show(name, got, ref, ms, flops):
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{name:58s} {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF-eq  err {err:.1e}", flush=True)

x = torch.relu(torch.randn(N, 750, device=dev, generator=gen)) * (torch.rand(N, 750, device=dev, generator=gen) > 0.75) * 4
dy = torch.randn(N, 1536, device=dev, generator=gen) * 1e-7
dy[:, 1506:] = 0
w = torch.randn(750, 1536, device=dev, generator=gen) * 0.05      # [K, P] layout
w[:, 1506:] = 0
x1, x2, sx = halves(x); d1, d2, sd = halves(dy); w1, w2, sw = halves(w)

# ---- forward  out[N,P] = x w
flops = 2.0 * N * 750 * 1506
ref = x.double() @ w.double()
out, ms = timed(lambda: x @ w); show("fwd native fp32 (P=1536)", out, ref, ms, flops)
for KP in (750, 752, 768):
    A = torch.cat([padcols(x1, KP), padcols(x1, KP), padcols(x2, KP)], 1).contiguous()
    B = torch.cat([padcols(w1.t(), KP).t(), padcols(w2.t(), KP).t(), padcols(w1.t(), KP).t()], 0).contiguous()
    out, ms = timed(lambda: torch.mm(A, B, out_dtype=torch.float32)); show(f"fwd halves, K piece padded to {KP}", out / (sx * sw), ref, ms, flops)
    Bt = B.t().contiguous()
    out, ms = timed(lambda: torch.mm(A, Bt.t(), out_dtype=torch.float32)); show(f"fwd halves, K piece {KP}, weight stored [P,3K]", out / (sx * sw), ref, ms, flops)

# ---- dX[N,K] = dy w^T   (reduce over P)
flops = 2.0 * N * 1506 * 750
ref = dy.double() @ w.double().t()
out, ms = timed(lambda: dy @ w.t()); show("dX native fp32", out, ref, ms, flops)
A = torch.cat([d1, d1, d2], 1).contiguous()                       # [N, 3P]
B = torch.cat([w1, w2, w1], 1).contiguous()                       # [K, 3P]
out, ms = timed(lambda: torch.mm(A, B.t(), out_dtype=torch.float32)); show("dX halves, weight [K,3P] (NT)", out / (sd * sw), ref, ms, flops)
Bn = B.t().contiguous()
out, ms = timed(lambda: torch.mm(A, Bn, out_dtype=torch.float32)); show("dX halves, weight [3P,K] (NN)", out / (sd * sw), ref, ms, flops)
Bn2 = padcols(Bn, 768)
out, ms = timed(lambda: torch.mm(A, Bn2, out_dtype=torch.float32)); show("dX halves, weight [3P,768] (NN, padded out)", out[:, :750] / (sd * sw), ref, ms, flops)

# ---- dW[K,P] = x^T dy   (reduce over N)
flops = 2.0 * N * 1506 * 750
ref = x.double().t() @ dy.double()
out, ms = timed(lambda: x.t() @ dy); show("dW native fp32", out, ref, ms, flops)
X = torch.cat([x1, x1, x2], 1).contiguous()                       # the forward's left operand [N, 3K]
D2 = torch.cat([d1, d2], 1).contiguous()                          # [N, 2P]
for S in (4, 8, 16, 32):
    R = N // S
    def f():
        # terms x1^T d1, x1^T d2 (one product against [d1|d2]) and x2^T d1, per row chunk, summed afterwards
        Xc = X[:S * R].view(S, R, -1)
        a = torch.bmm(Xc[:, :, :750].transpose(1, 2), D2[:S * R].view(S, R, -1), out_dtype=torch.float32).sum(0)
        b = torch.bmm(Xc[:, :, 1500:].transpose(1, 2), A[:S * R].view(S, R, -1)[:, :, :1536], out_dtype=torch.float32).sum(0)
        o = a[:, :1536] + a[:, 1536:] + b
        if S * R < N:
            o += torch.mm(X[S * R:, :750].t(), D2[S * R:], out_dtype=torch.float32).view(750, 2, 1536).sum(1) + \
                 torch.mm(X[S * R:, 1500:].t(), A[S * R:, :1536], out_dtype=torch.float32)
        return o
    out, ms = timed(f); show(f"dW halves, {S} row chunks (2 batched products)", out / (sx * sd), ref, ms, flops)
Xr = torch.cat([x1, x2, x1], 0).contiguous(); Dr = torch.cat([d1, d1, d2], 0).contiguous()
out, ms = timed(lambda: torch.mm(Xr.t(), Dr, out_dtype=torch.float32)); show("dW halves, rows concatenated (one GEMM)", out / (sx * sd), ref, ms, flops)
