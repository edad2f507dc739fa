#!/usr/bin/env python3
"""bot_gemm_halves3_nt_f32 (csrc/halves3.hip) against the hipBLASLt formulation (bot_gemm_halves_f32) on the config-2 NT shapes:
error of both against fp64, bitwise run-to-run and against the plain-loop build, interleaved timings in one process (HIP events, random
operands), and the ablation switches of the plain loop (what the stores, the LDS-DMA, the barrier and the A-fragment reads cost)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import _C, gemm, tuning  # noqa: E402

tuning.enable()
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(1)
PLAIN = 32


def t_ms(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def h3(xs, ws, out=None, mode=0):
    return _C.gemm_halves3_nt(xs.buf, ws.buf, xs.scale, ws.scale, xs.piece, ws.piece, xs.piece, out=out, mode=mode, a2_off=xs.h2_off)


def left(x, order=0):
    """a left operand in a GIVEN layout (0: [h1 | h1 | 2^11 h2], what the library form needs; 2: [h1 | 2^11 h2])"""
    piece = (x.shape[1] + gemm.PIECE_ALIGN - 1) // gemm.PIECE_ALIGN * gemm.PIECE_ALIGN
    scale = _C.halves_scale(x)
    return gemm.Halves(_C.halves_split(x, scale, order, piece), scale, x.shape[0], x.shape[1], piece, order)


if "--layouts" in sys.argv:  # both kernels on the config-2 shapes with the left operands in either layout (row pitch 3 or 2 pieces)
    N = 169343
    for name, (m, K, P) in (("fwd [N,750]x[1536,750]", (N, 750, 1536)), ("dx [N,1536]x[750,1536]", (N, 1536, 750)), ("out [N,750]x[240,750]", (N, 750, 240)),
                            ("dx out [N,240]x[750,240]", (N, 240, 750))):
        x = torch.randn(m, K, device=dev, generator=gen)
        w = torch.randn(P, K, device=dev, generator=gen) * 0.05
        ws = gemm.split(w, 1)
        out = torch.empty(m, P, device=dev)
        ops = {o: left(x, o) for o in (0, 2)}
        fs = {o: (lambda o=o: h3(ops[o], ws, out)) for o in (0, 2)}
        for f in fs.values():
            f()
        rounds = [tuple(t_ms(fs[o]) for o in (0, 2)) for _ in range(5)]
        print(f"NT {name}: 3-piece left {sorted(r[0] for r in rounds)[2]:.3f} ms   2-piece left {sorted(r[1] for r in rounds)[2]:.3f} ms")
    for name, (K, P) in (("[750,N]x[N,1536]", (750, 1536)), ("[1536,N]x[N,750]", (1536, 750)), ("[750,N]x[N,240]", (750, 240))):
        x = torch.randn(N, K, device=dev, generator=gen)
        d = torch.randn(N, P, device=dev, generator=gen) * 1e-3
        combos = {(a, b): (left(x, a), left(d, b)) for a in (0, 2) for b in (0, 2)}
        fs = {k: (lambda v=v: gemm.tn(*v)) for k, v in combos.items()}
        for f in fs.values():
            f()
        rounds = [{k: t_ms(f) for k, f in fs.items()} for _ in range(5)]
        print(f"TN {name}: " + "   ".join(f"x {3 if a == 0 else 2}-piece, d {3 if b == 0 else 2}-piece {sorted(r[(a, b)] for r in rounds)[2]:.3f} ms" for (a, b) in fs))
    sys.exit(0)


if "--pmc-tn" in sys.argv:   # a few launches of the TN kernel on the config-2 shape
    N = 169343
    x = torch.randn(N, 750, device=dev, generator=gen)
    d = torch.randn(N, 1536, device=dev, generator=gen) * 1e-3
    xs, ds = left(x), left(d)
    for _ in range(3):
        gemm.tn(xs, ds)
    torch.cuda.synchronize()
    sys.exit(0)

if "--pmc" in sys.argv:      # a few launches of each kernel on the forward shape, nothing else (tools/pmc_halves3.sh)
    m, K, P = 169343, 750, 1536
    x = torch.randn(m, K, device=dev, generator=gen)
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    xs, ws = left(x), gemm.split(w, 1)
    out = torch.empty(m, P, device=dev)
    for _ in range(3):
        _C.gemm_halves(xs.buf, ws.buf, gemm._alpha(xs, ws, P), trans_b=True, out=out)
        h3(xs, ws, out)
    torch.cuda.synchronize()
    sys.exit(0)

# correctness on ragged shapes
for (m, K, P) in ((1000, 96, 300), (513, 750, 1536), (20000, 1536, 750), (4099, 64, 40)):
    x = torch.randn(m, K, device=dev, generator=gen) * 3
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    xs, ws = left(x), gemm.split(w, 1)
    ref = x.double() @ w.double().t()
    lib = _C.gemm_halves(xs.buf, ws.buf, gemm._alpha(xs, ws, P), trans_b=True)
    mine = h3(xs, ws)
    plain = h3(xs, ws, mode=PLAIN)
    same = all(torch.equal(h3(xs, ws), mine) for _ in range(3))
    sc = ref.abs().max()
    print(f"m={m} K={K} P={P}: halves3 err {float((mine.double() - ref).abs().max() / sc):.2e}  hipBLASLt err {float((lib.double() - ref).abs().max() / sc):.2e}  "
          f"bitwise run-to-run {same}  bitwise == plain loop {torch.equal(mine, plain)}  max|halves3 - lib| {float((mine - lib).abs().max() / sc):.2e}")

# TN (weight gradient): correctness on ragged shapes, then the config-2 shape against the library formulation
for (n, K, P) in ((1000, 96, 300), (4099, 750, 1536), (50001, 168, 250), (33, 64, 40), (20000, 1536, 750)):
    x = torch.randn(n, K, device=dev, generator=gen) * 3
    d = torch.randn(n, P, device=dev, generator=gen) * 1e-3
    xs, ds = left(x), left(d)
    ref = x.double().t() @ d.double()
    gemm.TN_KERNEL = "lib"
    lib = gemm.tn(xs, ds)
    gemm.TN_KERNEL = "halves3"
    mine = gemm.tn(xs, ds)
    same = all(torch.equal(gemm.tn(xs, ds), mine) for _ in range(3))
    sc = ref.abs().max()
    print(f"TN n={n} K={K} P={P}: halves3 err {float((mine.double() - ref).abs().max() / sc):.2e}  library err {float((lib.double() - ref).abs().max() / sc):.2e}  "
          f"bitwise run-to-run {same}")
N = 169343
x = torch.randn(N, 750, device=dev, generator=gen)
d = torch.randn(N, 1536, device=dev, generator=gen) * 1e-3
xs, ds = left(x), left(d)


def tn_with(kind):
    gemm.TN_KERNEL = kind
    return gemm.tn(xs, ds)


for _ in range(3):
    tn_with("lib"), tn_with("halves3")
rounds = [(t_ms(lambda: tn_with("lib")), t_ms(lambda: tn_with("halves3"))) for _ in range(5)]
fl = 2.0 * N * 3 * 768 * 1536
a, b = (sorted(r[i] for r in rounds)[2] for i in range(2))
print(f"dW x^T d [750,N]x[N,1536]: hipBLASLt (batched chunks + combine) {a:.3f} ms ({fl / a / 1e9:.0f} TF)   halves3 TN {b:.3f} ms ({fl / b / 1e9:.0f} TF)   rounds {[tuple(round(v, 3) for v in r) for r in rounds]}")
gemm.TN_KERNEL = "halves3"
if "--ablate" in sys.argv:
    for md, what in ((1, "no DMA in the loop"), (2, "no barrier / vmcnt wait"), (3, "neither")):
        f = lambda md=md: _C.gemm_halves3_tn(xs.buf, ds.buf, xs.scale, ds.scale, xs.piece, ds.piece, 750, 1536, mode=md)
        f()
        print(f"   TN ablation {what}: {t_ms(f):.3f} ms")

# timing at the config-2 shapes
N = 169343
for name, (m, K, P) in (("fwd x W^T [N,750]x[1536,750]", (N, 750, 1536)), ("dx d W [N,1536]x[750,1536]", (N, 1536, 750))):
    x = torch.randn(m, K, device=dev, generator=gen)
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    xs, ws = left(x), gemm.split(w, 1)
    alpha = gemm._alpha(xs, ws, P)
    out1 = torch.empty(m, P, device=dev)
    out2 = torch.empty(m, P, device=dev)
    f_lib = lambda: _C.gemm_halves(xs.buf, ws.buf, alpha, trans_b=True, out=out1)
    f_new = lambda: h3(xs, ws, out2)
    f_plain = lambda: h3(xs, ws, out2, PLAIN)
    for _ in range(3):
        f_lib(), f_new(), f_plain()
    rounds = [(t_ms(f_lib), t_ms(f_new), t_ms(f_plain)) for _ in range(5)]
    fl = 2.0 * m * 3 * xs.piece * P
    a, b, c = (sorted(r[i] for r in rounds)[2] for i in range(3))
    print(f"{name}: hipBLASLt {a:.3f} ms ({fl / a / 1e9:.0f} TF)   halves3 {b:.3f} ms ({fl / b / 1e9:.0f} TF)   plain loop {c:.3f} ms   "
          f"rounds {[tuple(round(v, 3) for v in r) for r in rounds]}")
    if "--ablate" in sys.argv:
        for md, what in ((1, "pipelined, no stores"), (PLAIN | 1, "plain, no stores"), (PLAIN | 1 | 64, "plain, no stores, no barrier / vmcnt wait"),
                         (PLAIN | 1 | 128, "plain, no stores, no DMA in the loop"), (PLAIN | 1 | 64 | 128, "plain, no stores, no DMA, no barrier"),
                         (PLAIN | 1 | 256, "plain, no stores, one A fragment pair per k-step"), (PLAIN | 1 | 4, "plain, no stores, B fixed at k-step 0"),
                         (PLAIN | 1 | 8, "plain, no stores, A fixed"), (PLAIN | 1 | 12, "plain, no stores, A and B fixed")):
            f = lambda md=md: h3(xs, ws, out2, md)
            f()
            print(f"   ablation {what}: {t_ms(f):.3f} ms")
