#!/usr/bin/env python3
"""The B-direct form of the NT halves GEMM (csrc/halves3.hip gemm_halves3_nt_bd_kernel, mode bit 512) against the shipped kernel:
bitwise equality on ragged shapes (plain and grouped), then interleaved timings on the four NT shapes of config 2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import _C, gemm  # noqa: E402

dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(1)
BD = int(os.environ.get("BD_MODE", "512"))


def t_ms(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def left(x, order=2):
    piece = (x.shape[1] + gemm.PIECE_ALIGN - 1) // gemm.PIECE_ALIGN * gemm.PIECE_ALIGN
    scale = _C.halves_scale(x)
    return gemm.Halves(_C.halves_split(x, scale, order, piece), scale, x.shape[0], x.shape[1], piece, order)


def h3(xs, ws, out=None, mode=0):
    return _C.gemm_halves3_nt(xs.buf, ws.buf, xs.scale, ws.scale, xs.piece, ws.piece, xs.piece, out=out, mode=mode, a2_off=xs.h2_off)


ok = True
for (m, K, P) in ((1000, 96, 300), (513, 750, 1536), (20000, 1536, 750), (4099, 64, 40), (257, 32, 17), (70000, 750, 240)):
    x = torch.randn(m, K, device=dev, generator=gen) * 3
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    for order in (0, 2):
        xs, ws = left(x, order), gemm.split(w, 1)
        ref = x.double() @ w.double().t()
        a = h3(xs, ws)
        b = h3(xs, ws, mode=BD)
        same = torch.equal(a, b) and all(torch.equal(h3(xs, ws, mode=BD), b) for _ in range(3))
        ok &= same
        print(f"m={m} K={K} P={P} order {order}: bd == shipped bitwise {same}; err vs fp64 {float((b.double() - ref).abs().max() / ref.abs().max()):.2e}")
    # strided / 8-byte-pitch outputs
    xs, ws = left(x, 2), gemm.split(w, 1)
    big = torch.zeros(m, P + 6, device=dev)
    o1, o2 = big[:, 2:2 + P], torch.zeros(m, P + 6, device=dev)[:, 2:2 + P]
    h3(xs, ws, out=o1), h3(xs, ws, out=o2, mode=BD)
    ok &= torch.equal(o1, o2)
    print(f"   strided output equal {torch.equal(o1, o2)}")
print("ALL BITWISE EQUAL" if ok else "MISMATCH")

N = 169343
for name, (m, K, P) in (("fwd [N,750]x[1536,750]", (N, 750, 1536)), ("dx [N,1536]x[750,1536]", (N, 1536, 750)), ("out [N,750]x[240,750]", (N, 750, 240)),
                        ("dx out [N,240]x[750,240]", (N, 240, 750))):
    x = torch.randn(m, K, device=dev, generator=gen)
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    xs, ws = left(x, 2), gemm.split(w, 1)
    out = torch.empty(m, P, device=dev)
    fs = {0: (lambda: h3(xs, ws, out)), BD: (lambda: h3(xs, ws, out, mode=BD)), 1: (lambda: h3(xs, ws, out, mode=1)), BD | 1: (lambda: h3(xs, ws, out, mode=BD | 1))}
    if "--ablate" in sys.argv:
        for a in (1, 2, 3, 4):
            fs[BD | (a << 10)] = (lambda a=a: h3(xs, ws, out, mode=BD | (a << 10)))
            fs[BD | (a << 10) | 1] = (lambda a=a: h3(xs, ws, out, mode=BD | (a << 10) | 1))
    for f in fs.values():
        f()
    rounds = [{k: t_ms(f) for k, f in fs.items()} for _ in range(5)]
    med = {k: sorted(r[k] for r in rounds)[2] for k in fs}
    fl = 2.0 * m * 3 * xs.piece * P
    if "--ablate" in sys.argv:
        print("   B-direct ablations (with stores / without): " + "   ".join(f"{what} {med[BD | (a << 10)]:.3f} / {med[BD | (a << 10) | 1]:.3f}" for a, what in
              ((1, "no A DMA in loop"), (2, "no B loads in loop"), (3, "neither"), (4, "setprio 1 waves 4-7"))))
    print(f"NT {name}: shipped {med[0]:.3f} ms ({fl / med[0] / 1e9:.0f} TF)   B-direct {med[BD]:.3f} ms ({fl / med[BD] / 1e9:.0f} TF)   without stores: {med[1]:.3f} / {med[BD | 1]:.3f}")
