#!/usr/bin/env python3
"""The 128-byte-line NT halves GEMM (csrc/halves3.hip gemm_halves3_nt64_kernel) against the 128 x 64-wave-tile kernel (mode bit 1024 forces it):
bitwise equality on ragged shapes (plain, dual-scale, grouped), then interleaved timings on the NT shapes of config 2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import _C, gemm  # noqa: E402
from bot_amd.nn import fused  # noqa: E402

dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(1)
OLD = 1024


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


def h3(xs, ws, out=None, mode=0, **kw):
    return _C.gemm_halves3_nt(xs.buf, ws.buf, xs.scale, ws.scale, xs.piece, ws.piece, xs.piece, out=out, mode=mode, a2_off=xs.h2_off, **kw)


ok = True
for (m, K, P) in ((1000, 96, 300), (513, 750, 1536), (20000, 1536, 750), (4099, 64, 40), (257, 128, 17), (70000, 750, 240)):
    x = torch.randn(m, K, device=dev, generator=gen) * 3
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    for order in (0, 2):
        xs, ws = left(x, order), gemm.split(w, 1)
        ref = x.double() @ w.double().t()
        a = h3(xs, ws, mode=OLD)
        b = h3(xs, ws)
        same = torch.equal(a, b) and all(torch.equal(h3(xs, ws), b) for _ in range(3))
        ok &= same
        print(f"m={m} K={K} P={P} order {order}: nt64 == 128x64 bitwise {same}; err vs fp64 {float((b.double() - ref).abs().max() / ref.abs().max()):.2e}")
    xs, ws = left(x, 2), gemm.split(w, 1)
    o1, o2 = torch.zeros(m, P + 6, device=dev)[:, 2:2 + P], torch.zeros(m, P + 6, device=dev)[:, 2:2 + P]
    h3(xs, ws, out=o1, mode=OLD), h3(xs, ws, out=o2)
    ok &= torch.equal(o1, o2)
    if xs.piece >= 64:
        s2 = torch.tensor([float(xs.scale[0]) * 128, float(xs.scale[1]) / 128], device=dev)
        for split in (32, xs.piece - 32):
            d1, d2 = h3(xs, ws, mode=OLD, scale_a2=s2, k_split=split), h3(xs, ws, scale_a2=s2, k_split=split)
            ok &= torch.equal(d1, d2)
            print(f"   strided output equal {torch.equal(o1, o2)}; dual scale split {split} equal {torch.equal(d1, d2)}")
# grouped, the config-2 layer-0 arrangement
for (N, H, D, Fin, kp) in ((20011, 3, 250, 168, True), (5000, 2, 70, 40, False)):
    P2 = (H * D + 2 * H + 127) // 128 * 128
    FP, DP, g_fwd, g_dz, t_tn = fused._l0_tables(H, D, Fin, P2, kp, N)
    HD, KA = H * D, (1 + H) * FP
    A = (torch.randn(N, 2 * KA, device=dev, generator=gen) * 50).half()
    B = (torch.randn(HD, 6 * FP, device=dev, generator=gen) * 30).half()
    sa, sb = torch.tensor([4.0, 0.25], device=dev), torch.tensor([8.0, 0.125], device=dev)
    o1, o2 = torch.zeros(N, P2, device=dev), torch.zeros(N, P2, device=dev)
    _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, o1, g_fwd, FP // 32, mode=OLD)
    _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, o2, g_fwd, FP // 32)
    ok &= torch.equal(o1, o2)
    print(f"grouped forward N={N} H={H} D={D} Fin={Fin}: equal {torch.equal(o1, o2)}  (k_steps {2 * FP // 32}, k_seg {FP // 32})")
for (m, K, P) in ((1000, 96, 300), (513, 750, 1536), (4099, 64, 40), (257, 128, 17), (3000, 250, 193)):
    x = torch.randn(m, K, device=dev, generator=gen) * 3
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    xs, ws = left(x, 2), gemm.split(w, 1)
    wf = gemm.Halves(_C.halves_split_frag(w, ws.scale, ws.piece), ws.scale, P, K, ws.piece, 3)
    a = h3(xs, ws)
    b = _C.gemm_halves3_nt(xs.buf, wf.buf, xs.scale, wf.scale, xs.piece, wf.piece, xs.piece, a2_off=xs.h2_off, b_frag=True, n=P)
    ok &= torch.equal(a, b)
    print(f"fragment-major B m={m} K={K} P={P}: equal {torch.equal(a, b)}")
print("ALL BITWISE EQUAL" if ok else "MISMATCH")

N = 169343
for name, (m, K, P) in (("fwd [N,750]x[1536,750]", (N, 750, 1536)), ("dx [N,1536]x[750,1536]", (N, 1536, 750)), ("out [N,750]x[240,750]", (N, 750, 240)),
                        ("dx out [N,240]x[750,240]", (N, 240, 750))):
    x = torch.randn(m, K, device=dev, generator=gen)
    w = torch.randn(P, K, device=dev, generator=gen) * 0.05
    xs, ws = left(x, 2), gemm.split(w, 1)
    out = torch.empty(m, P, device=dev)
    wf = gemm.split_right(w)
    assert wf.order == 3
    f_frag = lambda: _C.gemm_halves3_nt(xs.buf, wf.buf, xs.scale, wf.scale, xs.piece, wf.piece, xs.piece, out=out, a2_off=xs.h2_off, b_frag=True, n=wf.n)
    ref_o = h3(xs, ws).clone()
    f_frag()
    print("   fragment-major B == row-major B bitwise:", torch.equal(out, ref_o))
    fs = {"128x64": (lambda: h3(xs, ws, out, mode=OLD)), "nt64": (lambda: h3(xs, ws, out)), "nt64 frag": f_frag}
    for f in fs.values():
        f()
    rounds = [{k: t_ms(f) for k, f in fs.items()} for _ in range(5)]
    med = {k: sorted(r[k] for r in rounds)[2] for k in fs}
    fl = 2.0 * m * 3 * xs.piece * P
    print(f"NT {name}: 128x64 {med['128x64']:.3f} ms ({fl / med['128x64'] / 1e9:.0f} TF)   nt64 {med['nt64']:.3f} ms ({fl / med['nt64'] / 1e9:.0f} TF)   "
          f"nt64 + fragment-major B {med['nt64 frag']:.3f} ms ({fl / med['nt64 frag'] / 1e9:.0f} TF)")
