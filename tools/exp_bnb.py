"""Cost of the BatchNorm-backward by-product on the NT halves product (ABI 18), in isolation, at the two config-2 shapes that carry it:
   python tools/exp_bnb.py   ->  us per launch: plain, with the by-product (dropout 0 / 0.75), and the reduce pass it replaces."""
import torch

from bot_amd import _C, gemm

DEV = "cuda"
N, F = 169343, 750


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


gen = torch.Generator(device=DEV).manual_seed(1)
xbuf = torch.randn(N, 752, device=DEV, generator=gen)
x = xbuf[:, :F]
mean, invstd = x.mean(0), (x.var(0, unbiased=False) + 1e-5).rsqrt()
bw, bb = torch.randn(F, device=DEV, generator=gen), torch.randn(F, device=DEV, generator=gen)
for K in (128, 1536):
    d = torch.randn(N, K, device=DEV, generator=gen)
    w = torch.randn(F, K, device=DEV, generator=gen) * 0.1
    ws = gemm.split(w, 1)
    piece = ws.piece
    frag = _C.halves_split_frag(w, ws.scale, piece)
    sc = _C.halves_scale(d)
    db = _C.halves_split(d, sc, 2, piece)
    out = torch.empty(N, F, device=DEV)
    kw = dict(a2_off=piece, b_frag=True, n=F, out=out)
    t0 = timed(lambda: _C.gemm_halves3_nt(db, frag, sc, ws.scale, piece, piece, piece, **kw))
    res = [f"K={K}: plain {t0:.0f} us"]
    for p in (0.0, 0.75):
        st = _C.BnBwdStats(x, mean, invstd, bw, bb, True, p, 1234)
        t1 = timed(lambda: _C.gemm_halves3_nt(db, frag, sc, ws.scale, piece, piece, piece, bn=st, **kw))
        t2 = timed(lambda: _C.bn_act_bwd_reduce(out, x, mean, invstd, bw, bb, True, p, 1234, want_max=True))
        t3 = timed(lambda: (st.sums(), st.bound(None, None, N, _C.absmax_slots(DEV))))
        res.append(f"p={p}: by-product {t1:.0f} (+{t1 - t0:.0f}), reduce pass {t2:.0f}, second stage {t3:.0f}")
    print("; ".join(res))
