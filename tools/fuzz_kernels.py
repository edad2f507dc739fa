"""Randomised differential test of the HIP operators against plain torch (autograd included): random graphs from empty to
dense (mean degree up to ~250, hubs, isolated nodes), random H / D / row pitches, edge logits, keep masks, residual addends.
Complements tests/test_gpu_parity.py::test_random_shapes_against_torch with the backward passes and the dense-graph paths.

    python tools/fuzz_kernels.py [trials=200] [seed=0]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bot_amd
from bot_amd import _C, ops, blocked

DEV = "cuda"
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gen = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))


def amax(t):
    return float(t.detach().abs().max()) if t.numel() else 0.0


def close(got, want, rel):
    return got.numel() == 0 or float((got.detach().double() - want.detach().double()).abs().max()) <= rel * max(1.0, amax(want))


def ref_attention(src, dst, n, el, er, ee, keep, slope):
    z = el[src] + (er[dst] if er is not None else 0) + (ee if ee is not None else 0)
    e = F.leaky_relu(z, slope)
    if keep is not None:
        e = e.masked_fill(keep.view(-1, 1, 1) == 0, float("-inf"))
    m = torch.full((n,) + e.shape[1:], float("-inf"), device=e.device, dtype=e.dtype).scatter_reduce(
        0, dst.view(-1, 1, 1).expand_as(e), e, "amax", include_self=True)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    ex = torch.exp(e - m[dst])
    s = torch.zeros_like(m).index_add_(0, dst, ex)
    return ex / s[dst].clamp_min(1e-38)


blocked_hits = 0
flat_hits = 0
for t in range(trials):
    n = ri(1, 400)
    dense = ri(0, 2) == 0
    e_raw = ri(0, n * ri(100, 250)) if dense else ri(0, 6 * n)
    src = torch.randint(0, n, (e_raw,), generator=gen)
    dst = (n * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** (1.0 + 1.5 * float(torch.rand((), generator=gen)))).long().clamp_(max=n - 1)
    g = bot_amd.Graph(src, dst, n, chunk=ri(4, 64) if not dense else None).to(DEV)
    E = g.number_of_edges()
    H, D = ri(1, 9), ri(1, 300) if ri(0, 2) else 4 * ri(1, 64)
    pad = ri(0, 2) * 4
    s_d, d_d = g.edges()
    # --- u_mul_e_sum with addend, forward + backward, strided x
    buf = torch.randn(n, H * D + pad, generator=gen, dtype=torch.float64).to(DEV)
    x64 = buf[:, :H * D].unflatten(1, (H, D)).clone().requires_grad_()
    a64 = torch.rand(E, H, 1, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
    r64 = torch.randn(n, H, D, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
    gout = torch.randn(n, H, D, generator=gen, dtype=torch.float64).to(DEV)
    ref = torch.zeros(n, H, D, device=DEV, dtype=torch.float64).index_add_(0, d_d, x64[s_d] * a64) + r64
    (ref * gout).sum().backward()
    xb = buf.float()
    x = xb[:, :H * D].unflatten(1, (H, D)).detach().requires_grad_()
    a = a64.detach().float().requires_grad_()
    r = r64.detach().float().requires_grad_()
    plan = blocked.plan_for(g.csc, n, H, D) if E else None
    blocked_hits += plan is not None
    out = ops.u_mul_e_sum(g, x, a, order="eid", addend=r)
    (out * gout.float()).sum().backward()
    assert close(out, ref, 1e-4), ("u_mul_e_sum", t, n, E, H, D, pad)
    for got, want, nm in ((x.grad, x64.grad, "dx"), (a.grad, a64.grad, "da"), (r.grad, r64.grad, "dr")):
        assert close(got, want, 1e-4), (nm, t, n, E, H, D, pad)
    # --- attention forward + backward with optional er / ee / keep
    Ha = ri(1, 10)
    el64 = torch.randn(n, Ha, 1, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
    er64 = torch.randn(n, Ha, 1, generator=gen, dtype=torch.float64).to(DEV).requires_grad_() if ri(0, 1) else None
    ee64 = torch.randn(E, Ha, 1, generator=gen, dtype=torch.float64).to(DEV).requires_grad_() if ri(0, 1) else None
    keep = (torch.rand(E, generator=gen) < 0.8).to(torch.uint8).to(DEV) if ri(0, 2) == 0 else None
    ga = torch.randn(E, Ha, 1, generator=gen, dtype=torch.float64).to(DEV)
    ra = ref_attention(s_d, d_d, n, el64, er64, ee64, keep, 0.2)
    (ra * ga).sum().backward()
    el = el64.detach().float().requires_grad_()
    er = er64.detach().float().requires_grad_() if er64 is not None else None
    ee = ee64.detach().float().requires_grad_() if ee64 is not None else None
    aa = ops.gat_attention(g, el, er, ee, keep=keep, negative_slope=0.2, order="eid")
    (aa * ga.float()).sum().backward()
    assert close(aa, ra, 2e-6), ("attention", t, n, E, Ha)
    for got, want, nm in ((el.grad, el64.grad, "del"), (None if er is None else er.grad, None if er64 is None else er64.grad, "der"),
                          (None if ee is None else ee.grad, None if ee64 is None else ee64.grad, "dee")):
        if want is not None:
            assert close(got, want, 2e-5), (nm, t, n, E, Ha)
    # --- copy_u_sum (odd widths are padded inside) and its backward
    W = ri(1, 70)
    f64 = torch.randn(n, W, generator=gen, dtype=torch.float64).to(DEV).requires_grad_()
    gw = torch.randn(n, W, generator=gen, dtype=torch.float64).to(DEV)
    rc = torch.zeros(n, W, device=DEV, dtype=torch.float64).index_add_(0, d_d, f64[s_d])
    (rc * gw).sum().backward()
    f = f64.detach().float().requires_grad_()
    oc = ops.copy_u_sum(g, f)
    (oc * gw.float()).sum().backward()
    assert close(oc, rc, 1e-4), ("copy_u_sum", t, n, E, W)
    assert close(f.grad, f64.grad, 1e-4), ("copy_u_sum bwd", t, n, E, W)
    # --- inference-only fused layer (round 2): logits + softmax + aggregation + residual + affine + ReLU in one sweep
    Hi = ri(1, 8)
    Di = ri(1, 300) if ri(0, 2) else 4 * ri(1, 64)
    if E and Di <= (1024 if Di % 4 == 0 else 512 if Di % 2 == 0 else 256):
        csc = g.csc
        xi = torch.randn(n, Hi, Di, generator=gen, dtype=torch.float64).to(DEV)
        eli = torch.randn(n, Hi, generator=gen, dtype=torch.float64).to(DEV)
        eri = torch.randn(n, Hi, generator=gen, dtype=torch.float64).to(DEV) if ri(0, 1) else None
        ewi = (torch.rand(E, generator=gen, dtype=torch.float64) + 0.5).to(DEV) if ri(0, 1) else None
        adi = torch.randn(n, Hi, Di, generator=gen, dtype=torch.float64).to(DEV) if ri(0, 1) else None
        sci = (torch.rand(Hi * Di, generator=gen, dtype=torch.float64) + 0.5).to(DEV) if ri(0, 1) else None
        shi = torch.randn(Hi * Di, generator=gen, dtype=torch.float64).to(DEV) if ri(0, 1) else None
        relu = bool(ri(0, 1))
        srcp = csc.indices.long()
        dstp = torch.repeat_interleave(torch.arange(n, device=DEV), (csc.indptr[1:] - csc.indptr[:-1]).long())
        ai = ref_attention(srcp, dstp, n, eli.unsqueeze(-1), None if eri is None else eri.unsqueeze(-1), None, None, 0.2)
        if ewi is not None:
            ai = ai * ewi.view(-1, 1, 1)
        refi = torch.zeros(n, Hi, Di, device=DEV, dtype=torch.float64).index_add_(0, dstp, xi[srcp] * ai)
        if adi is not None:
            refi = refi + adi
        refi = refi.reshape(n, -1)
        refi = refi * sci if sci is not None else refi
        refi = refi + shi if shi is not None else refi
        refi = torch.relu(refi) if relu else refi
        f32 = lambda t: None if t is None else t.float()
        outi = _C.gat_infer(csc, f32(xi), f32(eli), f32(eri), None, f32(ewi), 0.2, addend=f32(adi), scale=f32(sci), shift=f32(shi), relu=relu)
        assert close(outi.reshape(n, -1), refi, 1e-4), ("gat_infer", t, n, E, Hi, Di)
        # --- first-layer weight gradient: one in-edge sweep, the source row gathered once for all heads
        if Hi <= 4:
            xs = torch.randn(n, Di, generator=gen, dtype=torch.float64).to(DEV)
            ys = torch.randn(Hi, n, Di, generator=gen, dtype=torch.float64).to(DEV)
            refd = (xs[srcp].unsqueeze(1) * ys[:, dstp].permute(1, 0, 2)).sum(-1)
            assert close(_C.sddmm_dot_bcast(csc, xs.float(), ys.float()), refd, 1e-4), ("sddmm_dot_bcast", t, n, E, Hi, Di)
    # --- flat 16-byte-lane forward SpMM (round 3): 2..4 heads of a width that is not a multiple of 4, contiguous rows on an x4
    #     pitch — against fp64 and BITWISE against the head-segment kernel
    Hf, Df = ri(2, 4), ri(5, 255)
    if E and Df % 4 and Hf * Df <= 1024:
        pitch = (Hf * Df + 3) // 4 * 4 + 4 * ri(0, 3)
        xfb = torch.randn(n, pitch, generator=gen, dtype=torch.float64).to(DEV)
        wf = torch.rand(E, Hf, generator=gen, dtype=torch.float64).to(DEV)
        adf = torch.randn(n, pitch, generator=gen, dtype=torch.float64).to(DEV) if ri(0, 1) else None
        csc = g.csc
        srcp = csc.indices.long()
        dstp = torch.repeat_interleave(torch.arange(n, device=DEV), (csc.indptr[1:] - csc.indptr[:-1]).long())
        x3 = xfb[:, :Hf * Df].unflatten(1, (Hf, Df))
        reff = torch.zeros(n, Hf, Df, device=DEV, dtype=torch.float64).index_add_(0, dstp, x3[srcp] * wf.unsqueeze(-1))
        ad3 = None if adf is None else adf[:, :Hf * Df].unflatten(1, (Hf, Df))
        if ad3 is not None:
            reff = reff + ad3
        xs32 = xfb.float()[:, :Hf * Df].unflatten(1, (Hf, Df))
        ad32 = None if adf is None else adf.float()[:, :Hf * Df].unflatten(1, (Hf, Df))
        outs = {}
        for lay in ("flat", "segments"):
            _C.SPMM_LAYOUT = lay
            outs[lay] = _C.spmm(csc, xs32, wf.float(), torch.arange(E, dtype=torch.int32, device=DEV), addend=ad32)
            if lay == "flat":
                flat_hits += "flat" in _C._lib.bot_last_kernel().decode()
        _C.SPMM_LAYOUT = None
        assert close(outs["flat"], reff, 1e-4), ("spmm flat", t, n, E, Hf, Df, pitch)
        assert torch.equal(outs["flat"], outs["segments"]), ("spmm flat vs segments", t, n, E, Hf, Df, pitch)
    # --- dense products of round 2: random shapes, strides, magnitudes (fp64 reference; errors relative to the largest entry)
    m, k, nn = ri(1, 3000), ri(1, 256), ri(1, 300)
    mag = 10.0 ** ri(-9, 4)
    lda = k + ri(0, 5)
    A = (torch.randn(m, lda, generator=gen) * mag).to(DEV)[:, :k]
    Bw = (torch.randn(nn, k, generator=gen) * 0.3).to(DEV)
    refg = A.double() @ Bw.double().t()
    ldc = nn + ri(0, 7)
    Cbuf = torch.randn(m, ldc, generator=gen).to(DEV) * mag
    base = Cbuf[:, :nn].double().clone()
    acc = ri(0, 1) == 1
    b_kn = ri(0, 1) == 1
    _C.skinny_gemm(A, Bw.t().contiguous() if b_kn else Bw, b_is_kn=b_kn, out=Cbuf[:, :nn], accumulate=acc)
    want = refg + base if acc else refg
    assert close(Cbuf[:, :nn], want, 3e-6 * max(1.0, amax(want)) / max(1.0, amax(want))) or \
        float((Cbuf[:, :nn].double() - want).abs().max()) <= 3e-6 * amax(want), ("skinny_gemm", t, m, k, nn, mag, acc, b_kn)
    ky = ri(1, 300)
    Yt = (torch.randn(m, ky, generator=gen)).to(DEV)
    reft = A.double().t() @ Yt.double()
    gott = _C.tn_gemm(A, Yt)
    assert float((gott.double() - reft).abs().max()) <= 3e-6 * max(amax(reft), 1e-300), ("tn_gemm", t, m, k, ky, mag)
    if t % 8 == 0:
        from bot_amd import gemm
        mm = ri(8192, 12000)
        X = (torch.randn(mm, k, generator=gen) * mag).to(DEV)
        xs = gemm.split(X, 0)
        refh = X.double() @ Bw.double().t()
        goth = gemm.mm_nt(xs, gemm.split(Bw, 1))
        assert float((goth.double() - refh).abs().max()) <= 4e-6 * amax(refh), ("gemm_halves fwd", t, mm, k, nn, mag)
        Dm = (torch.randn(mm, nn, generator=gen) * mag).to(DEV)
        refw = X.double().t() @ Dm.double()
        gotw = gemm.tn(xs, gemm.split(Dm, 0))
        assert float((gotw.double() - refw).abs().max()) <= 4e-6 * amax(refw), ("gemm_halves tn", t, mm, k, nn, mag)
print(f"fuzz ok: {trials} trials (seed {seed}), {blocked_hits} of them on the L2-blocked path, {flat_hits} flat-lane SpMM launches")
