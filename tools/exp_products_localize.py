"""Diagnostic: layer 0 of the S-products GAT at full size, every stage against a chunked torch restatement on the same GPU
(64-bit-safe torch index ops), to localise a full-size-only forward difference.   python tools/exp_products_localize.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bot_amd import workloads, ops

dev = torch.device("cuda", 0)
wl = workloads.build("products", dev, drop=False)
model, g, ds = wl.model, wl.graph, wl.dataset
conv = model.convs[0]
H, D = conv._n_heads, conv._out_feats
x = ds.feat
n = x.shape[0]
print("N", n, "E", g.number_of_edges(), "H", H, "D", D)
with torch.no_grad():
    parts = [conv.src_fc.weight, conv.dst_fc.weight, conv.attn_src_fc.weight, conv.attn_dst_fc.weight]
    bias = torch.cat([torch.zeros(H * D, device=dev), conv.dst_fc.bias, torch.zeros(2 * H, device=dev)])
    Wc = torch.cat(parts)
    Y = F.linear(x, Wc, bias)
    print("merged GEMM output", tuple(Y.shape), Y.numel(), "elements")
    worst = 0.0
    for r0 in range(0, n, 200_000):
        ref = F.linear(x[r0:r0 + 200_000], Wc, bias)
        e = (Y[r0:r0 + 200_000] - ref).abs().max().item()
        if e > 1e-4:
            print("  GEMM rows", r0, "differ by", e)
        worst = max(worst, e)
    print("merged GEMM vs chunked GEMM: max diff", worst)
    # a known-good merged output for the following stages
    Yg = torch.cat([F.linear(x[r0:r0 + 200_000], Wc, bias) for r0 in range(0, n, 200_000)])
    ft, res, el, er = torch.split(Yg, [H * D, H * D, H, H], dim=1)
    ft, res = ft.unflatten(1, (H, D)), res.unflatten(1, (H, D))
    el, er = el.unsqueeze(-1), er.unsqueeze(-1)
    csc = g.csc
    a = ops.gat_attention(g, el, er, None, negative_slope=0.2, order="csc")            # [E,H,1] in CSC position order
    src = csc.indices.long()
    deg = (csc.indptr[1:] - csc.indptr[:-1]).long()
    dst = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    z = F.leaky_relu(el.view(n, H)[src] + er.view(n, H)[dst], 0.2)
    m = torch.full((n, H), -float("inf"), device=dev).scatter_reduce_(0, dst[:, None].expand(-1, H), z, "amax")
    p = torch.exp(z - m[dst])
    s = torch.zeros((n, H), device=dev).index_add_(0, dst, p)
    aref = p / s[dst]
    print("attention: HIP finite", bool(torch.isfinite(a).all()), "restated finite", bool(torch.isfinite(aref).all()),
          "m finite", bool(torch.isfinite(m).all()), "s min", s.min().item(), "deg min", int(deg.min()))
    ok = torch.isfinite(aref).all(1)
    print("attention: max diff on finite rows", (a.view(-1, H) - aref)[ok].abs().max().item(), "non-finite restated edges", int((~ok).sum()))
    rowsum = torch.zeros((n, H), device=dev).index_add_(0, dst, a.view(-1, H))
    print("attention: HIP row sums in", rowsum.min().item(), rowsum.max().item())
    aref = a.view(-1, H)
    out = ops.u_mul_e_sum(g, ft, a, order="csc", addend=res)
    oref = res.clone()
    E = src.numel()
    for e0 in range(0, E, 2_000_000):
        sl = slice(e0, e0 + 2_000_000)
        oref.index_add_(0, dst[sl], ft[src[sl]] * aref[sl].unsqueeze(-1))
    dif = (out - oref).abs()
    print("aggregation (strided ft view, pitch %d): max diff" % ft.stride(0), dif.max().item(), "rows over 1e-3:", int((dif.amax((1, 2)) > 1e-3).sum()))
    ftc = ft.contiguous()
    out2 = ops.u_mul_e_sum(g, ftc, a, order="csc", addend=res.contiguous())
    print("aggregation (contiguous ft): max diff", (out2 - oref).abs().max().item())
    hip = conv(g, x)
    print("whole GATConv vs restated: max diff", (hip - oref).abs().max().item())
