"""Differentiable sparse operators over a `bot_amd.Graph`, each backed by hand-written gfx950 kernels
through the C ABI (include/bot_gnn.h).  They replace the `dgl.ops` calls the reference makes
(SURVEY §2.1): copy_u_sum, u_mul_e_sum, copy_e_sum, copy_u, u_add_v, edge_softmax (+ `eids`), and add
the fused attention op the layers use (`gat_attention`: logits + leaky-ReLU + softmax in one sweep).

Edge tensors cross this API in edge-id order, like DGL's.  `gat_attention` / `u_mul_e_sum` also accept
`order="csc"`: the attention weights then stay in destination-major position order between the two
ops (no permutation gathers); the layers in `bot_amd.nn` use that form.

Backward formulas (hand-derived; checked against autograd of the oracle's forward definitions):
  copy_u_sum      dx = copy_u_sum on the reversed graph (CSR sweep)
  u_mul_e_sum     dx[u] = sum_{e: u->v} a_e * dout[v]  (CSR sweep, weights through csr2csc);
                  da_e  = <x[u], dout[v]>              (SDDMM dot, CSC sweep)
  gat_attention   t_v = sum a*da;  de = a*(da - t_v);  dz = de * leaky'(z);
                  d_er[v] = sum_in dz;  d_el[u] = sum_out dz;  d_ee = dz
"""
from __future__ import annotations

import os

import torch

from . import _C
from .graph import take_rows

__all__ = ["copy_u_sum", "u_mul_e_sum", "copy_e_sum", "copy_u", "u_add_v", "edge_softmax", "gat_attention"]


def _as3(x):
    """[n], [n,F] or [n,H,D] -> [n,H,D] view."""
    if x.dim() == 1:
        return x.view(-1, 1, 1)
    if x.dim() == 2:
        return x.unsqueeze(1)
    if x.dim() == 3:
        return x
    return _flat2(x).unsqueeze(1)


def _flat2(t):
    """[n, ...] -> [n, prod(rest)] with an explicit width (n may be 0)."""
    w = 1
    for s in t.shape[1:]:
        w *= int(s)
    return t.reshape(t.shape[0], w)


def _edge2(a, H=None):
    """[E,H,1] / [E,H] / [E] -> [E,H]  (explicit width: E may be 0)."""
    w = 1
    for s in a.shape[1:]:
        w *= int(s)
    return a.reshape(a.shape[0], w)


def _pad4(x2):
    """[n,F] -> [n,F'] with F' the next multiple of 4 (zero columns): odd widths such as 41 classes would otherwise force
    4-byte lanes and one row per wavefront; padded they run with 16-byte lanes and several rows per wavefront."""
    F = x2.shape[1]
    return x2 if F % 4 == 0 or F < 5 else torch.nn.functional.pad(x2, (0, 4 - F % 4))


class _CopyUSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, x):
        ctx.g, ctx.shape = g, x.shape
        if x.dim() == 2:
            F = x.shape[1]
            out = _C.spmm(g.csc, _as3(_pad4(x)))
            return out.view(out.shape[0], -1)[:, :F].contiguous() if out.shape[2] != F else out.view(out.shape[0], F)
        return _C.spmm(g.csc, _as3(x)).view((g.number_of_dst_nodes(),) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        if len(ctx.shape) == 2:
            F = ctx.shape[1]
            dx = _C.spmm(g.csr, _as3(_pad4(dout.contiguous())))
            return None, (dx.view(dx.shape[0], -1)[:, :F].contiguous() if dx.shape[2] != F else dx.view(ctx.shape))
        return None, _C.spmm(g.csr, _as3(dout.contiguous())).view(ctx.shape)


def copy_u_sum(g, x):
    """`update_all(fn.copy_src('h','m'), fn.sum('m','h'))` — models.py:374,381."""
    return _CopyUSum.apply(g, x)


class _UMulESum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, x, a, order, addend=None):
        x3, a2 = _as3(x), _edge2(a)
        ctx.g, ctx.order, ctx.xshape, ctx.ashape = g, order, x.shape, a.shape
        ctx.save_for_backward(x3, a2)
        wperm = g.csc.eid if order == "eid" else None
        ad3 = None if addend is None else _as3(addend)
        return _C.spmm(g.csc, x3, a2, wperm, addend=ad3).view((g.number_of_dst_nodes(),) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        x3, a2 = ctx.saved_tensors
        d3 = _as3(dout.contiguous())
        dx = da = None
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and x3.shape[2] <= _C.spmm_dot_max_d(d3):
            # one sweep over the out-edges yields both gradients: each gathered dout row is used twice
            wperm = g.csr.eid if ctx.order == "eid" else g.csr2csc
            dx, da = _C.spmm_dot(g.csr, d3, a2, wperm, x3)
            return None, dx.view(ctx.xshape), da.view(ctx.ashape), None, (dout if ctx.needs_input_grad[4] else None)
        if ctx.needs_input_grad[1]:
            wperm = g.csr.eid if ctx.order == "eid" else g.csr2csc
            dx = _C.spmm(g.csr, d3, a2, wperm).view(ctx.xshape)
        if ctx.needs_input_grad[2]:
            operm = g.csc.eid if ctx.order == "eid" else None
            da = _C.sddmm_dot(g.csc, x3, d3, operm).view(ctx.ashape)
        return None, dx, da, None, (dout if ctx.needs_input_grad[4] else None)


def u_mul_e_sum(g, x, a, order="eid", addend=None):
    """`update_all(fn.u_mul_e('ft','a','m'), fn.sum('m','ft'))` — models.py:547.
    x: [N,H,D]; a: [E,H,1] (or [E,H]) in edge-id order, or in CSC position order with order="csc".
    `addend` [N_dst,H,D]: added to the result in the kernel's epilogue (`rst + dst_fc(feat_dst)`, ogbn-proteins/models.py:159-160;
    `rst + res_fc(h)`, models.py:558-560) instead of a separate pass; its gradient is the incoming one."""
    return _UMulESum.apply(g, x, a, order, addend)


class _CopyESum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, w):
        ctx.g, ctx.shape = g, w.shape
        return _C.segment_sum(g.csc, _edge2(w), g.csc.eid).view((g.number_of_dst_nodes(),) + tuple(w.shape[1:]))

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        return None, _C.u_add_v(g.dst32, None, _flat2(dout).contiguous()).view(ctx.shape)


def copy_e_sum(g, w):
    """`update_all(fn.copy_e('feat','m'), fn.sum('m','feat'))` — ogbn-proteins/gat.py:58."""
    return _CopyESum.apply(g, w)


class _UAddV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, x, y):
        ctx.g, ctx.xshape = g, x.shape
        ctx.has_y = y is not None
        ctx.yshape = y.shape if ctx.has_y else None
        x2 = _flat2(x)
        y2 = None if y is None else _flat2(y)
        return _C.u_add_v(g.src32, g.dst32 if ctx.has_y else None, x2, y2).view((g.number_of_edges(),) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, de):
        g = ctx.g
        de2 = _flat2(de).contiguous()
        dx = _C.segment_sum(g.csr, de2, g.csr.eid).view(ctx.xshape) if ctx.needs_input_grad[1] else None
        dy = _C.segment_sum(g.csc, de2, g.csc.eid).view(ctx.yshape) if ctx.has_y and ctx.needs_input_grad[2] else None
        return None, dx, dy


def copy_u(g, x):
    """`apply_edges(fn.copy_u('el','e'))` — models.py:525."""
    return _UAddV.apply(g, x, None)


def u_add_v(g, x, y):
    """`apply_edges(fn.u_add_v('el','er','e'))` — models.py:523."""
    return _UAddV.apply(g, x, y)


class _EdgeMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ef_csc, W1, b1, W2):
        ctx.save_for_backward(ef_csc, W1, b1, W2)
        return _C.edge_mlp_fwd(ef_csc, W1, b1, W2)

    @staticmethod
    def backward(ctx, dz):
        ef_csc, W1, b1, W2 = ctx.saved_tensors
        dW1, db1, dW2 = _C.edge_mlp_bwd(ef_csc, W1, b1, W2, dz.contiguous())
        return None, dW1, db1, dW2


def edge_features_csc(g, efeat):
    """Edge features permuted once into CSC position order (they are constant inputs of the model)."""
    cache = getattr(g, "_bot_cache", None)
    if cache is None:
        cache = g._bot_cache = {}
    key = ("ef_csc", efeat.data_ptr(), efeat._version, tuple(efeat.shape))
    if key not in cache:
        cache[key] = take_rows(efeat, g.csc.eid).contiguous()
    return cache[key]


def edge_mlp(g, efeat, W1, b1, W2):
    """`attn_edge_fc(relu(edge_encoder(efeat)))` of the ogbn-proteins layer (src/ogbn-proteins/models.py:244-248, :130-131)
    evaluated per edge in registers; returns the logit term [E,H] in CSC position order (use with
    `gat_attention(..., ee=..., ee_order="csc")`).  efeat [E,8] is an input (no gradient), W1 [16,8], b1 [16], W2 [H,16]."""
    return _EdgeMLP.apply(edge_features_csc(g, efeat), W1, b1, W2)


class _GatAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, el, er, ee, keep, slope, order, ee_csc=False, keep_csc=False):
        csc = g.csc
        H = _flat2(el if el is not None else ee).shape[1]
        el2 = None if el is None else el.reshape(-1, H)
        er2 = None if er is None else er.reshape(-1, H)
        ee2 = None if ee is None else ee.reshape(-1, H)
        aperm = csc.eid if order == "eid" else None
        # one permutation (`eperm`) serves both edge-indexed inputs, so they must be in the same order
        if keep is not None and ee is not None and keep_csc != ee_csc:
            if keep_csc:
                raise ValueError("gat_attention: a CSC-ordered keep mask needs CSC-ordered (or no) edge logits")
            keep = take_rows(keep, csc.eid).contiguous()
        in_csc = ee_csc if ee is not None else keep_csc
        eperm = csc.eid if ((ee is not None or keep is not None) and not in_csc) else None
        zs = _C.zsign_buffer(csc, H, slope)
        a = _C.gat_attn_fwd(csc, el2, er2, ee2, eperm, keep, slope, H, aperm, zs)
        ctx.g, ctx.slope, ctx.order, ctx.H, ctx.ee_csc = g, slope, order, H, ee_csc
        ctx.shapes = tuple(None if t is None else t.shape for t in (el, er, ee))
        ctx.zs = zs
        ctx.save_for_backward(el2, er2, ee2, a)
        return a.view(-1, H, 1)

    @staticmethod
    def backward(ctx, da):
        g, H = ctx.g, ctx.H
        csc, csr = g.csc, g.csr
        el2, er2, ee2, a = ctx.saved_tensors
        aperm = csc.eid if ctx.order == "eid" else None
        eperm = csc.eid if (ee2 is not None and not ctx.ee_csc) else None
        # dz is written in edge-id order when something edge-indexed consumes it, else CSC position order
        z_eid = ee2 is not None and not ctx.ee_csc
        dz, der = _C.gat_attn_bwd(csc, el2, er2, ee2, eperm, ctx.slope, H, a, da.reshape(-1, H), aperm,
                                  csc.eid if z_eid else None, er2 is not None, ctx.zs)
        d_el = d_er = d_ee = None
        if el2 is not None and ctx.needs_input_grad[1]:
            d_el = _C.segment_sum(csr, dz, csr.eid if z_eid else g.csr2csc).view(ctx.shapes[0])
        if er2 is not None and ctx.needs_input_grad[2]:
            d_er = der.view(ctx.shapes[1])
        if ee2 is not None and ctx.needs_input_grad[3]:
            d_ee = dz.view(ctx.shapes[2])
        return None, d_el, d_er, d_ee, None, None, None, None, None


def gat_attention(g, el=None, er=None, ee=None, *, keep=None, negative_slope=0.2, order="eid", ee_order="eid", keep_order="eid"):
    """Attention weights of one GAT layer in a single sweep over the in-edges (models.py:517-544):

        z_e = el[src] (+ er[dst]) (+ ee_e);  a = softmax over in-edges of leaky_relu(z, negative_slope)

    el, er: [N,H,1]; ee: [E,H,1] in edge-id order; keep: optional uint8 [E] in edge-id order — edges
    with keep == 0 are excluded from the softmax and get a == 0 (the edge-drop branch, models.py:528-539).
    `ee_order="csc"`: `ee` is already in CSC position order (what `edge_mlp` returns); `keep_order="csc"` likewise for the mask
    (`random_edge_keep` draws a uniformly random subset, so a layer may use its mask in whichever order is cheapest).
    Returns a [E,H,1] in edge-id order (order="eid") or CSC position order (order="csc")."""
    return _GatAttention.apply(g, el, er, ee, keep, float(negative_slope), order, ee_order == "csc", keep_order == "csc")


def edge_softmax(graph, logits, eids=None, norm_by="dst"):
    """`dgl.ops.edge_softmax(graph, logits, eids=ALL)` — models.py:544 and, with `eids`, :537.

    With `eids`, `logits` holds values for those edges only and the softmax runs over the
    edge-induced subgraph that keeps all nodes; the result has the shape and order of `logits`."""
    if norm_by != "dst":
        raise NotImplementedError("edge_softmax: only norm_by='dst' is on the reference's path")
    if eids is None or not torch.is_tensor(eids):
        return gat_attention(graph, None, None, logits, negative_slope=1.0, order="eid").view(logits.shape)
    E = graph.number_of_edges()
    full = logits.new_zeros((E,) + tuple(logits.shape[1:])).index_copy(0, eids, logits)
    keep = torch.zeros(E, dtype=torch.uint8, device=logits.device)
    keep[eids] = 1
    a = gat_attention(graph, None, None, full, keep=keep, negative_slope=1.0, order="eid")
    return take_rows(a.view(full.shape), eids)


# ------------------------------------------------------------------------------------------------ fused hidden-layer epilogue
def _dist_group(bn):
    """(do_sync, group) — global statistics when `bn` is a bot_amd.dist.SyncBatchNorm1d and a process group is up."""
    import torch.distributed as dist
    if getattr(bn, "_bot_sync", False) and dist.is_available() and dist.is_initialized() and dist.get_world_size(bn.group) > 1:
        return True, bn.group
    return False, None


def bn_batch_stats(x, bn, bn_training, halves_p=None, partials=None):
    """Column statistics for the fused epilogue: (mean, invstd, total_count, sync, group); updates bn's running statistics
    exactly like nn.BatchNorm1d (momentum, unbiased running variance).  In partitioned mode the (count, mean, M2) triples of
    the ranks are merged with Chan et al.'s pairwise formula through two small all-reduces.
    `halves_p` (the epilogue's dropout rate) asks for a sixth value: the fp16-halves scale of the epilogue's output, derived
    from the column extremes in the same pass (bot_bn_stats_halves_f32) — None where that call does not apply (partitioned
    statistics, eval mode)."""
    import torch.distributed as dist
    n = x.shape[0]
    sync, group = _dist_group(bn)
    total = float(n)
    extra = () if halves_p is None else (None,)
    if not bn_training:
        return (bn.running_mean, torch.rsqrt(bn.running_var + bn.eps), total, sync, group) + extra
    if not sync and (bn.momentum is not None or not bn.track_running_stats):
        # one GPU: statistics, invstd and the running-statistics update in ONE call (nine elementwise launches per layer before)
        track = bn.track_running_stats
        args = (x, bn.eps, bn.momentum if track else 0.0, bn.running_mean if track else None, bn.running_var if track else None,
                bn.num_batches_tracked if track else None)
        with torch.no_grad():
            if halves_p is not None and partials is not None:
                # the producer of x delivered the column partials with it (`stats_partials_for`): no pass over x
                part, minmax, pivot = partials
                mean, invstd, hscale = _C.bn_stats_halves_partials(part, minmax, pivot, n, *args[1:], bn.weight, bn.bias, halves_p)
                return mean, invstd, total, sync, group, hscale
            if halves_p is not None:
                mean, invstd, hscale = _C.bn_stats_halves(*args, bn.weight, bn.bias, halves_p)
                return mean, invstd, total, sync, group, hscale
            mean, invstd = _C.bn_stats(*args)
        return mean, invstd, total, sync, group
    mean, m2 = _C.colstats(x)
    if sync:
        # the global row count is a constant of the partition: ask for it once (one host sync), then never again, so the
        # host keeps running ahead of the GPU during the step
        totals = bn.__dict__.setdefault("_bot_total_rows", {})
        key = (id(group), dist.get_rank(group), n)   # per process group and local row count (one partition per rank and group)
        if key not in totals:
            cnt = mean.new_full((1,), float(n))
            dist.all_reduce(cnt, group=group)
            totals[key] = float(cnt.item())
        total = totals[key]
        gmean = mean * (n / total)
        dist.all_reduce(gmean, group=group)
        m2 = m2 + n * (mean - gmean) ** 2
        dist.all_reduce(m2, group=group)
        mean = gmean
    invstd = torch.rsqrt(m2 / total + bn.eps)
    if bn.track_running_stats:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            # nn.BatchNorm1d: momentum None = cumulative moving average, factor 1 / num_batches_tracked (one host read per
            # step in that rarely used mode)
            mom = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(m2 / max(total - 1.0, 1.0), alpha=mom)
    return (mean, invstd, total, sync, group) + extra


STATS_BYPRODUCT = os.environ.get("BOT_STATS_BYPRODUCT", "1") != "0"


def stats_partials_for(bn, bn_training, n_rows, F, device, halves_ok):
    """Buffers for a producer that can deliver BatchNorm's column partials as a by-product (the grouped NT GEMM's epilogue), or None when
    `bn_batch_stats` would not take them (eval mode, partitioned statistics, no halves epilogue): (part, minmax, pivot), all written by the
    producer.  The shift of a 256-row tile's sums is the tile's own FIRST value (ABI 19; round 5 shifted by zero, which lost
    (mean / std)^2 ulps of the variance on un-centred columns): a state-independent choice inside the data; the finish re-bases
    the tiles onto the first tile's pivot exactly, in double (csrc/dense.hip colstats_tiles_final_kernel)."""
    sync, _ = _dist_group(bn)
    if not (STATS_BYPRODUCT and bn_training and halves_ok and not sync and (bn.momentum is not None or not bn.track_running_stats)):
        return None
    tiles = (n_rows + 255) // 256
    return (torch.empty((tiles, 2, F), dtype=torch.float32, device=device), torch.empty((tiles, 2, F), dtype=torch.float32, device=device),
            torch.empty((tiles, F), dtype=torch.float32, device=device))


def random_edge_keep(graph, drop):
    """Edge-drop mask of one training step (models.py:528-532: `perm = randperm(E); eids = perm[int(E * drop):]`): uint8
    [E], 1 for a uniformly random subset of exactly E - int(E * drop) edges.  The seed comes from torch's CPU generator."""
    E = graph.number_of_edges()
    return _C.random_keep(E, E - int(E * drop), new_dropout_seed(1.0), graph.device)


def new_dropout_seed(p):
    """Seed of the Philox stream of one fused-dropout call, drawn from torch's CPU generator (so torch.manual_seed governs it)."""
    return int(torch.empty((), dtype=torch.int64).random_().item()) if p > 0 else 0


MODULAR_BN_LINK = os.environ.get("BOT_BN_LINK_MODULAR", "1") != "0"     # the modular epilogue (GCN, the edge-GAT stacks) takes part in gemm.BnLink too


class _BNActDrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu, p, bn_training, halves=False):
        from . import gemm
        seed = new_dropout_seed(p)
        piece = gemm.epilogue_piece(x.shape[1], x) if halves and bn_training else None
        hscale = None
        if piece is not None:
            mean, invstd, total, sync, group, hscale = bn_batch_stats(x, bn, bn_training, halves_p=p)
        else:
            mean, invstd, total, sync, group = bn_batch_stats(x, bn, bn_training)
        if hscale is not None:       # the next projection's fp16 halves written by this pass (bot_amd.gemm.take picks them up)
            order = gemm.left_order(piece)
            y, buf = _C.bn_act_fwd(x, mean, invstd, weight, bias, relu, p, seed, halves=(hscale, piece, 2 if order == 2 else 3))
            hv = gemm.Halves(buf, hscale, x.shape[0], x.shape[1], piece, order)
            # the projection that takes these halves can deliver this pass's backward reduce with the gradient it sends back (gemm.BnLink)
            hv.bn_link = ctx.out_link = gemm.BnLink(x, mean, invstd, weight, bias, p, seed, relu=relu) if (gemm.BN_BYPRODUCT and MODULAR_BN_LINK) else None
            gemm.stash(y, hv)
        else:
            ctx.out_link = None
            y = _C.bn_act_fwd(x, mean, invstd, weight, bias, relu, p, seed)
        ctx.save_for_backward(x, mean, invstd, weight, bias)
        ctx.cfg = (relu, p, seed, bn_training, sync, group, total)
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, mean, invstd, weight, bias = ctx.saved_tensors
        relu, p, seed, bn_training, sync, group, total = ctx.cfg
        dy = dy.contiguous()
        st = ctx.out_link.claim(dy) if ctx.out_link is not None else None
        sg, sgx = st.sums() if st is not None else _C.bn_act_bwd_reduce(dy, x, mean, invstd, weight, bias, relu, p, seed)
        dw = sgx if weight is not None and ctx.needs_input_grad[1] else None  # local sums: ranks' parameter grads are
        db = sg if bias is not None and ctx.needs_input_grad[2] else None     # summed later with all the others
        dx = None
        if ctx.needs_input_grad[0]:
            if bn_training:
                if sync:
                    both = torch.stack([sg, sgx])
                    dist.all_reduce(both, group=group)
                    sg, sgx = both[0].contiguous(), both[1].contiguous()
                dx = _C.bn_act_bwd_apply(dy, x, mean, invstd, weight, bias, relu, p, seed, sg, sgx, total)
            else:
                dx = _C.bn_act_bwd_apply(dy, x, mean, invstd, weight, bias, relu, p, seed, None, None, total)
        return dx, dw, db, None, None, None, None, None


class _AddBias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b):
        return x + b

    @staticmethod
    def backward(ctx, dy):
        db = None
        if ctx.needs_input_grad[1]:
            db = _C.colsum(dy)                 # column sums by the tree-reduction kernel of the BatchNorm statistics
        return dy, db


def add_bias(x, b):
    """x [N,F] + b [F] with the bias gradient computed by the column-statistics kernel."""
    return _AddBias.apply(x, b) if x.dim() == 2 and x.is_cuda == b.is_cuda else x + b


class _SumAll(torch.autograd.Function):
    """Sum of all entries of a float32 CUDA tensor through bot_colsum_f32: the tensor is viewed as [n / 64, 64], the column sums
    are taken by the two-stage tree kernel (finished in double, fixed order) and the <= 127 leftovers by a single-workgroup reduce.

    Why not `x.sum()`: torch's multi-workgroup reductions (a staging buffer + a semaphore zeroed by a memset in front of the kernel)
    give WRONG results inside a replayed hipGraph on this stack — measured in round 3 on the final bias gradient of the GAT stack
    (`dy.sum(0)` over [N, 40]): bit-identical to the eager step for the first replays, then off by ~1e-1 of the gradient on every
    further replay, identical in every run (tools/dbg_captured_eval.py); the same step with `bot_colsum_f32` matches the eager step bit for
    bit over any number of replays.  The train step (bot_amd.train / bot_amd.dist) therefore takes its N-sized reductions — the loss
    numerator, the weight count, bias gradients — through the library's own deterministic kernels."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        flat = x.reshape(-1)
        n64 = flat.numel() // 64 * 64
        parts = []
        if n64:
            parts.append(_C.colsum(flat[:n64].view(-1, 64)))
        if n64 < flat.numel():
            parts.append(flat[n64:])
        return torch.cat(parts).sum() if parts else flat.sum()

    @staticmethod
    def backward(ctx, g):
        return g.expand(ctx.shape)


def sum_all(x):
    """x.sum() with a fixed summation order that is also safe under hipGraph replay (see _SumAll); plain x.sum() off the GPU."""
    return _SumAll.apply(x) if (x.is_cuda and x.dtype == torch.float32 and x.numel() >= 4096) else x.sum()


SPLITK_MIN_ROWS = 1 << 20      # weight gradients over at least this many rows are reduced in row chunks
SPLITK_CHUNK_ROWS = 1 << 14


def weight_grad(dy, x):
    """dy^T x ([P,N] x [N,K]) — the weight gradient of a Linear, a reduction over all N rows.  From 2^20 rows on it is computed
    as N / 2^14 partial products over row chunks (ONE batched GEMM) summed afterwards: at S-products (N = 2.45 M, P = K = 480)
    the single GEMM's fp32 accumulation is 1.0e-4 of the largest entry away from the fp64 product, the chunked one 1.3e-6 to
    5e-6, and it is faster (8.7 vs 9.7 ms; tools/exp_splitk_dw.py).  The pre-BatchNorm gradients make this reduction
    cancellation-heavy (their column sums are zero), which is why the accumulation order shows."""
    n = x.shape[0]
    if n < SPLITK_MIN_ROWS or not dy.is_contiguous() or not x.is_contiguous():
        return dy.t() @ x
    S = n // SPLITK_CHUNK_ROWS
    R = n // S
    dw = torch.bmm(dy[:S * R].view(S, R, -1).transpose(1, 2), x[:S * R].view(S, R, -1)).sum(0)
    if S * R < n:
        dw = dw + dy[S * R:].t() @ x[S * R:]
    return dw


class _MatmulT(torch.autograd.Function):
    """y = x W^T with the weight gradient through `weight_grad`."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return torch.mm(x, w.t())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = torch.mm(dy, w) if ctx.needs_input_grad[0] else None
        dw = weight_grad(dy.contiguous(), x) if ctx.needs_input_grad[1] else None
        return dx, dw


def linear(x, weight, bias=None):
    """`F.linear` for [N, K] inputs whose two all-row reductions are done carefully: the bias gradient (a column sum over N
    rows) runs on the colstats kernel — the stock reduction takes 19 ms for [2 449 029, 47] (S-products classifier,
    ogbn-products/models.py:262), longer than the classifier GEMMs — and the weight gradient goes through `weight_grad`."""
    from . import gemm
    y = gemm.linear(x, weight)                                          # on the fp16 matrix cores where the shapes pay
    if y is None:
        y = _MatmulT.apply(x, weight) if x.shape[0] >= SPLITK_MIN_ROWS and torch.is_grad_enabled() else torch.mm(x, weight.t())
    return y if bias is None else _AddBias.apply(y, bias)


def bn_relu_dropout(x, bn, *, relu=True, p=0.0, training=False, halves=False):
    """`dropout(relu(bn(x)))` over the node axis in 2 reads + 1 write (models.py:636-639, :726-731).

    `bn` is the layer's nn.BatchNorm1d (its parameters, running statistics and train/eval state are used and
    updated exactly as nn.BatchNorm1d would); `p` is the dropout rate, applied when `training`."""
    w = bn.weight if bn.affine else None
    b = bn.bias if bn.affine else None
    bn_training = bn.training or not bn.track_running_stats
    return _BNActDrop.apply(x, w, b, bn, bool(relu), float(p) if training else 0.0, bn_training, bool(halves))
