"""One autograd node for a hidden GAT layer of the reference stack — `GATConv.forward` (models.py:475-566) followed by
`flatten -> BatchNorm1d -> relu -> dropout` (models.py:725-731) — arranged so that nothing of size N*H*D is touched
more often than it has to be:

  forward   ONE GEMM  h @ [W_fc ; W_res ; W_fc^T-folded attn_l ; attn_r]^T  ->  [ft | res | el | er]  (row-padded to x128:
            the extra columns ride in tiles the GEMM pads to anyway), fused attention kernel, SpMM that reads `ft` in place
            (strided slab) and adds `res` in its epilogue, column statistics, BatchNorm+ReLU+dropout in one pass.
  backward  BN/ReLU/dropout backward writes dx straight into the `res` columns of the [N, P] gradient buffer; the fused
            spmm_dot kernel (one gather) reads dx there and writes d ft into the `fc` columns; d el / d er go into theirs;
            ONE GEMM each for dW_cat and dh.

Versus the modular path this drops the skinny [N,K]x[K,H] score GEMMs, the separate residual GEMMs/adds and every
slice/cat copy: 29.5 -> ~25 ms per step at BASELINE config 2.  Used by `bot_amd.nn.GAT` when the layer's options allow
(`can_fuse`); the modular path (bot_amd.nn.GATConv + ops.bn_relu_dropout) stays the general fallback, and both are held to
the same golden vectors.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _C, gemm, halo, side
from ..graph import take_rows
from ..ops import bn_batch_stats, new_dropout_seed, stats_partials_for


ABSMAX_BYPRODUCT = os.environ.get("BOT_ABSMAX_BYPRODUCT", "1") != "0"   # max|gradient| from its producers instead of a pass (_GATHidden.backward)
SKIP_Y = os.environ.get("BOT_SKIP_Y", "1") != "0"   # hidden states whose one consumer is a halves GEMM exist as halves only (_epilogue_forward)
HANDLES = 0    # epilogues that returned a handle instead of a stored fp32 tensor (tests)
SKINNY = os.environ.get("BOT_SKINNY", "1") != "0"   # the small-K products of the aggregate-first layer on bot_skinny_gemm_f32
FORCE = False  # tests set this to run the fused node over the emulated (CPU) backend
CALLS = 0      # number of fused-layer invocations (tests assert the path was actually taken)
AGG_CALLS = 0  # ... of which aggregate-before-project
L0_CALLS = 0   # ... of which on the grouped-halves kernels (_l0_halves_ok)


def can_fuse(conv, norm, activation, graph, training, stack_residual) -> bool:
    """`norm` None = the layer has no BatchNorm/ReLU/dropout epilogue (the stack's output layer)."""
    epilogue_ok = norm is None or (isinstance(norm, nn.BatchNorm1d) and activation in (F.relu, torch.relu) and not stack_residual)
    return (epilogue_ok and hasattr(conv, "fc") and not (conv._use_symmetric_norm and graph.halo is not None)
            and conv._activation is None
            and not (training and (conv.edge_drop > 0 or conv.feat_drop.p > 0))
            and (graph.halo is not None or not graph.is_block) and conv._out_feats <= 256)


def sym_scales(graph):
    """Symmetric normalisation of the attention (models.py:500-505, :550-555: `feat_src *= out_deg^-1/2` before the logits,
    `rst *= in_deg^+1/2` after the aggregation) folded into the EDGE WEIGHTS, so the layer node needs no pass over [N,H,D]:
        rst[v] = s_in[v] * sum_e a_e * (s_out[u] ft[u])  =  sum_e (a_e * s_out[u_e] * s_in[v_e]) ft[u]
    Returns (s_out [N] for the logits `el`, w_e [E] = s_out[src] * s_in[dst] in CSC position order); static per graph."""
    from . import _graph_cache, degree_norm
    c = _graph_cache(graph)
    if "sym_scales" not in c:
        csc = graph.csc
        s_out, s_in = degree_norm(graph, "out", -0.5), degree_norm(graph, "in", 0.5)
        deg = (csc.indptr[1:] - csc.indptr[:-1]).long()
        dst = torch.repeat_interleave(torch.arange(csc.n_rows, device=deg.device), deg)
        c["sym_scales"] = (s_out, (take_rows(s_out, csc.indices) * s_in[dst]).unsqueeze(1).contiguous())
    return c["sym_scales"]


def block_width(HD: int) -> int:
    """Column width of the `ft` and `res` blocks of the merged projection [ft | res | el | er | pad]: H*D rounded up to a multiple of 4,
    so that the residual block — whose rows the fused backward sweep GATHERS — starts on a 16-byte boundary (3 x 250: 752)."""
    return (HD + 3) // 4 * 4


def cat_weight(conv):
    """[W_fc ; W_res ; wl ; wr ; 0-pad] with wl[h] = W_fc[h]^T attn_l[h]  (so that h @ wl^T == <fc(h)[h], attn_l[h]>,
    models.py:517,521).  Differentiable: the gradients reach fc.weight, res_fc.weight, attn_l, attn_r through these ops."""
    H, D = conv._num_heads, conv._out_feats
    W = conv.fc.weight
    Wh = W.view(H, D, -1)
    gap = block_width(H * D) - H * D                 # zero rows behind each of the two copied blocks (block_width)
    rows = [W] + ([W.new_zeros(gap, W.shape[1])] if gap else [])
    if conv.res_fc is not None:
        rows += [conv.res_fc.weight] + ([W.new_zeros(gap, W.shape[1])] if gap else [])
    rows.append((Wh * conv.attn_l.view(H, D, 1)).sum(1))
    if conv.attn_r is not None:
        rows.append((Wh * conv.attn_r.view(H, D, 1)).sum(1))
    used = sum(r.shape[0] for r in rows)
    pad = (-used) % 128
    if pad:
        rows.append(W.new_zeros(pad, W.shape[1]))
    return torch.cat(rows)


class _MergeWeight(torch.autograd.Function):
    """`_kp(cat_weight(conv))` / `_kp(cat_weight_aggfirst(conv))` in one launch forward and two backward (merge.hip) instead of
    ~11 + ~11 stock launches per layer and step.  Same columns, [K, P] layout."""

    @staticmethod
    def forward(ctx, W, Wres, attn_l, attn_r, H, D, with_fc):
        blk = block_width(H * D) if with_fc else H * D      # the aggregate-first form has no `ft` block: `res` starts at column 0
        used = (blk if with_fc else 0) + (blk if Wres is not None else 0) + H + (H if attn_r is not None else 0)
        P = used + (-used) % 128
        ctx.save_for_backward(W, attn_l, attn_r)
        ctx.wres_ref = Wres                     # (only its .grad is looked at, backward)
        ctx.cfg = (H, D, P, with_fc, Wres is not None, blk)
        return _C.merge_weight_fwd(W, Wres, attn_l.reshape(-1), None if attn_r is None else attn_r.reshape(-1), H, D, P, with_fc, block=blk)

    @staticmethod
    def backward(ctx, dm):
        W, attn_l, attn_r = ctx.saved_tensors
        H, D, P, with_fc, has_res, blk = ctx.cfg

        def body():
            return _C.merge_weight_bwd(W, attn_l.reshape(-1), None if attn_r is None else attn_r.reshape(-1), H, D, P, with_fc, has_res, dm, block=blk)
        if side.produced(dm):
            # the merged gradient came from the side stream (bot_amd.side): its split runs there too when autograd will only STEAL the
            # results (each parameter's one contribution, no .grad yet); otherwise the main stream joins first
            # (a tensor hook or post-accumulate-grad hook - gradient clipping, a DDP-style reducer - would touch the in-flight
            # gradient on the main stream before the end-of-backward join: with any hook registered the main stream joins first)
            if with_fc and side.MERGE_ON_SIDE and side.usable(dm) and all(p is None or (p.grad is None and side.unhooked(p))
                                                                          for p in (W, ctx.wres_ref, attn_l, attn_r)):
                dW, dWres, dal, dar = side.run(body, dm, W, attn_l, attn_r)
            else:
                side.join()
                dW, dWres, dal, dar = body()
        else:
            dW, dWres, dal, dar = body()
        return dW, dWres, dal.view_as(attn_l), (None if dar is None else dar.view_as(attn_r)), None, None, None


def merged_weight(conv, with_fc=True):
    """The layer's merged projection weight in the layout the GEMMs want.  On the GPU (and [K, P] layout) by the merge kernels;
    otherwise (CPU test backend) by the tensor-op definition `cat_weight` / `cat_weight_aggfirst`."""
    W = conv.fc.weight
    if W.is_cuda and WEIGHT_KP:
        return _MergeWeight.apply(W, conv.res_fc.weight if conv.res_fc is not None else None, conv.attn_l,
                                  conv.attn_r if isinstance(conv.attn_r, torch.Tensor) else None, conv._num_heads, conv._out_feats,
                                  with_fc)
    return _kp(cat_weight(conv) if with_fc else cat_weight_aggfirst(conv))


def _ext_width(HD, H):
    """Row width of the halo-extended table [ft | el | pad]: keeps the slab's vector alignment."""
    return HD + (H + 3) // 4 * 4


def _extend_forward(graph, out, HD, H, c):
    """Partitioned mode: build the source table [n_ext, W] = [ft | el] for owned + halo rows.  Owned rows are copied out of
    the GEMM output, the rows other ranks need are packed by the row-gather kernel and exchanged with ONE all-to-all."""
    import torch.distributed as dist
    plan = graph.halo
    n_own = out.shape[0]
    W = _ext_width(HD, H)
    ext = torch.empty((n_own + plan.n_halo, W), dtype=out.dtype, device=out.device)
    ext[:n_own, :HD] = out[:, :HD]
    ext[:n_own, HD:HD + H] = out[:, c:c + H]
    if W > HD + H:
        ext[:n_own, HD + H:].zero_()
    send = _C.gather_rows(ext[:n_own], plan.send_rows) if plan.n_send else ext.new_empty((0, W))
    halo.a2a(ext[n_own:], send, plan.recv_splits, plan.send_splits, plan.group)
    return ext


def _extend_backward(graph, dext, n_own):
    """Reverse exchange: halo-row gradients go back to their owners and are added into the owned rows, peer by peer in
    rank order (each peer's rows are sorted-unique: one writer per row, deterministic)."""
    import torch.distributed as dist
    plan = graph.halo
    back = torch.empty((plan.n_send, dext.shape[1]), dtype=dext.dtype, device=dext.device)
    halo.a2a(back, dext[n_own:], plan.send_splits, plan.recv_splits, plan.group)
    own = dext[:n_own]
    off = 0
    for cnt in plan.send_splits:
        if cnt:
            _C.scatter_add_rows(own, plan.send_rows[off:off + cnt], back[off:off + cnt])
        off += cnt
    return own


# Partitioned mode, overlapped form (round 3).  The merged-GEMM layers (`_GATHidden`) ship `el` of the halo rows in a small exchange,
# start the big one ([n_halo, H*D] projected rows) ASYNCHRONOUSLY, and meanwhile run everything that does not need it: the attention
# (scores only) and the aggregation over the in-edges whose source is owned; the halo-source edges follow into the same rows once the
# transfer has landed (Graph.halo_split).  Backward likewise: the halo rows' gradients are swept first and sent while the owned rows
# are swept and the attention backward runs.  The exchange helpers and the switch (BOT_HALO_OVERLAP / `halo.OVERLAP`) live in
# bot_amd/halo.py, which gives the modular layers the same form.
OVERLAP_CALLS = 0   # layer forwards that took the overlapped form (tests assert the path was taken)
_ship_rows, _return_rows, _fold_back = halo.ship_rows, halo.return_rows, halo.fold_back


def _epilogue_forward(x, bn, bn_w, bn_b, bn_training, drop_p, y_needed=True, partials=None):
    """BatchNorm statistics + the fused BatchNorm / ReLU / dropout pass.  When the next projection runs on fp16 halves
    (bot_amd.gemm) the pass writes them too and the scale comes from the statistics pass: y is not read again before its GEMM.
    y_needed=False (the caller KNOWS the one consumer of y is such a projection, `halves_only_consumer`): the fp32 y is not stored at
    all — 508 MB per hidden layer at config 2 that nobody would read (this pass's backward reads x) — and the returned tensor is a
    HANDLE: zeros of y's shape on ONE element (stride 0), which carries the autograd edge and the key of the stashed halves.
    `gemm.take` refuses a handle without its halves, so a handle can never be split into a GEMM operand by mistake."""
    HD = x.shape[1]
    piece = gemm.epilogue_piece(HD, x) if bn_training else None
    seed = new_dropout_seed(drop_p)
    if piece is not None:
        # `partials`: the kernel that wrote x delivered BatchNorm's column partials with it (no statistics pass over x)
        mean, invstd, total, sync, group, hscale = bn_batch_stats(x, bn, bn_training, halves_p=drop_p, partials=partials)
        if hscale is not None:
            order = gemm.left_order(piece)
            y, buf = _C.bn_act_fwd(x, mean, invstd, bn_w, bn_b, True, drop_p, seed, halves=(hscale, piece, 2 if order == 2 else 3), want_y=y_needed)
            if y is None:
                global HANDLES
                HANDLES += 1
                y = gemm.make_handle(x, x.shape[0], HD)
            hv = gemm.Halves(buf, hscale, x.shape[0], HD, piece, order)
            # the consumer's backward can deliver this epilogue's reduce pass with the gradient it sends back (gemm.BnLink)
            hv.bn_link = link = gemm.BnLink(x, mean, invstd, bn_w, bn_b, drop_p, seed) if gemm.BN_BYPRODUCT else None
            gemm.stash(y, hv)
            return y, mean, invstd, total, sync, group, seed, link
    else:
        mean, invstd, total, sync, group = bn_batch_stats(x, bn, bn_training)
    return _C.bn_act_fwd(x, mean, invstd, bn_w, bn_b, True, drop_p, seed), mean, invstd, total, sync, group, seed, None


def _bwd_reduce(ctx, dy, x, mean, invstd, bn_w, bn_b, drop_p, seed, want_max=False):
    """(sum_g, sum_gx, maxima) of the epilogue's backward: from the partials the consumer layer's `d h` product delivered with `dy`
    (gemm.BnLink, include/bot_gnn.h "v18") when there are any for exactly this tensor, else by the reduce pass.  maxima: what `_bwd_bound`
    reads (None unless want_max or delivered)."""
    link = getattr(ctx, "out_link", None)
    st = link.claim(dy) if link is not None else None
    if st is not None:
        sg, sgx = st.sums()
        return sg, sgx, st
    if want_max:
        return _C.bn_act_bwd_reduce(dy, x, mean, invstd, bn_w, bn_b, True, drop_p, seed, want_max=True)
    sg, sgx = _C.bn_act_bwd_reduce(dy, x, mean, invstd, bn_w, bn_b, True, drop_p, seed)
    return sg, sgx, None


def _bwd_reduce_bound(ctx, dy, x, mean, invstd, bn_w, bn_b, drop_p, seed, bn_training, total, slots):
    """`_bwd_reduce(want_max=True)` + `_bwd_bound` for one rank (no cross-rank reduction of the sums in between): delivered partials are
    finished in ONE launch (bot_bn_bwd_partials_finish_f32).  -> (sum_g, sum_gx), the bound folded into `slots`."""
    link = getattr(ctx, "out_link", None)
    st = link.claim(dy) if link is not None else None
    if st is not None:
        return st.finish(bn_training, total, slots)
    sg, sgx, ws = _C.bn_act_bwd_reduce(dy, x, mean, invstd, bn_w, bn_b, True, drop_p, seed, want_max=True)
    _C.bn_bwd_bound(ws, dy.shape[0], sg if bn_training else None, sgx if bn_training else None, total, bn_w, invstd, slots)
    return sg, sgx


def _bwd_bound(mx, n, sg, sgx, total, bn_w, invstd, slots):
    if isinstance(mx, _C.BnBwdStats):
        return mx.bound(sg, sgx, total, slots)
    return _C.bn_bwd_bound(mx, n, sg, sgx, total, bn_w, invstd, slots)


# A hidden layer's gradient operand without a split pass (ABI 17, include/bot_gnn.h "v17"): the BatchNorm backward and the transposed sweep
# write their column blocks of [d ft | d res | d el | d er | 0] as halves under a BOUNDED scale, the attention columns follow under a
# second scale; no fp32 [N, P] buffer, no halves_split pass (1 GB read + 1 GB written per hidden layer at config 2).
DOUT_DIRECT = os.environ.get("BOT_DOUT_DIRECT", "1") != "0"
DOUT_DIRECT_CALLS = 0
TAIL_CAP = 2.0 ** 20     # the attention columns' scale may be at most this much finer than the big blocks' (the GEMM rescales its accumulators by the ratio)


def rowsum_bound(graph, attn_p: float) -> float:
    """max over the sources u of sum_{e out of u} a_d[e, h] is at most (largest out-degree) / (1 - attn_p): every attention weight is
    <= 1 (a softmax over the in-edges of its destination) and attention dropout rescales the kept ones by 1 / (1 - p).  Static per graph
    (one host read, cached like the zero-in-degree check)."""
    from . import _graph_cache
    c = _graph_cache(graph)
    if "max_out_deg" not in c:
        ip = graph.csr.indptr
        c["max_out_deg"] = int((ip[1:] - ip[:-1]).max()) if ip.numel() > 1 else 0
    return max(1.0, c["max_out_deg"] / (1.0 - attn_p))


def _dout_direct_ok(ctx, g, h, H, D, P, B, has_res, epi) -> bool:
    HD = H * D
    c = 2 * B
    return (DOUT_DIRECT and ctx.halves is not None and has_res and epi is not None and not ctx.overlap and g.halo is None and not ctx.sym
            and HD % 2 == 0 and H >= 2 and c % 32 == 0 and P % 64 == 0 and gemm.left_order(P) == 2
            and gemm.NT_KERNEL == "halves3" and gemm.TN_KERNEL == "halves3" and (ctx.halves[2] * P >= gemm.TN_MIN_OUT or gemm.FORCE or FORCE) and not epi[3])


class _GATHidden(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, Wcat, bn_w, bn_b, graph, bn, H, D, has_res, has_er, slope, attn_p, drop_p, bn_training, kp, sym, y_needed=True):
        N, HD = h.shape[0], H * D
        csc = graph.csc
        ctx.sym = sym                                                   # symmetric normalisation folded into the edge weights
        ctx.kp = kp                                                     # Wcat is [K, P] (see WEIGHT_KP) instead of [P, K]
        xh = None
        if gemm.enabled(h):                                             # fp32 GEMM on the fp16 matrix cores (bot_amd.gemm)
            xh = gemm.take(h, 0)                                        # written by the previous layer's epilogue, or split here
            ws = gemm.split_right(Wcat.t().contiguous() if kp else Wcat)
            ctx.wscale = ws.scale                                       # the backward splits the transpose: same entries, same scale
            out = gemm.mm_nt(xh, ws)
        else:
            out = torch.mm(h, Wcat) if kp else torch.mm(h, Wcat.t())    # [N, P] = [ft | res | el | er | pad]
        ctx.halves = None if xh is None else (xh.n, xh.F, xh.piece, xh.order)
        ctx.bn_link = None if xh is None else xh.bn_link                # the epilogue that wrote h (its backward's reduce pass rides on `d h`)
        ctx.out_link = None
        B = block_width(HD)                                             # [ft (HD) pad -> B | res (HD) pad -> B | el | er | pad]
        c = 2 * B if has_res else B
        ext = None
        ctx.overlap = halo.enabled(graph) and not sym
        if ctx.overlap:                                                 # partitioned, overlapped: see OVERLAP above
            global OVERLAP_CALLS
            OVERLAP_CALLS += 1
            plan, sp = graph.halo, graph.halo_split
            el_own = out[:, c:c + H].contiguous()
            el_halo, _, _ = _ship_rows(plan, el_own)                    # small: the scores of the halo rows, needed first
            ft_halo, work, send_keep = _ship_rows(plan, out[:, :HD], async_op=True)   # big: in flight from here
            el = torch.cat([el_own, el_halo])
            er = out[:, c + H:c + 2 * H].contiguous() if has_er else None
            ctx.zs = _C.zsign_buffer(csc, H, slope)
            ctx.adrop = (attn_p, new_dropout_seed(attn_p)) if attn_p > 0 else None
            a, a_d = _C.gat_attn_fwd(csc, el, er, None, None, None, slope, H, None, ctx.zs, drop=ctx.adrop or (0.0, 0))
            res = out[:, B:B + HD].unflatten(1, (H, D)) if has_res else None
            x3 = _C.spmm(sp["csc_own"], out[:, :HD].unflatten(1, (H, D)), a_d, sp["csc_own_pos"], addend=res)   # owned-source edges
            work.wait()
            del send_keep
            if sp["csc_halo"].nnz:                                      # halo-source edges into the same rows (in place)
                _C.spmm(sp["csc_halo"], ft_halo.unflatten(1, (H, D)), a_d, sp["csc_halo_pos"], out=x3, addend=x3)
            x = x3.view(N, HD)
            ctx.graph = graph
            keep = (h if xh is None else xh.buf, Wcat, out, el, er, a, a_d, ft_halo)
            if xh is not None:
                ctx.xscale = xh.scale
            if bn is None:
                ctx.save_for_backward(*keep)
                ctx.cfg = (H, D, has_res, has_er, slope, None)
                return x
            y, mean, invstd, total, sync, group, seed, ctx.out_link = _epilogue_forward(x, bn, bn_w, bn_b, bn_training, drop_p, y_needed)
            ctx.save_for_backward(*keep, x, mean, invstd, bn_w, bn_b)
            ctx.cfg = (H, D, has_res, has_er, slope, (drop_p, seed, bn_training, sync, group, total))
            return y
        if graph.halo is not None:                                      # partitioned: owned + halo source rows
            ext = _extend_forward(graph, out, HD, H, c)
            ft = ext[:, :HD].unflatten(1, (H, D))
            el = ext[:, HD:HD + H].contiguous()
        else:
            ft = out[:, :HD].unflatten(1, (H, D))
            el = out[:, c:c + H].contiguous()
        er = out[:, c + H:c + 2 * H].contiguous() if has_er else None
        if sym:
            s_out, w_e = sym_scales(graph)
            el = el * s_out.unsqueeze(1)                                # logits see the scaled projection (models.py:505, :517)
        ctx.zs = _C.zsign_buffer(csc, H, slope)
        # nn.Dropout on the attention weights (models.py:544) inside the attention kernel: a_d = a * keep / (1 - p), Philox mask
        ctx.adrop = (attn_p, new_dropout_seed(attn_p)) if attn_p > 0 else None
        a, a_d = _C.gat_attn_fwd(csc, el, er, None, None, None, slope, H, None, ctx.zs, drop=ctx.adrop or (0.0, 0))
        if sym:
            a_d = a_d * w_e                                             # the SpMM weights; d a below is scaled back by w_e
        res = out[:, B:B + HD].unflatten(1, (H, D)) if has_res else None
        if HD % 4 and bn is not None and sweep_is_row_kernel(graph, H, D):
            # rows of 750 floats: give the pre-BatchNorm tensor a row pitch of 752 so that its rows are 16-byte aligned — the
            # BatchNorm kernels read it four times per step and move 16-byte aligned operands at full width (dense.hip)
            xbuf = torch.empty((N, (HD + 3) // 4 * 4), dtype=out.dtype, device=out.device)
            x = xbuf[:, :HD]
            _C.spmm(csc, ft, a_d, None, out=x.unflatten(1, (H, D)), addend=res)
        else:
            x = _C.spmm(csc, ft, a_d, None, addend=res).view(N, HD)     # aggregation + residual (models.py:547-560)
        ctx.graph = graph
        # the weight gradient needs the layer input: its fp16 halves when the GEMMs run on them (h itself is not kept then)
        keep = (h if xh is None else xh.buf, Wcat, ext if ext is not None else out, el, er, a, a_d)
        if xh is not None:
            ctx.xscale = xh.scale
        if bn is None:                                                  # output layer: no epilogue
            ctx.save_for_backward(*keep)
            ctx.cfg = (H, D, has_res, has_er, slope, None)
            return x
        y, mean, invstd, total, sync, group, seed, ctx.out_link = _epilogue_forward(x, bn, bn_w, bn_b, bn_training, drop_p, y_needed)
        ctx.save_for_backward(*keep, x, mean, invstd, bn_w, bn_b)
        ctx.cfg = (H, D, has_res, has_er, slope, (drop_p, seed, bn_training, sync, group, total))
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        H, D, has_res, has_er, slope, epi = ctx.cfg
        g = ctx.graph
        dy = dy.contiguous()
        d_bn_w = d_bn_b = None
        ft_halo = None
        saved = ctx.saved_tensors
        if ctx.overlap:
            ft_halo, saved = saved[7], saved[:7] + saved[8:]
        if epi is None:
            h, Wcat, table, el, er, a, a_d = saved
        else:
            h, Wcat, table, el, er, a, a_d, x, mean, invstd, bn_w, bn_b = saved
            drop_p, seed, bn_training, sync, group, total = epi
        kp = ctx.kp
        N, HD, P = h.shape[0], H * D, Wcat.shape[1 if kp else 0]
        B = block_width(HD)
        if _dout_direct_ok(ctx, g, h, H, D, P, B, has_res, epi):
            out = _GATHidden._backward_direct(ctx, dy, g, h, Wcat, table, el, er, a, a_d, x, mean, invstd, bn_w, bn_b, epi, H, D, P, B, has_er, slope, kp)
            if out is not None:
                return out
        dout = torch.empty((N, P), dtype=dy.dtype, device=h.device)
        dx = dout[:, B:B + HD] if has_res else torch.empty((N, HD), dtype=dy.dtype, device=h.device)
        slots = None
        if B != HD:                                                     # the pad columns meet zero weight rows: they must be finite
            dout[:, HD:B].zero_()
            if has_res:
                dout[:, B + HD:2 * B].zero_()
        if epi is None:
            dx.copy_(dy)
        else:
            sg, sgx, _ = _bwd_reduce(ctx, dy, x, mean, invstd, bn_w, bn_b, drop_p, seed)
            d_bn_w, d_bn_b = sgx, sg                                     # local sums (ranks' parameter grads are summed later)
            if bn_training and sync:
                both = torch.stack([sg, sgx])
                dist.all_reduce(both, group=group)
                sg, sgx = both[0].contiguous(), both[1].contiguous()
            # the gradient buffer becomes a halves-GEMM operand below: its two big producers (this pass: dx, the fused sweep: d ft)
            # deliver max|value| as they write, instead of a separate 1 GB pass (include/bot_gnn.h "Maxima as by-products")
            if ctx.halves is not None and has_res and not ctx.overlap and g.halo is None and ABSMAX_BYPRODUCT:
                slots = _C.absmax_slots(dy.device)
            _C.bn_act_bwd_apply(dy, x, mean, invstd, bn_w, bn_b, True, drop_p, seed, sg if bn_training else None,
                                sgx if bn_training else None, total, out=dx, absmax=slots)
        c = 2 * B if has_res else B
        if ctx.overlap:
            # halo rows first — their gradients travel back while the owned rows are swept and the attention backward runs
            plan, sp = g.halo, g.halo_split
            dx3 = dx.unflatten(1, (H, D))
            da = torch.empty((g.csr.nnz, H), dtype=dy.dtype, device=h.device)
            dft_halo = torch.empty((plan.n_halo, HD), dtype=dy.dtype, device=h.device)
            if sp["csr_halo"].nnz:
                _C.spmm_dot(sp["csr_halo"], dx3, a_d, sp["csr_halo_c2c"], ft_halo.unflatten(1, (H, D)), out=dft_halo.unflatten(1, (H, D)), dot=da)
            elif plan.n_halo:
                dft_halo.zero_()
            back_ft, work = _return_rows(plan, dft_halo, async_op=True)
            _C.spmm_dot(sp["csr_own"], dx3, a_d, sp["csr_own_c2c"], table[:, :HD].unflatten(1, (H, D)),
                        out=dout[:, :HD].unflatten(1, (H, D)), dot=da)
            dz, der = _C.gat_attn_bwd(g.csc, el, er, None, None, slope, H, a, da, None, None, has_er, ctx.zs, drop=ctx.adrop)
            d_el = _C.segment_sum(g.csr, dz, g.csr2csc)                 # [n_own + n_halo, H]
            back_el, _ = _return_rows(plan, d_el[N:].contiguous())      # small, synchronous
            work.wait()
            _fold_back(plan, dout[:, :HD], back_ft)
            dout[:, c:c + H] = _fold_back(plan, d_el[:N].contiguous(), back_el)
            halo, skip_sweep = False, True
        else:
            halo, skip_sweep = g.halo is not None, False
        ft = table[:, :HD].unflatten(1, (H, D))
        if skip_sweep:
            pass
        elif halo:                                                      # gradients of the extended table [d ft | d el]
            dext = torch.empty_like(table)
            dft_dst = dext[:, :HD].unflatten(1, (H, D))
        else:
            dft_dst = dout[:, :HD].unflatten(1, (H, D))
        if not skip_sweep:
            _, da = _C.spmm_dot(g.csr, dx.unflatten(1, (H, D)), a_d, g.csr2csc, ft, out=dft_dst, absmax=slots)
            if ctx.sym:
                s_out, w_e = sym_scales(g)
                da = da * w_e
            dz, der = _C.gat_attn_bwd(g.csc, el, er, None, None, slope, H, a, da, None, None, has_er, ctx.zs, drop=ctx.adrop)
            d_el = _C.segment_sum(g.csr, dz, g.csr2csc)
            if ctx.sym:
                d_el = d_el * s_out.unsqueeze(1)
        if skip_sweep:
            pass
        elif halo:
            dext[:, HD:HD + H] = d_el
            if dext.shape[1] > HD + H:
                dext[:, HD + H:].zero_()
            own = _extend_backward(g, dext, N)
            dout[:, :HD] = own[:, :HD]
            dout[:, c:c + H] = own[:, HD:HD + H]
        else:
            dout[:, c:c + H] = d_el
        if has_er:
            dout[:, c + H:c + 2 * H] = der
        used = c + (2 * H if has_er else H)
        if used < P:
            dout[:, used:].zero_()
        dW = dh = None
        if ctx.halves is not None:
            xh = gemm.Halves(h, ctx.xscale, *ctx.halves)
            if slots is not None:                                        # + the few score columns no big producer covers
                _C.absmax_into(dout[:, c:used], slots)
                dh_ = gemm.split(dout, 0, scale=_C.halves_scale_from_slots(slots))
            else:
                dh_ = gemm.split(dout, 0)
            if ctx.needs_input_grad[0]:
                dh = gemm.mm_nt(dh_, gemm.split_right(Wcat if kp else Wcat.t().contiguous(), scale=ctx.wscale), link=ctx.bn_link)
            if ctx.needs_input_grad[1]:
                def wgrad():
                    dW = gemm.tn(xh, dh_)                                # [K, P]
                    return dW if kp else dW.t().contiguous()
                # only the optimizer needs it: on the side stream (bot_amd.side) it runs beside the previous layer's BatchNorm / sparse
                # backward, which leave the matrix cores idle.  Issued BEHIND the input gradient's launch: two GEMM workgroups do not fit
                # one CU's LDS, so the reduction starts when that product ends and the main stream has moved on to HBM-bound kernels
                dW = side.run(wgrad, xh, dh_) if (side.usable(dout) and xh.piece * dh_.piece >= side.MIN_OUT) else wgrad()
        else:
            if ctx.needs_input_grad[1]:
                dW = torch.mm(h.t(), dout) if kp else torch.mm(dout.t(), h)
            if ctx.needs_input_grad[0]:
                dh = torch.mm(dout, Wcat.t()) if kp else torch.mm(dout, Wcat)
        return (dh, dW, d_bn_w if ctx.needs_input_grad[2] else None, d_bn_b if ctx.needs_input_grad[3] else None,
                None, None, None, None, None, None, None, None, None, None, None, None, None)


def _backward_direct(ctx, dy, g, h, Wcat, table, el, er, a, a_d, x, mean, invstd, bn_w, bn_b, epi, H, D, P, B, has_er, slope, kp):
    """_GATHidden.backward with the gradient operand written by its producers (see DOUT_DIRECT).  None: the sweep's all-heads layout does
    not cover this shape - the caller takes the fp32 form."""
    global DOUT_DIRECT_CALLS
    drop_p, seed, bn_training, sync, group, total = epi
    N, HD = h.shape[0], H * D
    c = 2 * B
    used = c + (2 * H if has_er else H)
    piece = P                                           # the operand [h1 | 2^11 h2]: second half `piece` columns behind the first
    buf = torch.empty((N, 2 * piece), dtype=torch.float16, device=h.device)
    dxb = torch.empty((N, B), dtype=dy.dtype, device=h.device)          # fp32 d res beside the halves: the sweep gathers its rows (16-byte aligned pitch)
    dx = dxb[:, :HD]
    ft = table[:, :HD].unflatten(1, (H, D))
    if not _C.spmm_dot_halves_fits(dx.unflatten(1, (H, D)), ft, buf, D, piece):
        return None
    DOUT_DIRECT_CALLS += 1
    slots = _C.absmax_slots(dy.device)
    sg, sgx = _bwd_reduce_bound(ctx, dy, x, mean, invstd, bn_w, bn_b, drop_p, seed, bn_training, total, slots)      # (no sync statistics here: _dout_direct_ok)
    d_bn_w, d_bn_b = sgx, sg
    # one scale for both big blocks: |d res| <= the BatchNorm bound, |d ft[u]| <= (row sum of the edge weights out of u) x that bound
    s1 = _C.halves_scale_from_slots(slots, mult=rowsum_bound(g, ctx.adrop[0] if ctx.adrop else 0.0))
    _C.bn_act_bwd_apply_halves(dy, x, mean, invstd, bn_w, bn_b, True, drop_p, seed, sg if bn_training else None, sgx if bn_training else None, total,
                               s1, buf[:, B:], HD, HD, out=dx, h2_off=piece)
    da = _C.spmm_dot_halves(g.csr, dx.unflatten(1, (H, D)), a_d, g.csr2csc, ft, s1, buf, D, piece)
    dz, der = _C.gat_attn_bwd(g.csc, el, er, None, None, slope, H, a, da, None, None, has_er, ctx.zs, drop=ctx.adrop)
    d_el = _C.segment_sum(g.csr, dz, g.csr2csc)
    # the attention columns (known only now, of their own magnitude) under a second scale, and the zero padding of the operand
    slots2 = _C.absmax_slots(dy.device)
    _C.absmax_into(d_el, slots2)
    if has_er:
        _C.absmax_into(der, slots2)
    s2 = _C.halves_scale_from_slots(slots2, cap=s1, cap_ratio=TAIL_CAP)
    segs = [(c, H, d_el)] + ([(c + H, H, der)] if has_er else [])
    if B != HD:
        segs += [(HD, B - HD, None), (B + HD, B - HD, None)]
    if used < P:
        segs.append((used, P - used, None))
    _C.halves_tail(segs, s2, buf, piece)
    dW = dh = None
    xh = gemm.Halves(h, ctx.xscale, *ctx.halves)
    if ctx.needs_input_grad[0]:
        Ws = gemm.split_right(Wcat if kp else Wcat.t().contiguous(), scale=ctx.wscale)
        link = ctx.bn_link
        st = link.stats_for(N, Ws.n, piece) if link is not None else None
        dh = _C.gemm_halves3_nt(buf, Ws.buf, s1, Ws.scale, piece, Ws.piece, piece, a2_off=piece, scale_a2=s2, k_split=c, b_frag=Ws.order == 3, n=Ws.n, bn=st)
        if st is not None:
            link.deliver(st, dh)
    if ctx.needs_input_grad[1]:
        def wgrad():
            dWk = _C.gemm_halves3_tn(xh.buf, buf, xh.scale, s1, xh.piece, piece, xh.F, P, x2_off=xh.h2_off, d2_off=piece, scale_d2=s2, p_split=c)
            return dWk if kp else dWk.t().contiguous()
        dW = side.run(wgrad, xh, buf, s1, s2) if (side.usable(buf) and xh.piece * piece >= side.MIN_OUT) else wgrad()
    return (dh, dW, d_bn_w if ctx.needs_input_grad[2] else None, d_bn_b if ctx.needs_input_grad[3] else None,
            None, None, None, None, None, None, None, None, None, None, None, None, None)


_GATHidden._backward_direct = staticmethod(_backward_direct)


def cat_weight_aggfirst(conv):
    """[W_res ; wl ; wr ; 0-pad] — the GEMM that remains on the layer INPUT when the aggregation runs before the projection."""
    H, D = conv._num_heads, conv._out_feats
    W = conv.fc.weight
    Wh = W.view(H, D, -1)
    rows = [conv.res_fc.weight] if conv.res_fc is not None else []
    rows.append((Wh * conv.attn_l.view(H, D, 1)).sum(1))
    if conv.attn_r is not None:
        rows.append((Wh * conv.attn_r.view(H, D, 1)).sum(1))
    used = sum(r.shape[0] for r in rows)
    pad = (-used) % 128
    if pad:
        rows.append(W.new_zeros(pad, W.shape[1]))
    return torch.cat(rows)


# Layout of the merged weight handed to the layer node: [K, P] (a transposed copy, K*P floats per step) instead of [P, K].
# Same three GEMMs, other operand layouts, i.e. other library kernels: on [169 343, 750] x [750, 1536] the best kernels reach
# 128 / 134 / 147 TFLOP/s (forward / dX / dW) against 123 / 118 / 142 for the [P, K] form (tools/exp_gemm_layouts.py).
WEIGHT_KP = True


def _kp(w):
    return w.t().contiguous() if WEIGHT_KP else w


AGG_FIRST = True  # aggregate-before-project for layers whose input is narrower than one head (Fin <= D, H <= 4)


# The aggregate-first layer's dense products on the grouped fp16-halves kernels (csrc/halves3.hip, ABI 16) instead of bot_skinny_gemm_f32 /
# bot_tn_gemm_f32 / the stock fp32 GEMM: the aggregated slab is written by the SpMM as a halves operand next to the input's halves, one
# launch computes  rst_h = [x | z_h] [Wres_h | W_h]^T  for all heads, one  d z_h = d rst_h W_h,  one all weight gradients.
L0_HALVES = os.environ.get("BOT_L0_HALVES", "1") != "0"
# ... and the layer's BatchNorm backward writes d rst directly as that launch's left operand, under a scale BOUNDED from the reduce pass
# (bot_bn_bwd_bound_f32) instead of measured on a finished fp32 dx: no fp32 dx, no split pass
L0_DIRECT = os.environ.get("BOT_L0_DIRECT", "1") != "0"
_L0_TABLES = {}
_L0_DH = {}


def _l0_dh(device, N, H, D, DP):
    """The gradient operand [N, 2 H DP] of the direct form: allocated once per shape with its padding columns zeroed (the apply pass writes
    only the D columns of each head's block: the key carries D, a narrower layer must not inherit a wider one's columns); it lives inside
    one backward call, so layers and steps can share it."""
    key = (str(device), N, H, D, DP)
    buf = _L0_DH.get(key)
    if buf is None:
        buf = torch.zeros((N, 2 * H * DP), dtype=torch.float16, device=device)
        if not (buf.is_cuda and torch.cuda.is_current_stream_capturing()):     # (a buffer born inside a capture belongs to that graph's pool)
            _L0_DH[key] = buf
    return buf


def _l0_tables(H, D, Fin, P2, kp, N):
    """Group / tile lists of the three grouped launches (host-side, cached per shape)."""
    key = (H, D, Fin, P2, kp, N)
    if key not in _L0_TABLES:
        FP = (Fin + 63) // 64 * 64           # piece width of x and of every z_h
        DP = (D + 63) // 64 * 64             # a head's block of the gradient operand
        HD = H * D
        # forward: group h = output columns h D .. h D + D - 1; reduction over x (A columns 0 ..) then z_h (A columns FP (1 + h) ..)
        fwd = [(h * D, D, 0, h * FP, 2 * FP // 32, h * D) for h in range(H)]
        # d z_h [N, Fin] = d rst_h [N, D] W_h [D, Fin]: A columns h DP .., B rows h Fin .. (W_h^T), output slab h
        dz = [(h * Fin, Fin, 0, h * DP, DP // 32, h * N * Fin) for h in range(H)]
        # weight gradients: x-role (192-column tiles) = [x | z_0 .. z_{H-1}] (Fin of FP columns used), d-role = the gradient operand's head
        # blocks in 128-column tiles (DP = 256: two per head, 128 and D - 128 columns used): 192 x 128 tiles, 15 % padding (the other way
        # round - 192-column tiles over a head's 250 columns - pads 43 %)
        tn = []
        for h in range(H):
            for j in range((D + 127) // 128):
                pv = min(128, D - 128 * j)
                col = h * D + 128 * j                                # row of d W / column of the merged gradient
                tn.append((FP * (1 + h), Fin, h * DP + 128 * j, pv, col * Fin, Fin, 1))                  # d W_h [D, Fin] = (z_h^T d rst_h)^T
                if kp:
                    tn.append((0, Fin, h * DP + 128 * j, pv, HD * Fin + col, P2, 0))                       # d Wres^T [Fin, P2] = x^T d rst
                else:
                    tn.append((0, Fin, h * DP + 128 * j, pv, HD * Fin + col * Fin, Fin, 1))               # d Wres [P2, Fin]
        _L0_TABLES[key] = (FP, DP, fwd, dz, tn)
    return _L0_TABLES[key]


def _l0_halves_ok(h, H, D, Fin, has_res, sym, attn_p, graph) -> bool:
    """The grouped-halves form of the aggregate-first layer: needs the residual branch (its columns are the launch's output), row sums of
    the edge weights <= 1 / (1 - attn_p) < 4 (the slab shares x's scale: fp16 has two binades of headroom above it) and the hand-written
    kernels as the halves path."""
    return (L0_HALVES and (h.is_cuda or FORCE) and has_res and not sym and attn_p <= 0.7 and D <= 256 and Fin <= 192 and H * 2 * ((D + 127) // 128) <= 16
            and gemm.MODE == "halves" and gemm.NT_KERNEL == "halves3" and gemm.TN_KERNEL == "halves3" and (h.shape[0] >= gemm.MIN_ROWS or gemm.FORCE or FORCE))


def _small_mm(a, b, b_is_kn, out):
    """out = a @ (b if b_is_kn else b^T) for a handful of output columns (the attention columns of the merged projection)."""
    if a.is_cuda and SKINNY:
        return _C.skinny_gemm(a, b, b_is_kn=b_is_kn, out=out)
    out.copy_(a @ (b if b_is_kn else b.t()))
    return out


def use_agg_first(conv) -> bool:
    return AGG_FIRST and conv._in_src_feats <= conv._out_feats and conv._num_heads <= 4 and conv._in_src_feats <= 256


class _GATHiddenAggFirst(torch.autograd.Function):
    """Same layer as _GATHidden with the aggregation moved in front of the projection:
        rst[v,h,:] = W_h (sum_e a[e,h] x[u,:]) + res[v,h,:]      (models.py:490-492, :547, :558-560, linearity of the sum)
    The sparse sweeps gather Fin floats per edge forward and H*Fin backward instead of H*D both ways, the projection becomes
    a batched GEMM over heads on the aggregated slab [H, N, Fin], and in partitioned mode the halo rows are [x | el]."""

    @staticmethod
    def forward(ctx, h, W, Wr, bn_w, bn_b, graph, bn, H, D, has_res, has_er, slope, attn_p, drop_p, bn_training, kp, sym, y_needed=True):
        N, Fin, HD = h.shape[0], h.shape[1], H * D
        csc = graph.csc
        ctx.sym = sym
        ctx.kp = kp                                                     # Wr is [Fin, P2] (see WEIGHT_KP) instead of [P2, Fin]
        # small-K products (K = Fin <= 256) on bot_skinny_gemm_f32: fp32 operands split in registers into bf16 terms, MFMA products
        skinny = SKINNY and h.is_cuda and Fin <= 256 and N >= 4096
        l0h = _l0_halves_ok(h, H, D, Fin, has_res, sym, attn_p, graph) and not ctx.needs_input_grad[0]
        ctx.l0h = l0h
        if l0h:
            # only the attention columns now (el / er feed the aggregation); the residual columns come out of the grouped launch below
            P2 = Wr.shape[1 if kp else 0]
            out2 = torch.empty((N, P2), dtype=h.dtype, device=h.device)
            _small_mm(h, Wr[:, HD:] if kp else Wr[HD:], kp, out2[:, HD:])
        elif skinny:
            out2 = _C.skinny_gemm(h, Wr, b_is_kn=kp, out=torch.empty((N, Wr.shape[1 if kp else 0]), dtype=h.dtype, device=h.device))
        else:
            out2 = torch.mm(h, Wr) if kp else torch.mm(h, Wr.t())       # [N, P2] = [res | el | er | pad]
        ctx.skinny = skinny
        c = HD if has_res else 0
        ext = None
        if graph.halo is not None:
            import torch.distributed as dist
            plan = graph.halo
            Wd = _ext_width(Fin, H)
            ext = torch.empty((N + plan.n_halo, Wd), dtype=h.dtype, device=h.device)
            ext[:N, :Fin] = h
            ext[:N, Fin:Fin + H] = out2[:, c:c + H]
            if Wd > Fin + H:
                ext[:N, Fin + H:].zero_()
            send = _C.gather_rows(ext[:N], plan.send_rows) if plan.n_send else ext.new_empty((0, Wd))
            halo.a2a(ext[N:], send, plan.recv_splits, plan.send_splits, plan.group)
            xsrc = ext[:, :Fin]
            el = ext[:, Fin:Fin + H].contiguous()
        else:
            xsrc = h
            el = out2[:, c:c + H].contiguous()
        er = out2[:, c + H:c + 2 * H].contiguous() if has_er else None
        if sym:
            s_out, w_e = sym_scales(graph)
            el = el * s_out.unsqueeze(1)
        ctx.zs = _C.zsign_buffer(csc, H, slope)
        ctx.adrop = (attn_p, new_dropout_seed(attn_p)) if attn_p > 0 else None
        a, a_d = _C.gat_attn_fwd(csc, el, er, None, None, None, slope, H, None, ctx.zs, drop=ctx.adrop or (0.0, 0))
        if sym:
            a_d = a_d * w_e
        if l0h:
            global L0_CALLS
            L0_CALLS += 1
            FP, DP, g_fwd, _, _ = _l0_tables(H, D, Fin, out2.shape[1], kp, N)
            KA = (1 + H) * FP
            # left operand [x | z_0 .. z_{H-1}] (first halves, then the second halves KA columns behind), ONE scale: the rows of a_d sum
            # to <= 1 / (1 - attn_p), so max|z| <= max|x| / (1 - attn_p), inside fp16's range above x's scale (_l0_halves_ok)
            # (partitioned mode: the sweep gathers halo rows too - the scale must cover every row of the extended table)
            xscale = _C.halves_scale(h) if ext is None else _C.halves_scale_from_slots(_C.absmax_into(xsrc, _C.absmax_slots(h.device)))
            A = torch.empty((N, 2 * KA), dtype=torch.float16, device=h.device)
            _C.halves_split_cols(h, xscale, 2, A, KA, 0, FP)
            _C.spmm_bcast_halves(csc, xsrc, a_d, None, xscale, A, FP, FP, KA, FP)
            # right operand: row j = output column j = [Wres_j | W_j], one scale
            Wres = (Wr.t()[:HD] if kp else Wr[:HD]).contiguous()
            slots = _C.absmax_slots(h.device)
            _C.absmax_into(Wres, slots)
            _C.absmax_into(W, slots)
            wscale = _C.halves_scale_from_slots(slots)
            B = torch.empty((HD, 6 * FP), dtype=torch.float16, device=h.device)
            _C.halves_split_cols(Wres, wscale, 1, B, 2 * FP, 0, FP)
            _C.halves_split_cols(W, wscale, 1, B, 2 * FP, FP, FP)
            x = out2[:, :HD]
            # BatchNorm's batch statistics as a by-product of the launch that writes x (per 256-row tile and column: sum, sum of squares,
            # extremes): the statistics pass over the [N, H D] output is gone (ops.stats_partials_for says when the epilogue takes them)
            partials = None
            if bn is not None and h.is_cuda | FORCE:
                # (only the 256 x 32 form of the grouped launch carries the by-product: BOT_NT_KERNEL=128x64 or an odd number of k-steps takes
                # the statistics pass instead of failing in bot_gemm_halves3_nt_grouped2_f32)
                carries = int(_C._lib.bot_gemm_halves3_nt_bn_rows(2 * FP)) == 256
                partials = stats_partials_for(bn, bn_training, N, HD, h.device, carries and gemm.epilogue_piece(HD, x) is not None)
            _C.gemm_halves3_nt_grouped(A, B, xscale, wscale, KA, 2 * FP, out2, g_fwd, FP // 32, stats=partials)
            ctx.graph = graph
            keep = (h, W, Wr, A, ext if ext is not None else h, el, er, a, a_d, xscale)
            if bn is None:
                ctx.save_for_backward(*keep)
                ctx.cfg = (H, D, has_res, has_er, slope, None)
                return x
            y, mean, invstd, total, sync, group, seed, ctx.out_link = _epilogue_forward(x, bn, bn_w, bn_b, bn_training, drop_p, y_needed, partials=partials)
            ctx.save_for_backward(*keep, x, mean, invstd, bn_w, bn_b)
            ctx.cfg = (H, D, has_res, has_er, slope, (drop_p, seed, bn_training, sync, group, total))
            return y
        z = _C.spmm_bcast(csc, xsrc, a_d, None, head_outer=True)        # [H, N, Fin]
        Wh = W.view(H, D, Fin)
        # per-head projection (plain 2-D GEMMs: each has its own tuned kernel selection, see bot_amd/tuning), accumulated in
        # place onto the residual columns of the [N, P2] buffer (beta = 1): no [N, H, D] add pass, x is a row-strided view
        if has_res and skinny:      # the H heads as one strided batch, written side by side onto the residual columns
            _C.skinny_gemm(z, Wh, b_is_kn=False, out=out2, accumulate=True, batch=H, strides=(N * Fin, D * Fin, D), m=N, n=D, k=Fin)
            x = out2[:, :HD]
        elif has_res:
            for i in range(H):
                out2[:, i * D:(i + 1) * D].addmm_(z[i], Wh[i].t())
            x = out2[:, :HD]
        elif skinny:
            agg = _C.skinny_gemm(z, Wh, b_is_kn=False, out=torch.empty((H, N, D), dtype=h.dtype, device=h.device), batch=H,
                                 strides=(N * Fin, D * Fin, N * D), m=N, n=D, k=Fin)
            x = agg.permute(1, 0, 2).reshape(N, HD)
        else:
            agg = torch.empty((H, N, D), dtype=h.dtype, device=h.device)
            for i in range(H):
                torch.mm(z[i], Wh[i].t(), out=agg[i])
            x = agg.permute(1, 0, 2).reshape(N, HD)
        ctx.graph = graph
        keep = (h, W, Wr, z, ext if ext is not None else h, el, er, a, a_d)
        if bn is None:
            ctx.save_for_backward(*keep)
            ctx.cfg = (H, D, has_res, has_er, slope, None)
            return x
        y, mean, invstd, total, sync, group, seed, ctx.out_link = _epilogue_forward(x, bn, bn_w, bn_b, bn_training, drop_p, y_needed)
        ctx.save_for_backward(*keep, x, mean, invstd, bn_w, bn_b)
        ctx.cfg = (H, D, has_res, has_er, slope, (drop_p, seed, bn_training, sync, group, total))
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        H, D, has_res, has_er, slope, epi = ctx.cfg
        g = ctx.graph
        dy = dy.contiguous()
        d_bn_w = d_bn_b = None
        l0h, xscale = ctx.l0h, None
        saved = ctx.saved_tensors
        if l0h:                                                          # z: the operand [x | z_0 .. z_{H-1}] as fp16 halves
            xscale, saved = saved[9], saved[:9] + saved[10:]
        if epi is None:
            h, W, Wr, z, table, el, er, a, a_d = saved
        else:
            h, W, Wr, z, table, el, er, a, a_d, x, mean, invstd, bn_w, bn_b = saved
            drop_p, seed, bn_training, sync, group, total = epi
        kp = ctx.kp
        N, Fin, HD, P2 = h.shape[0], h.shape[1], H * D, Wr.shape[1 if kp else 0]
        dout2 = torch.empty((N, P2), dtype=h.dtype, device=h.device)
        dx = dout2[:, :HD] if has_res else torch.empty((N, HD), dtype=h.dtype, device=h.device)
        direct = l0h and L0_DIRECT and epi is not None and D % 2 == 0
        slots = _C.absmax_slots(dy.device) if l0h and (ABSMAX_BYPRODUCT or direct) and epi is not None else None
        Dh = dscale = None
        if epi is None:
            dx.copy_(dy)
        else:
            one_rank = direct and not (bn_training and sync)
            if one_rank:
                sg, sgx = _bwd_reduce_bound(ctx, dy, x, mean, invstd, bn_w, bn_b, drop_p, seed, bn_training, total, slots)
            else:
                sg, sgx, ws = _bwd_reduce(ctx, dy, x, mean, invstd, bn_w, bn_b, drop_p, seed, want_max=direct)
            d_bn_w, d_bn_b = sgx, sg
            if bn_training and sync:
                both = torch.stack([sg, sgx])
                dist.all_reduce(both, group=group)
                sg, sgx = both[0].contiguous(), both[1].contiguous()
            if direct:
                # a bound on max|dx| from the reduce pass's column maxima and the final sums -> the operand's scale -> dx written as halves
                DP = (D + 63) // 64 * 64
                if not one_rank:
                    _bwd_bound(ws, N, sg if bn_training else None, sgx if bn_training else None, total, bn_w, invstd, slots)
                dscale = _C.halves_scale_from_slots(slots)
                Dh = _C.bn_act_bwd_apply_halves(dy, x, mean, invstd, bn_w, bn_b, True, drop_p, seed, sg if bn_training else None,
                                                sgx if bn_training else None, total, dscale, _l0_dh(dy.device, N, H, D, DP), D, DP)
            else:
                _C.bn_act_bwd_apply(dy, x, mean, invstd, bn_w, bn_b, True, drop_p, seed, sg if bn_training else None,
                                    sgx if bn_training else None, total, out=dx, absmax=slots)
        Wh = W.view(H, D, Fin)
        dz = torch.empty((H, N, Fin), dtype=h.dtype, device=h.device)    # gradient of the aggregated slab
        dW3 = torch.empty((H, D, Fin), dtype=h.dtype, device=h.device) if ctx.needs_input_grad[1] and not l0h else None
        if l0h:
            # the gradient of the layer's output as a LEFT halves operand, each head's D columns in a block of DP (zero padded): it is the
            # A operand of d z_h = d rst_h W_h and the x-role of every weight gradient
            FP, DP, _, g_dz, t_tn = _l0_tables(H, D, Fin, P2, kp, N)
            if Dh is None:
                dscale = _C.halves_scale_from_slots(slots) if slots is not None else _C.halves_scale(dx)
                Dh = _C.halves_split_heads(dx, dscale, H, D, DP)
            Wt = gemm.split(Wh.transpose(1, 2).reshape(H * Fin, D), 1)           # rows h Fin + f = W_h[:, f]: the right operand of d z_h
            _C.gemm_halves3_nt_grouped(Dh, Wt.buf, dscale, Wt.scale, H * DP, Wt.piece, dz[0], g_dz, 0)
            # every weight gradient from ONE grouped launch over the head blocks: d W_h = d rst_h^T z_h and the residual rows / columns of
            # the merged gradient d rst^T x (written in Wr's layout).  Both operands exist from here on and only the optimizer needs the
            # result: on the side stream (bot_amd.side) the launch runs beside the sparse sweep and the attention backward below
            flat = torch.empty(HD * Fin + P2 * Fin, dtype=h.dtype, device=h.device)

            def wgrad():
                _C.gemm_halves3_tn_grouped(z, Dh, xscale, dscale, (1 + H) * FP, H * DP, flat, t_tn)
                return flat
            if side.usable(flat):
                side.run(wgrad, z, Dh, xscale, dscale, flat)
            else:
                wgrad()
        elif ctx.skinny and D <= 256:     # d z_i = d x_i W_i for the H heads in one launch (A = column slices of d x)
            _C.skinny_gemm(dx, Wh, b_is_kn=True, out=dz, batch=H, strides=(D, D * Fin, N * Fin), m=N, n=Fin, k=D)
        if ctx.skinny and dW3 is not None:   # d W_i = d x_i^T z_i, reductions over the N rows: fp32 MFMA, chunked, one launch for the H heads
            _C.tn_gemm(dx, z, out=dW3, batch=H, strides=(D, N * Fin, 0), n=N, kx=D, ky=Fin)
        for i in range(H if not l0h else 0):
            dxi = dx[:, i * D:(i + 1) * D]                               # [N, D] column slice (row-strided)
            if not (ctx.skinny and D <= 256):
                torch.mm(dxi, Wh[i], out=dz[i])
            if dW3 is not None and not ctx.skinny:
                torch.mm(dxi.t(), z[i], out=dW3[i])
        dW = dW3.view(HD, Fin) if dW3 is not None else None
        halo = g.halo is not None
        c = HD if has_res else 0
        need_dh = ctx.needs_input_grad[0]
        if not need_dh:
            # The layer input is DATA (the first layer of a stack: features + label columns, run.py:256-263): only the gradient
            # of the attention weights is wanted from the sparse sweep.  d a[e,h] = <x[u], dz[v,h,:]> by ONE sweep over the
            # in-edges that gathers the Fin-wide source row once for all heads (4*Fin bytes per edge) — instead of the transposed
            # sweep that gathers the [H, Fin] gradient slab per edge (4*H*Fin) and also produces the unused d x.
            xsrc = table[:, :Fin] if halo else h
            da = _C.sddmm_dot_bcast(g.csc, xsrc, dz)
            dh_g = None
        elif halo:
            dext = torch.empty_like(table)
            _, da = _C.spmm_dot_bcast(g.csr, dz, a_d, g.csr2csc, table[:, :Fin], out=dext[:, :Fin])
        else:
            dh_g, da = _C.spmm_dot_bcast(g.csr, dz, a_d, g.csr2csc, h)
        if ctx.sym:
            s_out, w_e = sym_scales(g)
            da = da * w_e
        dz_e, der = _C.gat_attn_bwd(g.csc, el, er, None, None, slope, H, a, da, None, None, has_er, ctx.zs, drop=ctx.adrop)
        d_el = _C.segment_sum(g.csr, dz_e, g.csr2csc)
        if ctx.sym:
            d_el = d_el * s_out.unsqueeze(1)
        if halo and not need_dh:                                        # only d el travels back: [n_ext, H] rows, not [n_ext, Fin + H]
            dsm = d_el.new_zeros((d_el.shape[0], (H + 3) // 4 * 4))
            dsm[:, :H] = d_el
            dout2[:, c:c + H] = _extend_backward(g, dsm, N)[:, :H]
        elif halo:
            dext[:, Fin:Fin + H] = d_el
            if dext.shape[1] > Fin + H:
                dext[:, Fin + H:].zero_()
            own = _extend_backward(g, dext, N)
            dh_g = own[:, :Fin]
            dout2[:, c:c + H] = own[:, Fin:Fin + H]
        else:
            dout2[:, c:c + H] = d_el
        if has_er:
            dout2[:, c + H:c + 2 * H] = der
        used = c + (2 * H if has_er else H)
        if used < P2:
            dout2[:, used:].zero_()
        dWr = None
        if l0h:
            # the attention columns of the merged gradient (a handful, their own magnitude) apart: other rows / columns of `flat` than the
            # grouped launch on the side stream writes, so this narrow product runs while that one finishes
            dW = flat[:HD * Fin].view(HD, Fin)
            dWr = flat[HD * Fin:].view((Fin, P2) if kp else (P2, Fin))
            tail = dout2[:, HD:]
            if h.is_cuda and tail.shape[1] <= 32:
                _C.tn_narrow(tail, h, dWr[:, HD:] if kp else dWr[HD:], transpose_out=kp)
            elif h.is_cuda:
                _C.tn_gemm(tail, h, out=dWr[:, HD:] if kp else dWr[HD:], transpose_out=kp)
            elif kp:
                dWr[:, HD:] = h.t() @ tail
            else:
                dWr[HD:] = tail.t() @ h
            # fc.weight's gradient has a second contribution (through the merged weight's attention rows) that autograd ADDS on this
            # stream: joined here - this layer is the end of the backward pass, nothing else is left to run beside the side stream
            side.join()
            if not ctx.needs_input_grad[1]:
                dW = None
            if not ctx.needs_input_grad[2]:
                dWr = None
        elif ctx.needs_input_grad[2]:
            dWr = torch.mm(h.t(), dout2) if kp else torch.mm(dout2.t(), h)
        dh = None
        if need_dh:
            dh = torch.addmm(dh_g, dout2, Wr.t() if kp else Wr)
        return (dh, dW, dWr, d_bn_w if ctx.needs_input_grad[3] else None, d_bn_b if ctx.needs_input_grad[4] else None,
                None, None, None, None, None, None, None, None, None, None, None, None, None)


def halves_only_consumer(conv, norm, activation, graph, training, stack_residual, n_rows, on_device) -> bool:
    """Will `gat_hidden_layer(conv, norm, ...)` read its input ONLY as the fp16 halves the previous layer's epilogue stashed?  True for the
    merged-GEMM node on the halves path (not the aggregate-first node, which gathers the fp32 rows; not the modular fallback).  The
    stack asks this BEFORE it runs the previous layer, which may then skip storing its fp32 output (`_epilogue_forward`)."""
    return (SKIP_Y and (on_device or FORCE) and can_fuse(conv, norm, activation, graph, training, stack_residual) and not use_agg_first(conv)
            and gemm.MODE == "halves" and (gemm.FORCE or (on_device and n_rows >= gemm.MIN_ROWS)))


_INPUT_DROPPED = [False]


class input_already_dropped:
    """While active, the NEXT stack forward skips its input dropout (models.py:711 / GCN's `input_drop`): bot_amd.train assembled the
    layer-0 operand with the dropout applied in the same pass (bot_build_input_f32).  Consumed by the first stack that asks."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        _INPUT_DROPPED[0] = self.on
        return self

    def __exit__(self, *exc):
        _INPUT_DROPPED[0] = False
        return False


def take_input_dropped() -> bool:
    v = _INPUT_DROPPED[0]
    _INPUT_DROPPED[0] = False
    return v


def gat_hidden_layer(conv, bn, graph, h, dropout_p, training, y_needed=True):
    """`dropout(relu(bn(conv(graph, h).flatten(1))))` as one autograd node (bn None: just `conv(graph, h).flatten(1)`,
    the stack's output layer).  h: [N, Fin] -> [N, H*D].  y_needed=False: see `_epilogue_forward`."""
    from . import has_zero_in_degree
    if not conv._allow_zero_in_degree:
        assert not has_zero_in_degree(graph), "0-in-degree nodes (models.py:477-479)"
    global CALLS
    CALLS += 1
    H, D = conv._num_heads, conv._out_feats
    attn_p = conv.attn_drop.p if training else 0.0
    bn_w = bn.weight if bn is not None and bn.affine else None
    bn_b = bn.bias if bn is not None and bn.affine else None
    bn_training = bn is not None and (bn.training or not bn.track_running_stats)
    if use_agg_first(conv):
        global AGG_CALLS
        AGG_CALLS += 1
        return _GATHiddenAggFirst.apply(h, conv.fc.weight, merged_weight(conv, with_fc=False), bn_w, bn_b, graph, bn, H, D,
                                        conv.res_fc is not None, conv.attn_r is not None, conv.leaky_relu.negative_slope,
                                        attn_p, dropout_p if training and bn is not None else 0.0, bn_training, WEIGHT_KP,
                                        conv._use_symmetric_norm, y_needed)
    if bn is None:
        return _GATHidden.apply(h, merged_weight(conv), None, None, graph, None, H, D, conv.res_fc is not None,
                                conv.attn_r is not None, conv.leaky_relu.negative_slope, attn_p, 0.0, False, WEIGHT_KP,
                                conv._use_symmetric_norm)
    return _GATHidden.apply(h, merged_weight(conv), bn.weight if bn.affine else None, bn.bias if bn.affine else None, graph, bn,
                            H, D, conv.res_fc is not None, conv.attn_r is not None, conv.leaky_relu.negative_slope,
                            conv.attn_drop.p if training else 0.0, dropout_p if training else 0.0, bn_training, WEIGHT_KP,
                            conv._use_symmetric_norm, y_needed)


# ------------------------------------------------------------------------------------------------ inference-only forward (f3)
# `evaluate()` (run.py:290-322) runs the stack in eval mode under no_grad once per epoch, plus once per label-reuse iteration
# (run.py:304-308).  Nothing is needed for a backward, eval-mode BatchNorm is a per-column affine and dropout is the identity, so
# a whole layer is ONE GEMM + ONE sweep over the in-edges (bot_gat_infer_f32): no attention weights, no sign bytes and no
# pre-BatchNorm [N, H*D] tensor reach HBM.

_GENERATION = [0]


def bump_generation():
    """Parameters / running statistics were changed by something that moves no tensor version counter — a hipGraph replay of a
    train step (bot_amd.train.CapturedTrainStep: optimizer and BatchNorm updates happen inside the graph).  Every cached
    inference-path value derived from them is stale from here on."""
    _GENERATION[0] += 1


def _versions(*ts):
    return (_GENERATION[0],) + tuple((t.data_ptr(), t._version) for t in ts if t is not None)


def _cached(owner, slot, key, make):
    """Value derived from parameters, recomputed when a parameter was modified in place (optimizer step) or replaced."""
    c = owner.__dict__.setdefault("_bot_infer_cache", {})
    hit = c.get(slot)
    if hit is None or hit[0] != key:
        hit = c[slot] = (key, make())
    return hit[1]


def _infer_key(conv):
    return _versions(conv.fc.weight, conv.res_fc.weight if conv.res_fc is not None else None, conv.attn_l, conv.attn_r) + (WEIGHT_KP,)


def infer_weight(conv):
    """Merged projection weight of the layer in the layout the GEMM wants, cached across forward calls (evaluate() calls the
    model 1 + n_label_iters times between two optimizer steps)."""
    return _cached(conv, "wcat", _infer_key(conv), lambda: merged_weight(conv).detach())


def eval_affine(mod):
    """(scale, shift) of the stack's eval-mode epilogue in front of the activation: nn.BatchNorm1d with running statistics
    (models.py:726-727) or the bias-only ElementWiseLinear (models.py:728-729, :734)."""
    if isinstance(mod, nn.BatchNorm1d):
        key = _versions(mod.weight, mod.bias, mod.running_mean, mod.running_var)

        def make():
            scale = torch.rsqrt(mod.running_var + mod.eps)
            if mod.affine:
                scale = scale * mod.weight
            shift = -mod.running_mean * scale
            if mod.affine:
                shift = shift + mod.bias
            return scale.contiguous(), shift.contiguous()
        return _cached(mod, "affine", key, make)
    return (mod.weight.detach() if mod.weight is not None else None), (mod.bias.detach() if mod.bias is not None else None)


def sweep_is_row_kernel(graph, H, D) -> bool:
    """True when the aggregation of this graph runs on the row-per-wavefront kernels — the family the fused inference sweep
    belongs to.  Dense graphs (mean degree >= 96: S-reddit, S-proteins) aggregate with the L2-blocked SpMM, which is ~2x faster
    than any row kernel there (10.6 vs 19 ms at S-proteins), so their eval-mode forward keeps the attention kernel + blocked SpMM."""
    from .. import blocked
    csc = graph.csc
    return not csc.indptr.is_cuda or blocked.plan_for(csc, graph.number_of_nodes(), H, D) is None


def can_infer(conv, epi, activation, graph, stack_residual, last) -> bool:
    """`epi`: the module applied behind the layer (BatchNorm1d in eval mode, or a bias-only ElementWiseLinear)."""
    if torch.is_grad_enabled() or not hasattr(conv, "fc") or conv._activation is not None or conv.training:
        return False
    if isinstance(epi, nn.BatchNorm1d):
        if epi.training or not epi.track_running_stats or epi.running_mean is None:
            return False
    elif not (type(epi).__name__ == "ElementWiseLinear" and not epi.inplace):
        return False
    if not last and (activation not in (F.relu, torch.relu) or stack_residual):
        return False
    if last and conv._num_heads != 1:
        return False
    return (not (conv._use_symmetric_norm and graph.halo is not None) and (graph.halo is not None or not graph.is_block)
            and conv._out_feats <= 256 and sweep_is_row_kernel(graph, conv._num_heads, conv._out_feats))


class _LabelReuse:
    """Set by `label_reuse`: the first `static_cols` input columns do not change between the forward calls made inside the
    context (only the label columns are rewritten, run.py:304-308), so their share of the first layer's projection is
    computed once."""
    active = None


class label_reuse:
    def __init__(self, static_cols):
        self.static_cols, self.store = int(static_cols), {}

    def __enter__(self):
        self.prev, _LabelReuse.active = _LabelReuse.active, self
        return self

    def __exit__(self, *exc):
        _LabelReuse.active = self.prev
        self.store.clear()
        return False


INFER_CALLS = 0  # number of inference-layer invocations (tests assert the path was taken)


def count_infer():
    global INFER_CALLS
    INFER_CALLS += 1


# The eval-mode forward of the aggregate-first layer on the training forward's kernels (round 4): attention columns -> softmax -> the SpMM
# that writes the aggregated slab as a halves operand -> ONE grouped NT launch whose epilogue applies the eval-mode BatchNorm / bias and
# the ReLU.  Instead of the [N, 168] x [168, 1536] projection + a sweep that gathers 3 000-byte rows: 672-byte rows and no [N, H D] slab.
L0_INFER = os.environ.get("BOT_L0_INFER", "1") != "0"
L0_INFER_CALLS = 0


L0_INFER_OVER_LABEL_REUSE = os.environ.get("BOT_L0_INFER_LABEL_REUSE", "1") != "0"


def _infer_l0_ok(conv, graph, h, label_reuse_cols) -> bool:
    # (label-reuse iterations: the static-column form saves the feature columns' share of a PROJECT-first layer 0 with stock fp32 GEMMs;
    # this form has no such projection — the whole layer is cheaper than the remainder of that one, tools/r04_infer_ab.sh)
    return (L0_INFER and L0_HALVES and use_agg_first(conv) and (not label_reuse_cols or L0_INFER_OVER_LABEL_REUSE) and graph.halo is None
            and conv.res_fc is not None
            and not conv._use_symmetric_norm and conv._out_feats <= 256 and h.shape[1] <= 192 and (h.is_cuda or FORCE)
            and gemm.MODE == "halves" and gemm.NT_KERNEL == "halves3" and (h.shape[0] >= gemm.MIN_ROWS or gemm.FORCE or FORCE))


def _infer_l0(conv, epi, graph, h, relu):
    global L0_INFER_CALLS
    L0_INFER_CALLS += 1
    H, D = conv._num_heads, conv._out_feats
    HD, N, Fin = H * D, h.shape[0], h.shape[1]
    has_er = conv.attn_r is not None
    csc = graph.csc
    Wr = _cached(conv, "wcat_agg", _infer_key(conv), lambda: merged_weight(conv, with_fc=False).detach())   # [Fin, P2] / [P2, Fin]: [res | el | er | pad]
    kp = WEIGHT_KP
    P2 = Wr.shape[1 if kp else 0]
    FP, _, g_fwd, _, _ = _l0_tables(H, D, Fin, P2, kp, N)
    KA = (1 + H) * FP

    def right_operand():        # row j = output column j = [Wres_j | W_j] as fp16 halves under one scale; lives with the weights' versions
        Wres = (Wr.t()[:HD] if kp else Wr[:HD]).contiguous()
        W = conv.fc.weight.detach()
        slots = _C.absmax_slots(h.device)
        _C.absmax_into(Wres, slots)
        _C.absmax_into(W, slots)
        wscale = _C.halves_scale_from_slots(slots)
        B = torch.empty((HD, 6 * FP), dtype=torch.float16, device=h.device)
        _C.halves_split_cols(Wres, wscale, 1, B, 2 * FP, 0, FP)
        _C.halves_split_cols(W, wscale, 1, B, 2 * FP, FP, FP)
        return B, wscale
    B, wscale = _cached(conv, "infer_l0_halves", _infer_key(conv) + (str(h.device),), right_operand)
    tail = torch.empty((N, P2 - HD), dtype=h.dtype, device=h.device)              # the attention columns (el | er | pad)
    _small_mm(h, Wr[:, HD:] if kp else Wr[HD:], kp, tail)
    el = tail[:, :H]
    er = tail[:, H:2 * H] if has_er else None
    a = _C.gat_attn_fwd(csc, el, er, None, None, None, conv.leaky_relu.negative_slope, H, None)
    # [x | z_0 .. z_{H-1}] as one left operand (rows of `a` sum to 1: max|z| <= max|x|)
    xscale = _C.halves_scale(h)
    A = torch.empty((N, 2 * KA), dtype=torch.float16, device=h.device)
    _C.halves_split_cols(h, xscale, 2, A, KA, 0, FP)
    _C.spmm_bcast_halves(csc, h, a, None, xscale, A, FP, FP, KA, FP)
    scale, shift = eval_affine(epi)
    y = torch.empty((N, HD), dtype=h.dtype, device=h.device)
    slots = _C.absmax_slots(h.device) if (relu and ABSMAX_BYPRODUCT and gemm.enabled(h)) else None
    _C.gemm_halves3_nt_grouped(A, B, xscale, wscale, KA, 2 * FP, y, g_fwd, FP // 32, col_scale=scale, col_shift=shift, relu=relu, absmax=slots)
    if slots is not None:
        gemm.stash_scale(y, _C.halves_scale_from_slots(slots))
    return y


@torch.no_grad()
def gat_infer_layer(conv, epi, graph, h, relu, first=False):
    """Eval-mode `act(epi(conv(graph, h).flatten(1)))` (models.py:716-731 with dropout off; for the output layer
    `biases[-1](conv(graph, h).mean(1))`, models.py:733-734, one head) as one GEMM + one fused sweep.  h: [N, Fin] -> [N, H*D]."""
    from . import has_zero_in_degree
    if not conv._allow_zero_in_degree:
        assert not has_zero_in_degree(graph), "0-in-degree nodes (models.py:477-479)"
    global INFER_CALLS
    INFER_CALLS += 1
    H, D = conv._num_heads, conv._out_feats
    HD, N = H * D, h.shape[0]
    has_res, has_er = conv.res_fc is not None, conv.attn_r is not None
    W = infer_weight(conv)                                              # [K, P] or [P, K]
    ctx = _LabelReuse.active
    F0 = ctx.static_cols if (ctx is not None and first and 0 < ctx.static_cols < h.shape[1]) else 0
    if _infer_l0_ok(conv, graph, h, F0):
        return _infer_l0(conv, epi, graph, h, relu)
    if F0:
        Wk = W if WEIGHT_KP else W.t()
        key = (id(conv), h.data_ptr(), tuple(h.shape)) + _versions(W)   # same layer, same input buffer, same weights
        if ctx.store.get("key") != key:                                 # the static columns' share of the projection
            ctx.store["key"], ctx.store["base"] = key, torch.mm(h[:, :F0], Wk[:F0])
        out = torch.addmm(ctx.store["base"], h[:, F0:], Wk[F0:])
    elif gemm.enabled(h):                                               # fp32 GEMM on the fp16 matrix cores (bot_amd.gemm)
        # the weight's halves live and die with the merged weight they were split from: same key (the parameters' versions)
        out = gemm.mm_nt(gemm.split_with_stash(h, 0), _cached(conv, "infer_halves", _infer_key(conv),
                                                   lambda: gemm.split_right(W.t().contiguous() if WEIGHT_KP else W)))
    else:
        out = torch.mm(h, W) if WEIGHT_KP else torch.mm(h, W.t())       # [N, P] = [ft | res | el | er | pad]
    B = block_width(HD)
    c = 2 * B if has_res else B
    if graph.halo is not None:                                          # partitioned: owned + halo source rows [ft | el]
        ext = _extend_forward(graph, out, HD, H, c)
        ft, el = ext[:, :HD], ext[:, HD:HD + H]
    else:
        ft, el = out[:, :HD], out[:, c:c + H]
    er = out[:, c + H:c + 2 * H] if has_er else None
    ew = None
    if conv._use_symmetric_norm:                                        # models.py:500-505, :550-555 folded into the edge weights
        s_out, w_e = sym_scales(graph)
        el = el * s_out.unsqueeze(1)
        ew = w_e
    scale, shift = eval_affine(epi)
    res = out[:, B:B + HD].unflatten(1, (H, D)) if has_res else None
    # a hidden layer's output is the next layer's GEMM operand: the sweep delivers max|y| as it stores y (no pass for the halves' scale)
    slots = _C.absmax_slots(h.device) if (relu and ABSMAX_BYPRODUCT and gemm.enabled(h)) else None
    y = _C.gat_infer(graph.csc, ft.unflatten(1, (H, D)), el, er, None, ew, conv.leaky_relu.negative_slope, addend=res,
                     scale=scale, shift=shift, relu=relu, absmax=slots)
    y = y.view(N, HD)
    if slots is not None:
        gemm.stash_scale(y, _C.halves_scale_from_slots(slots))
    return y
