"""TEST INFRASTRUCTURE ONLY — functional CPU restatement of the reference's layer/stack logic.

PINNED: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which
were produced by running the reference's own modules (see oracle/gen_golden.py).  This file exists
because /root/reference is not present on the GPU box: `-m gpu` tests compare the HIP path with
these functions on seeded inputs larger than the committed fixtures.

All functions are pure: parameters come in as a dict keyed like the reference's `state_dict()`
(e.g. "convs.0.fc.weight"), graphs as `CooGraph`.  Dropout is not restated (parity runs use rate 0
or eval mode, SURVEY §7 "Edge-id order is user-visible").
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import ref_ops as R


@dataclass
class CooGraph:
    src: torch.Tensor  # int64 [E], edge-id order
    dst: torch.Tensor  # int64 [E]
    num_nodes: int

    @property
    def num_edges(self):
        return int(self.src.numel())


def _is_c(g):
    return type(g).__name__ == "CGraph"  # oracle/c_ops.py: same semantics on the C restatement (timed CPU baseline)


def _copy_u_sum(g, x):
    if _is_c(g):
        from . import c_ops
        return c_ops.copy_u_sum(g, x)
    return R.copy_u_sum(g.src, g.dst, g.num_nodes, x)


def _u_mul_e_sum(g, x, a):
    if _is_c(g):
        from . import c_ops
        return c_ops.u_mul_e_sum(g, x, a)
    return R.u_mul_e_sum(g.src, g.dst, g.num_nodes, x, a)


def _u_add_v(g, x, y):
    if _is_c(g):
        from . import c_ops
        return c_ops.u_add_v(g, x, y)
    return R.copy_u(g.src, x) if y is None else R.u_add_v(g.src, g.dst, x, y)


def _edge_softmax(g, e, keep_eids=None):
    """All-edges softmax, or (models.py:534-539) zeros scattered with the softmax over the kept edges."""
    if _is_c(g):
        from . import c_ops
        keep = None
        if keep_eids is not None:
            keep = torch.zeros(g.num_edges, dtype=torch.uint8)
            keep[keep_eids] = 1
        return c_ops.edge_softmax(g, e, keep)
    if keep_eids is None:
        return R.edge_softmax(g.dst, g.num_nodes, e)
    return torch.zeros_like(e).index_put((keep_eids,), R.edge_softmax(g.dst, g.num_nodes, e[keep_eids], keep_eids))


class ZeroInDegreeError(Exception):
    """Counterpart of the DGLError / `assert False` raised at models.py:334-346 and :477-479."""


def _deg_pow(deg: torch.Tensor, p: float, like: torch.Tensor) -> torch.Tensor:
    # models.py:352-355 / 388-394 / 501-504 / 551-554: float(), clamp(min=1), pow, reshape to broadcast
    # NB the reference calls `.float()`: the norm vector is float32 even for float64 features.
    d = deg.float().clamp(min=1)
    return torch.pow(d, p).reshape((-1,) + (1,) * (like.dim() - 1))


def graphconv_forward(g: CooGraph, feat, weight, bias, norm="both", activation=None,
                      allow_zero_in_degree=False):
    """GraphConv.forward — src/no-sampling/models.py:287-403."""
    in_deg = R.in_degrees(g.dst, g.num_nodes)
    if not allow_zero_in_degree and bool((in_deg == 0).any()):  # :334-346
        raise ZeroInDegreeError("There are 0-in-degree nodes in the graph")
    h = feat
    if norm == "both":  # :351-356
        h = h * _deg_pow(R.out_degrees(g.src, g.num_nodes), -0.5, h)
    w_first = weight is not None and weight.shape[0] > weight.shape[1]  # :368 in_feats > out_feats
    if w_first:  # :368-376
        h = torch.matmul(h, weight)
    rst = _copy_u_sum(g, h)  # :374 / :381
    if not w_first and weight is not None:  # :384-385
        rst = torch.matmul(rst, weight)
    if norm == "both":  # :387-395
        rst = rst * _deg_pow(in_deg, -0.5, rst)
    elif norm == "right":
        rst = rst * (1.0 / in_deg.float().clamp(min=1)).reshape((-1,) + (1,) * (rst.dim() - 1))
    if bias is not None:  # :397-398
        rst = rst + bias
    if activation is not None:
        rst = activation(rst)
    return rst


def gatconv_forward(g: CooGraph, feat, fc_weight, attn_l, attn_r=None, res_fc_weight=None, *,
                    num_heads, out_feats, negative_slope=0.2, use_symmetric_norm=False,
                    keep_eids=None, allow_zero_in_degree=False, activation=None, return_attention=False, leaky=None):
    """GATConv.forward (non-tuple, non-block branch) — src/no-sampling/models.py:475-566.

    `keep_eids` restates the training-time edge-drop branch :528-539 for a *given* kept-edge set
    (the reference draws it with randperm).  `leaky` (test infrastructure, default None = F.leaky_relu): a callable
    (e, slope) -> leaky_relu(e) that may take the 0/1 side of every logit from outside (tests/full_size.py:KinkGates)."""
    n = g.num_nodes
    if not allow_zero_in_degree and bool((R.in_degrees(g.dst, n) == 0).any()):  # :477-479
        raise ZeroInDegreeError("0-in-degree nodes")
    ft = F.linear(feat, fc_weight).view(-1, num_heads, out_feats)  # :490-492
    ft_dst = ft  # :498 — bound BEFORE the symmetric scaling below, so `er` sees the unscaled projection
    if use_symmetric_norm:  # :500-505
        ft = ft * _deg_pow(R.out_degrees(g.src, n), -0.5, ft)
    el = (ft * attn_l).sum(dim=-1).unsqueeze(-1)  # :517
    if attn_r is not None:  # :520-523
        er = (ft_dst * attn_r).sum(dim=-1).unsqueeze(-1)
        e = _u_add_v(g, el, er)
    else:  # :525
        e = _u_add_v(g, el, None)
    e = F.leaky_relu(e, negative_slope) if leaky is None else leaky(e, negative_slope)  # :526
    a = _edge_softmax(g, e, keep_eids)  # :528-539 (edge drop) / :544
    rst = _u_mul_e_sum(g, ft, a)  # :547-548
    if use_symmetric_norm:  # :550-555  (+0.5)
        rst = rst * _deg_pow(R.in_degrees(g.dst, n), 0.5, rst)
    if res_fc_weight is not None:  # :558-560
        rst = rst + F.linear(feat, res_fc_weight).view(feat.shape[0], -1, out_feats)
    if activation is not None:  # :563-564
        rst = activation(rst)
    return (rst, a) if return_attention else rst


def _bn(h, sd, prefix, training, eps=1e-5):
    # nn.BatchNorm1d over the node dimension — models.py:609 / :698; biased variance in training.
    return F.batch_norm(h, sd[prefix + ".running_mean"].clone(), sd[prefix + ".running_var"].clone(),
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training=training, eps=eps)


def gcn_forward(g: CooGraph, feat, sd: dict, *, n_layers, norm="none", norm_adj="symm",
                use_linear=False, residual=False, activation=F.relu, training=False):
    """GCN.forward — src/no-sampling/models.py:616-641 (dropout rates 0)."""
    h, h_last = feat, None
    conv_norm = "both" if norm_adj == "symm" else "right"  # :603
    for i in range(n_layers):
        conv = graphconv_forward(g, h, sd[f"convs.{i}.weight"], sd.get(f"convs.{i}.bias"), conv_norm)
        if use_linear:  # :625-627
            conv = conv + F.linear(h, sd[f"linear.{i}.weight"])
        h = conv
        if i < n_layers - 1:
            if residual and h_last is not None:  # :632-634
                h = h + h_last
            h_last = h
            if norm == "batch":  # :636-637
                h = _bn(h, sd, f"norms.{i}", training)
            h = activation(h)
    return h


def gat_forward(g: CooGraph, feat, sd: dict, *, n_layers, n_heads, n_hidden, n_classes, norm="none",
                non_interactive_attn=False, use_symmetric_norm=False, linear=False, residual=False,
                activation=F.relu, training=False, keep_eids=None, leaky=None):
    """GAT.forward — src/no-sampling/models.py:709-736 (dropout rates 0)."""
    h, h_last = feat, None
    for i in range(n_layers):
        heads = n_heads if i < n_layers - 1 else 1  # :679-681
        out = n_hidden if i < n_layers - 1 else n_classes
        h = gatconv_forward(
            g, h, sd[f"convs.{i}.fc.weight"], sd[f"convs.{i}.attn_l"],
            sd[f"convs.{i}.attn_r"] if non_interactive_attn else None,
            sd[f"convs.{i}.res_fc.weight"] if linear else None,
            num_heads=heads, out_feats=out, use_symmetric_norm=use_symmetric_norm,
            keep_eids=None if keep_eids is None else keep_eids[i], leaky=leaky)
        if i < n_layers - 1:
            if residual and h_last is not None:  # :721-723
                h = h + h_last
            h_last = h
            h = h.flatten(1)  # :725
            if norm == "batch":  # :726-727
                h = _bn(h, sd, f"norms.{i}", training)
            else:  # :728-729
                h = h + sd[f"biases.{i}.bias"]
            h = activation(h)
    h = h.mean(1)  # :733
    last_bias = sd[f"biases.{n_layers - 1}.bias"] if norm != "batch" else sd["biases.0.bias"]  # :696-702,734
    return h + last_bias


class _LinearF64Grad(torch.autograd.Function):
    """y = x W^T + b exactly as F.linear computes it (fp32); the WEIGHT and BIAS gradients — reductions over all rows of x —
    accumulated in fp64.  For full-size parity runs only: at N = 2.45 M rows torch's CPU sgemm is itself 2.1e-4 (of the largest
    entry) away from the fp64 product of its own operands (tests/diag_products_dw.py), which is above the 1e-4 the
    comparison allows; this removes the ORACLE's reduction noise and nothing else."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gw = (gy.double().t() @ x.double()).to(w.dtype)
        gb = gy.double().sum(0).to(w.dtype) if ctx.has_b else None
        return gx, gw, gb


def linear_f64grad(x, w, b=None):
    return _LinearF64Grad.apply(x, w, b)


def proteins_gatconv_forward(g: CooGraph, feat_src, sd: dict, prefix: str, *, n_heads, out_feats,
                             feat_edge=None, negative_slope=0.2, keep_eids=None, activation=None, leaky=None, linear=F.linear):
    """ogbn-proteins GATConv.forward — src/ogbn-proteins/models.py:87-168 (full-graph branch,
    `use_symmetric_norm=False` as constructed at :219); same layer in ogbn-products/models.py:88-167.
    `leaky`: see gatconv_forward; `linear`: F.linear or `linear_f64grad` (both test infrastructure)."""
    n = g.num_nodes
    p = lambda k: sd[f"{prefix}{k}"]
    ft = linear(feat_src, p("src_fc.weight")).view(-1, n_heads, out_feats)  # :106
    res = linear(feat_src, p("dst_fc.weight"), p("dst_fc.bias")).view(-1, n_heads, out_feats)  # :107
    a_src = linear(feat_src, p("attn_src_fc.weight")).view(-1, n_heads, 1)  # :108
    if f"{prefix}attn_dst_fc.weight" in sd:  # :122-125
        a_dst = linear(feat_src, p("attn_dst_fc.weight")).view(-1, n_heads, 1)
        e = _u_add_v(g, a_src, a_dst)
    else:  # :127
        e = _u_add_v(g, a_src, None)
    if feat_edge is not None:  # :130-133
        e = e + linear(feat_edge, p("attn_edge_fc.weight")).view(-1, n_heads, 1)
    e = F.leaky_relu(e, negative_slope) if leaky is None else leaky(e, negative_slope)  # :134
    a = _edge_softmax(g, e, keep_eids)  # :136-141 (edge drop) / :143
    rst = _u_mul_e_sum(g, ft, a)  # :146-148
    rst = rst + res  # :159-160
    if activation is not None:
        rst = activation(rst)
    return rst


def proteins_gat_forward(g: CooGraph, node_feat, edge_feat, sd: dict, *, n_layers, n_heads, n_hidden,
                         training=False, use_node_encoder=True, residual=True, activation=F.relu, leaky=None,
                         linear=F.linear):
    """ogbn-proteins GAT.forward full-graph branch — src/ogbn-proteins/models.py:230-264
    (`use_node_encoder=False, residual=<flag>` gives ogbn-products/models.py:233-265).  `activation` is the stack's
    (the reference passes F.relu, gat.py:100); `activation` / `leaky` may be tests/full_size.py:KinkGates hooks."""
    h = node_feat
    if use_node_encoder:  # :237-239
        h = F.relu(linear(h, sd["node_encoder.weight"], sd["node_encoder.bias"]))
    h_last = None
    for i in range(n_layers):
        ee = None
        if edge_feat is not None and f"edge_encoder.{i}.weight" in sd:  # :244-248
            ee = F.relu(linear(edge_feat, sd[f"edge_encoder.{i}.weight"], sd[f"edge_encoder.{i}.bias"]))
        h = proteins_gatconv_forward(g, h, sd, f"convs.{i}.", n_heads=n_heads, out_feats=n_hidden,
                                     feat_edge=ee, leaky=leaky, linear=linear).flatten(1, -1)  # :251
        if residual and h_last is not None:  # :253-254
            h = h + h_last
        h_last = h
        h = _bn(h, sd, f"norms.{i}", training)  # :258
        h = activation(h)
    return linear(h, sd["pred_linear.weight"], sd["pred_linear.bias"])  # :262


# ---------------------------------------------------------------------------------------------
# callers of the hot path — src/no-sampling/run.py
# ---------------------------------------------------------------------------------------------

EPSILON = 1 - math.log(2)  # run.py:34


def add_labels(feat, labels, idx, n_classes):
    """add_labels — run.py:240-243."""
    onehot = torch.zeros([feat.shape[0], n_classes], dtype=feat.dtype)
    onehot[idx, labels[idx, 0]] = 1
    return torch.cat([feat, onehot], dim=-1)


def compute_loss(x, labels, loss="logit"):
    """compute_loss — run.py:229-237."""
    y = F.cross_entropy(x, labels[:, 0], reduction="none")
    if loss == "loge":
        y = torch.log(EPSILON + y) - math.log(EPSILON)
    elif loss == "savage":
        y = (1 - torch.exp(-y)) ** 2
    elif loss != "logit":
        raise ValueError(loss)
    return torch.mean(y)


def warmup_lr(lr, epoch):
    """adjust_learning_rate — run.py:246-249 (returns None when the epoch leaves lr untouched)."""
    return lr * epoch / 50 if epoch <= 50 else None
