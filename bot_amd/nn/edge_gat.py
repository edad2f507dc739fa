"""GAT with edge features — the layer of src/ogbn-proteins/models.py:19-168 (identical in
src/ogbn-products/models.py:20-167) and the two stacks built from it, on the full graph
(the `not isinstance(g, list)` branch, ogbn-proteins/models.py:231-234): attention scores come from
linear maps of the INPUT features (`attn_src_fc`, `attn_dst_fc`) plus `attn_edge_fc` of the edge
embedding; the destination branch `dst_fc` (with bias) is the residual.

The edge term enters the fused attention kernel as a per-edge logit in edge-id order
(`bot_amd.ops.gat_attention(..., ee=...)`), so no [E,H] intermediate beyond it is materialised.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import halo, ops
from ..graph import take_rows
from . import _bcast, _epilogue, _pair, degree_norm, has_zero_in_degree
from . import fused as _fused

__all__ = ["GATConv", "ProteinsGAT", "ProductsGAT"]


class GATConv(nn.Module):
    def __init__(self, node_feats, edge_feats, out_feats, n_heads=1, attn_drop=0.0, edge_drop=0.0, negative_slope=0.2,
                 residual=True, activation=None, use_attn_dst=True, allow_zero_in_degree=True, use_symmetric_norm=False):
        super().__init__()
        self._n_heads = n_heads
        self._in_src_feats, self._in_dst_feats = _pair(node_feats)
        self._out_feats = out_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        self._use_symmetric_norm = use_symmetric_norm
        self.src_fc = nn.Linear(self._in_src_feats, out_feats * n_heads, bias=False)
        if residual:
            self.dst_fc = nn.Linear(self._in_src_feats, out_feats * n_heads)
            self.bias = None
        else:  # the reference's `nn.Parameter(int)` here cannot run; give it the evident meaning
            self.dst_fc = None
            self.bias = nn.Parameter(torch.zeros(out_feats * n_heads))
        self.attn_src_fc = nn.Linear(self._in_src_feats, n_heads, bias=False)
        self.attn_dst_fc = nn.Linear(self._in_src_feats, n_heads, bias=False) if use_attn_dst else None
        self.attn_edge_fc = nn.Linear(edge_feats, n_heads, bias=False) if edge_feats > 0 else None
        self.attn_drop = nn.Dropout(attn_drop)
        self.edge_drop = edge_drop
        self.leaky_relu = nn.LeakyReLU(negative_slope, inplace=True)
        self.activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        for lin in (self.src_fc, self.dst_fc, self.attn_src_fc, self.attn_dst_fc, self.attn_edge_fc):
            if lin is not None:
                nn.init.xavier_normal_(lin.weight, gain=gain)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    merge_projections = True   # one GEMM for the four Linears on the layer input (full-graph / partitioned blocks alike)

    def _infer(self, graph, ft, attn_src, attn_dst, res, feat_edge, edge_encoder, epilogue):
        from .. import _C
        H = self._n_heads
        csc = graph.csc
        ee = None
        if feat_edge is not None and edge_encoder is not None:
            ee = _C.edge_mlp_fwd(ops.edge_features_csc(graph, feat_edge), edge_encoder.weight, edge_encoder.bias, self.attn_edge_fc.weight)
        elif feat_edge is not None:
            ee = take_rows(self.attn_edge_fc(feat_edge).view(-1, H), csc.eid)          # edge-id order -> position order
        scale, shift, relu = epilogue if epilogue is not None else (None, None, False)
        _fused.count_infer()
        return _C.gat_infer(csc, ft, attn_src.reshape(-1, H), None if attn_dst is None else attn_dst.reshape(-1, H), ee, None,
                            self.leaky_relu.negative_slope, addend=res, scale=scale, shift=shift, relu=relu)

    def forward(self, graph, feat_src, feat_edge=None, keep=None, edge_encoder=None, epilogue=None):
        """`edge_encoder` (the stack's nn.Linear(8 -> 16)) given: `feat_edge` holds the RAW edge features and the
        encoder + ReLU + attn_edge_fc run fused per edge (bot_amd.ops.edge_mlp); otherwise `feat_edge` is the embedding."""
        if not self._allow_zero_in_degree:
            assert not has_zero_in_degree(graph), "0-in-degree nodes (ogbn-proteins/models.py:89-91)"
        H, D = self._n_heads, self._out_feats
        feat_dst = feat_src
        res = None
        if self._use_symmetric_norm or not self.merge_projections:
            if self._use_symmetric_norm:
                feat_src = feat_src * _bcast(degree_norm(graph, "out", -0.5), feat_src)
            ft = self.src_fc(feat_src).view(-1, H, D)
            attn_src = self.attn_src_fc(feat_src).view(-1, H, 1)
            attn_dst = self.attn_dst_fc(feat_dst).view(-1, H, 1) if self.attn_dst_fc is not None else None
        else:
            # src_fc, dst_fc, attn_src_fc and attn_dst_fc all read the same input (ogbn-proteins/models.py:107-124): ONE GEMM on
            # the concatenated weights.  Its backward is one dh GEMM, where four separate Linears leave autograd three
            # [N, H*D]-sized gradient additions per layer (8.4 ms per layer at S-products); the residual goes into the SpMM epilogue.
            parts, biases = [self.src_fc.weight], [None]
            if self.dst_fc is not None:
                parts.append(self.dst_fc.weight), biases.append(self.dst_fc.bias)
            parts.append(self.attn_src_fc.weight), biases.append(None)
            if self.attn_dst_fc is not None:
                parts.append(self.attn_dst_fc.weight), biases.append(None)
            # only dst_fc has a bias: it is added to ITS columns after the GEMM — a bias vector over the whole merged output (zeros
            # elsewhere) was a 19 GB element-wise pass per layer at S-products for 480 of 968 columns that needed it
            sizes = [w.shape[0] for w in parts]
            pieces = None
            if feat_src.dim() == 2 and (feat_src.is_cuda or _fused.FORCE):
                from .. import gemm
                pieces = gemm.linear_blocks(feat_src, torch.cat(parts), sizes)          # halves path: the blocks' gradients never meet in a `cat`
                if pieces is None:
                    pieces = torch.split(ops.linear(feat_src, torch.cat(parts)), sizes, dim=1)
            else:
                pieces = torch.split(F.linear(feat_src, torch.cat(parts)), sizes, dim=1)
            pieces = list(pieces)
            pieces = [p if b is None else ops.add_bias(p, b) for p, b in zip(pieces, biases)]
            ft = pieces.pop(0).unflatten(1, (H, D))
            if self.dst_fc is not None:
                res = pieces.pop(0).unflatten(1, (H, D))
            attn_src = pieces.pop(0).unsqueeze(-1)
            attn_dst = pieces.pop(0).unsqueeze(-1) if self.attn_dst_fc is not None else None
        infer = (not torch.is_grad_enabled() and not self.training and res is not None and (ft.is_cuda or _fused.FORCE) and H <= 8
                 and self.activation is None and _fused.sweep_is_row_kernel(graph, H, D))
        # partitioned mode: the halo rows of `attn_src` (small) arrive here; those of `ft` travel while the edge logits and the
        # attention are formed and the owned-source edges are swept (bot_amd.halo), or in one exchange here (inference sweep,
        # BOT_HALO_OVERLAP=0)
        transfer = halo.start(graph, ft) if (halo.enabled(graph) and not infer) else None
        if transfer is None:
            ft = graph.extend(ft)
        attn_src = graph.extend(attn_src)
        if infer:
            # inference (evaluate(): eval mode under no_grad, ogbn-proteins/gat.py:136-160): logits + softmax + aggregation +
            # dst_fc residual (+ the stack's eval-mode BatchNorm and ReLU when it hands them in) in ONE sweep, nothing
            # edge-sized written beyond the edge logits of the edge-feature term (bot_gat_infer_f32)
            return self._infer(graph, ft, attn_src, attn_dst, res, feat_edge, edge_encoder, epilogue)
        ee, ee_order = None, "eid"
        if feat_edge is not None and edge_encoder is not None:
            ee = ops.edge_mlp(graph, feat_edge, edge_encoder.weight, edge_encoder.bias, self.attn_edge_fc.weight).view(-1, H, 1)
            ee_order = "csc"
        elif feat_edge is not None:
            ee = self.attn_edge_fc(feat_edge).view(-1, H, 1)
        keep_order = "eid"
        if keep is None and self.training and self.edge_drop > 0:
            # a uniformly random edge subset is uniformly random in any fixed edge order: use the mask as CSC-ordered whenever the
            # edge logits are (or there are none), which saves permuting E bytes per layer
            keep = ops.random_edge_keep(graph, self.edge_drop)
            keep_order = "csc" if (ee is None or ee_order == "csc") else "eid"
        a = ops.gat_attention(graph, attn_src, attn_dst, ee, keep=keep, negative_slope=self.leaky_relu.negative_slope,
                              order="csc", ee_order=ee_order, keep_order=keep_order)
        if transfer is None:
            rst = ops.u_mul_e_sum(graph, ft, self.attn_drop(a), order="csc", addend=res)
        else:
            rst = halo.u_mul_e_sum(graph, ft, self.attn_drop(a), addend=res, transfer=transfer)
        if self._use_symmetric_norm:
            rst = rst * _bcast(degree_norm(graph, "in", 0.5), rst)
        if res is not None:
            pass                                             # dst_fc(feat_dst) was added in the SpMM epilogue
        elif self.dst_fc is not None:
            rst = rst + self.dst_fc(feat_dst).view(-1, H, D)
        else:
            rst = rst + self.bias.view(1, H, D)
        if self.activation is not None:
            rst = self.activation(rst, inplace=True)
        if epilogue is not None:   # the stack handed its eval-mode BatchNorm + ReLU in, but this call took the generic path
            scale, shift, relu = epilogue
            rst = rst.flatten(1) * scale + shift
            rst = (torch.relu(rst) if relu else rst).view(-1, H, D)
        return rst


class _EdgeGAT(nn.Module):
    """Shared body of the proteins / products stacks (ogbn-proteins/models.py:171-264, ogbn-products/models.py:170-265)."""

    def __init__(self, node_feats, edge_feats, n_classes, n_layers, n_heads, n_hidden, edge_emb, activation, dropout,
                 input_drop, attn_drop, edge_drop, use_attn_dst, allow_zero_in_degree, first_in):
        super().__init__()
        self.n_layers, self.n_heads, self.n_hidden, self.n_classes = n_layers, n_heads, n_hidden, n_classes
        self.convs, self.norms = nn.ModuleList(), nn.ModuleList()
        self.node_encoder = nn.Linear(node_feats, n_hidden)
        self.edge_encoder = nn.ModuleList() if edge_emb > 0 else None
        for i in range(n_layers):
            if self.edge_encoder is not None:
                self.edge_encoder.append(nn.Linear(edge_feats, edge_emb))
            self.convs.append(GATConv(n_heads * n_hidden if i > 0 else first_in, edge_emb, n_hidden, n_heads=n_heads,
                                      attn_drop=attn_drop, edge_drop=edge_drop, use_attn_dst=use_attn_dst,
                                      allow_zero_in_degree=allow_zero_in_degree, use_symmetric_norm=False))
            self.norms.append(nn.BatchNorm1d(n_heads * n_hidden))
        self.pred_linear = nn.Linear(n_heads * n_hidden, n_classes)
        self.input_drop, self.dropout = nn.Dropout(input_drop), nn.Dropout(dropout)
        self.activation = activation

    fuse_edge_mlp = True

    def _fusable_edge_mlp(self, i, efeat):
        from .. import _C
        enc, conv = self.edge_encoder[i], self.convs[i]
        return (self.fuse_edge_mlp and conv.attn_edge_fc is not None and not efeat.requires_grad and enc.bias is not None
                and _C.edge_mlp_supported(enc.in_features, enc.out_features, conv._n_heads))

    def _body(self, g, h, residual):
        h = self.input_drop(g.to_internal(h))   # node features arrive in original order; edge features are in edge-id order
        h_last = None
        efeat = g.edata.get("feat") if self.edge_encoder is not None else None
        infer = (not torch.is_grad_enabled() and not self.training and (h.is_cuda or _fused.FORCE)
                 and self.activation in (F.relu, torch.relu))
        for i in range(self.n_layers):
            # evaluate(): without an inter-layer residual the eval-mode BatchNorm + ReLU ride in the layer's fused sweep too
            epi = None
            if infer and not residual and not self.norms[i].training and self.norms[i].track_running_stats and self.convs[i].dst_fc is not None:
                epi = _fused.eval_affine(self.norms[i]) + (True,)
            if efeat is not None and self._fusable_edge_mlp(i, efeat):
                h = self.convs[i](g, h, efeat, edge_encoder=self.edge_encoder[i], epilogue=epi).flatten(1, -1)  # f2: no [E,16] tensor exists
            else:
                emb = F.relu(self.edge_encoder[i](efeat), inplace=True) if efeat is not None else None
                h = self.convs[i](g, h, emb, epilogue=epi).flatten(1, -1)
            if epi is not None:
                continue
            if residual and h_last is not None:
                h = h + h_last[: h.shape[0], :]
            h_last = h
            h = _epilogue(h, self.norms[i], self.activation, self.dropout, self.training, halves=True)  # BatchNorm + ReLU + dropout (+ the next GEMM's operand), fused
        return g.to_original(ops.linear(h, self.pred_linear.weight, self.pred_linear.bias))


class ProteinsGAT(_EdgeGAT):
    """`GAT` of src/ogbn-proteins/models.py:171-264: node encoder + ReLU first, inter-layer residual always on."""

    def __init__(self, node_feats, edge_feats, n_classes, n_layers, n_heads, n_hidden, edge_emb, activation, dropout,
                 input_drop, attn_drop, edge_drop, use_attn_dst=True, allow_zero_in_degree=False):
        super().__init__(node_feats, edge_feats, n_classes, n_layers, n_heads, n_hidden, edge_emb, activation, dropout,
                         input_drop, attn_drop, edge_drop, use_attn_dst, allow_zero_in_degree, first_in=n_hidden)

    def forward(self, g):
        h = F.relu(self.node_encoder(g.srcdata["feat"]), inplace=True)
        return self._body(g, h, residual=True)


class ProductsGAT(_EdgeGAT):
    """`GAT` of src/ogbn-products/models.py:170-265: `node_encoder` is constructed but not applied (:198, :237-239
    absent), the inter-layer residual is a flag."""

    def __init__(self, node_feats, edge_feats, n_classes, n_layers, n_heads, n_hidden, edge_emb, activation, dropout,
                 input_drop, attn_drop, edge_drop, use_attn_dst=True, allow_zero_in_degree=False, residual=False):
        super().__init__(node_feats, edge_feats, n_classes, n_layers, n_heads, n_hidden, edge_emb, activation, dropout,
                         input_drop, attn_drop, edge_drop, use_attn_dst, allow_zero_in_degree, first_in=node_feats)
        self.residual = residual

    def forward(self, g, inference=False):
        return self._body(g, g.srcdata["feat"], residual=self.residual)
