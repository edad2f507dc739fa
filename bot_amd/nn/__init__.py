"""Layer and stack modules of the full-batch path, interface-identical to the reference's
src/no-sampling/models.py (constructor arguments and defaults, attribute names, `state_dict` keys,
`forward(graph, feat)`), so the paper's tricks and the training loop run unchanged — with the DGL
sparse operators underneath replaced by the gfx950 kernels in `bot_amd.ops`.

Differences are internal only: the attention weights stay in CSC position order between the fused
attention op and the aggregation (no edge-id permutation gathers), per-graph invariants (degree
norms, the zero-in-degree check) are computed once per graph instead of once per call.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import gemm, halo, ops
from ..errors import DGLError

__all__ = ["ElementWiseLinear", "GraphConv", "GATConv", "GCN", "GAT"]


def _pair(x):
    return x if isinstance(x, tuple) else (x, x)


def _graph_cache(graph):
    c = getattr(graph, "_bot_cache", None)
    if c is None:
        c = graph._bot_cache = {}
    return c


def has_zero_in_degree(graph) -> bool:
    """`(graph.in_degrees() == 0).any()` (models.py:335,478), evaluated once per graph: it is an
    integer property of the structure and costs a device->host sync."""
    c = _graph_cache(graph)
    if "zero_in" not in c:
        c["zero_in"] = bool((graph.in_degrees() == 0).any())
    return c["zero_in"]


def degree_norm(graph, which: str, power: float) -> torch.Tensor:
    """`degs.float().clamp(min=1) ** power` as float32 [N] (models.py:352-353, 388-392, 501-502, 551-552)."""
    c = _graph_cache(graph)
    key = (which, power)
    if key not in c:
        if which == "out":  # degrees of the nodes whose features the caller holds (owned nodes in partitioned mode)
            deg = graph.global_out_degrees if graph.global_out_degrees is not None else graph.out_degrees()
        else:
            deg = graph.in_degrees()
        d = deg.float().clamp(min=1)
        c[key] = 1.0 / d if power == -1.0 else torch.pow(d, power)
    return c[key]


def _epilogue(h, norm, activation, dropout, training, halves=False):
    """`dropout(activation(norm(h)))` — models.py:636-639 / :726-731.  BatchNorm1d + ReLU (+ dropout) run as the
    fused HIP epilogue (2 reads + 1 write instead of ~10 round trips); other activations take the stock ops.
    `halves`: the result feeds a projection that runs on fp16 halves (bot_amd.gemm): write them in the same pass."""
    if isinstance(norm, nn.BatchNorm1d) and activation in (F.relu, torch.relu) and h.dim() == 2:
        return ops.bn_relu_dropout(h, norm, relu=True, p=dropout.p, training=training, halves=halves)
    return dropout(activation(norm(h)))


def _bcast(norm, like):
    return norm.reshape(norm.shape + (1,) * (like.dim() - 1))


class ElementWiseLinear(nn.Module):
    """Per-feature scale and/or shift — models.py:18-50."""

    def __init__(self, size, weight=True, bias=True, inplace=False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(size)) if weight else None
        self.bias = nn.Parameter(torch.zeros(size)) if bias else None
        self.inplace = inplace

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.ones_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x):
        if self.inplace:
            if self.weight is not None:
                x.mul_(self.weight)
            if self.bias is not None:
                x.add_(self.bias)
            return x
        if self.weight is not None:
            x = x * self.weight
        if self.bias is not None:
            # on the GPU the bias gradient — a column sum over all N rows — comes from the library's deterministic two-stage kernel:
            # more accurate than the stock reduction (finished in double) and, unlike it, correct under hipGraph replay (ops._SumAll)
            x = ops.add_bias(x, self.bias) if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32) else x + self.bias
        return x


class GraphConv(nn.Module):
    """GCN layer — models.py:114-413.  out = D_in^{-1/2} A D_out^{-1/2} X W (+ b) for norm='both',
    D_in^{-1} A X W for 'right', A X W for 'none'; W is applied before the aggregation when it narrows
    the features (in_feats > out_feats), after it otherwise."""

    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False):
        super().__init__()
        if norm not in ("none", "both", "right"):
            raise DGLError(f'Invalid norm value. Must be either "none", "both" or "right". But got "{norm}".')
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        if weight:
            self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        else:
            self.register_parameter("weight", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_feats))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._activation = activation

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def forward(self, graph, feat, weight=None):
        if not self._allow_zero_in_degree and has_zero_in_degree(graph):
            raise DGLError(
                "There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                "Adding self-loop on the input graph by calling `g = g.add_self_loop()` will resolve the issue. "
                "Setting ``allow_zero_in_degree`` to be `True` when constructing this module will suppress the check.")
        if isinstance(feat, tuple):
            raise NotImplementedError("bipartite (block) inputs belong to the sampled scripts, outside the full-batch path")
        if weight is not None and self.weight is not None:
            raise DGLError("External weight is provided while at the same time the module has defined its own weight "
                           "parameter. Please create the module with flag weight=False.")
        w = self.weight if weight is None else weight
        h = feat
        if self._norm == "both":
            h = h * _bcast(degree_norm(graph, "out", -0.5), h)
        # graph.extend: identity on one GPU; in partitioned mode the halo rows arrive here, after the
        # narrowing GEMM when there is one (so the narrower tensor is what crosses xGMI)
        # (bot_amd.halo: the exchange runs beside the sweep over the owned-source edges)
        agg = (lambda t: halo.copy_u_sum(graph, t)) if halo.enabled(graph) else (lambda t: ops.copy_u_sum(graph, graph.extend(t)))
        if self._in_feats > self._out_feats:
            if w is not None:
                h = gemm.matmul(h, w)
            rst = agg(h)
        else:
            rst = agg(h)
            if w is not None:
                rst = gemm.matmul(rst, w)
        if self._norm == "both":
            rst = rst * _bcast(degree_norm(graph, "in", -0.5), rst)
        elif self._norm == "right":
            rst = rst * _bcast(degree_norm(graph, "in", -1.0), rst)
        if self.bias is not None:
            rst = ops.add_bias(rst, self.bias)   # bias gradient = a column sum over N rows: colstats kernel, not the stock reduction
        if self._activation is not None:
            rst = self._activation(rst)
        return rst

    def extra_repr(self):
        s = f"in={self._in_feats}, out={self._out_feats}, normalization={self._norm}"
        if "_activation" in self.__dict__:
            s += f", activation={self._activation}"
        return s


class GATConv(nn.Module):
    """GAT layer with the paper's options — models.py:416-566.

    Note the reference's flag semantics (models.py:444-447): `non_interactive_attn=True` is what
    CREATES `attn_r` (scores use source and destination); the default (False) scores from the source
    only.  With `use_symmetric_norm` the result is D_in^{+1/2} · softmax-attention · D_out^{-1/2}
    (models.py:500-505, 550-555), and `attn_r` sees the projection BEFORE that scaling (models.py:498)."""

    def __init__(self, in_feats, out_feats, num_heads=1, feat_drop=0.0, attn_drop=0.0, edge_drop=0.0,
                 negative_slope=0.2, linear=True, activation=None, allow_zero_in_degree=False,
                 use_symmetric_norm=False, non_interactive_attn=False):
        super().__init__()
        self._num_heads = num_heads
        self._in_src_feats, self._in_dst_feats = _pair(in_feats)
        self._out_feats = out_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        self._use_symmetric_norm = use_symmetric_norm
        if isinstance(in_feats, tuple):
            self.fc_src = nn.Linear(self._in_src_feats, out_feats * num_heads, bias=False)
            self.fc_dst = nn.Linear(self._in_dst_feats, out_feats * num_heads, bias=False)
        else:
            self.fc = nn.Linear(self._in_src_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        if non_interactive_attn:
            self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        else:
            self.register_buffer("attn_r", None)
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.edge_drop = edge_drop
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        if linear:
            self.res_fc = nn.Linear(self._in_dst_feats, num_heads * out_feats, bias=False)
        else:
            self.register_buffer("res_fc", None)
        self.reset_parameters()
        self._activation = activation

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        for lin in (getattr(self, n, None) for n in ("fc", "fc_src", "fc_dst", "res_fc")):
            if isinstance(lin, nn.Linear):
                nn.init.xavier_normal_(lin.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        if isinstance(self.attn_r, nn.Parameter):
            nn.init.xavier_normal_(self.attn_r, gain=gain)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def _kept_edges(self, graph):
        """Training-time edge drop, models.py:528-532: a random subset of int(E*p) edges is left out."""
        return ops.random_edge_keep(graph, self.edge_drop)

    def forward(self, graph, feat, keep=None):
        """`keep` (uint8 [E], edge-id order) overrides the random edge-drop mask — used by parity tests."""
        if not self._allow_zero_in_degree:
            assert not has_zero_in_degree(graph), "0-in-degree nodes (models.py:477-479)"
        if isinstance(feat, tuple) or not hasattr(self, "fc"):
            raise NotImplementedError("bipartite (block) inputs belong to the sampled scripts, outside the full-batch path")
        H, D = self._num_heads, self._out_feats
        h = self.feat_drop(feat)
        W = self.fc.weight
        ft = F.linear(h, W).view(-1, H, D)
        # Attention scores without a pass over ft:  el[n,h] = <ft[n,h,:], attn_l[h,:]> = h[n,:] . (W_h^T attn_l[h])
        # (models.py:517, :521 — same value up to fp32 summation order; saves three 508 MB round trips per layer
        # forward and as many backward).  attn_r sees the projection BEFORE the symmetric scaling (models.py:498).
        Wh = W.view(H, D, -1)
        el = F.linear(h, (Wh * self.attn_l.view(H, D, 1)).sum(1))
        er = F.linear(h, (Wh * self.attn_r.view(H, D, 1)).sum(1)).unsqueeze(-1) if self.attn_r is not None else None
        if self._use_symmetric_norm:
            norm = degree_norm(graph, "out", -0.5)
            ft = ft * _bcast(norm, ft)
            el = el * norm.unsqueeze(-1)
        # partitioned mode: the halo rows of `el` (small) arrive here; those of `ft` travel while the attention is formed and the
        # owned-source edges are swept (bot_amd.halo), or — BOT_HALO_OVERLAP=0 — in one exchange here
        transfer = halo.start(graph, ft) if halo.enabled(graph) else None
        if transfer is None:
            ft = graph.extend(ft)  # identity on one GPU
        el = graph.extend(el).unsqueeze(-1)
        keep_order = "eid"
        if keep is None and self.training and self.edge_drop > 0:
            keep, keep_order = self._kept_edges(graph), "csc"   # a uniform random subset: any fixed edge order will do
        a = ops.gat_attention(graph, el, er, keep=keep, negative_slope=self.leaky_relu.negative_slope, order="csc",
                              keep_order=keep_order)
        a = self.attn_drop(a)
        rst = ops.u_mul_e_sum(graph, ft, a, order="csc") if transfer is None else halo.u_mul_e_sum(graph, ft, a, transfer=transfer)
        if self._use_symmetric_norm:
            rst = rst * _bcast(degree_norm(graph, "in", 0.5), rst)
        if self.res_fc is not None:  # residual folded into the GEMM epilogue (beta = 1)
            rst = torch.addmm(rst.reshape(rst.shape[0], H * D), h, self.res_fc.weight.t()).view(-1, H, D)
        if self._activation is not None:
            rst = self._activation(rst)
        return rst


class GCN(nn.Module):
    """GCN stack — models.py:569-641."""

    def __init__(self, in_feats, n_classes, n_hidden, n_layers, activation, norm="none", norm_adj="symm",
                 dropout=0.0, input_drop=0, residual=False, use_linear=False):
        super().__init__()
        self.n_layers, self.n_hidden, self.n_classes = n_layers, n_hidden, n_classes
        self.use_linear, self.residual = use_linear, residual
        self.convs = nn.ModuleList()
        if use_linear:
            self.linear = nn.ModuleList()
        self.norms = nn.ModuleList()
        for i in range(n_layers):
            fin = n_hidden if i > 0 else in_feats
            fout = n_hidden if i < n_layers - 1 else n_classes
            last = i == n_layers - 1
            self.convs.append(GraphConv(fin, fout, "both" if norm_adj == "symm" else "right", bias=norm == "none" or last))
            if use_linear:
                self.linear.append(nn.Linear(fin, fout, bias=False))
            if not last and norm == "batch":
                self.norms.append(nn.BatchNorm1d(fout))
        self.input_drop, self.dropout = nn.Dropout(input_drop), nn.Dropout(dropout)
        self.activation = activation

    def forward(self, graph, feat):
        from . import fused
        h = graph.to_internal(feat)  # identity unless the graph was renumbered (bot_amd.reorder_graph)
        if not fused.take_input_dropped():  # bot_amd.train assembles the input with the dropout applied (one pass) and says so
            h = self.input_drop(h)
        h_last = None
        for i in range(self.n_layers):
            conv = self.convs[i](graph, h)
            h = conv + self.linear[i](h) if self.use_linear else conv
            if i < self.n_layers - 1:
                if self.residual and h_last is not None:
                    h = h + h_last
                h_last = h
                if len(self.norms):
                    h = _epilogue(h, self.norms[i], self.activation, self.dropout, self.training)
                else:
                    h = self.dropout(self.activation(h))
        return graph.to_original(h)


class GAT(nn.Module):
    """GAT stack — models.py:644-736.  Hidden layers use `n_heads` heads and concatenate them
    (`flatten(1)`); the last layer has one head of `dim_output` and the heads are averaged.  With
    norm='batch' the only entry of `biases` is the final one (models.py:696-702)."""

    def __init__(self, dim_node, dim_edge, dim_output, n_hidden, n_layers, n_heads, activation, norm="none",
                 dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0, non_interactive_attn=False,
                 use_symmetric_norm=False, linear=False, residual=False):
        super().__init__()
        self.n_node_feats, self.n_hidden, self.n_classes = dim_node, n_hidden, dim_output
        self.n_layers, self.num_heads = n_layers, n_heads
        self.convs, self.norms, self.biases = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(n_layers):
            last = i == n_layers - 1
            fin = n_heads * n_hidden if i > 0 else dim_node
            fout = dim_output if last else n_hidden
            heads = 1 if last else n_heads
            self.convs.append(GATConv(fin, fout, num_heads=heads, attn_drop=attn_drop, edge_drop=edge_drop,
                                      non_interactive_attn=non_interactive_attn,
                                      use_symmetric_norm=use_symmetric_norm, linear=linear))
            if last:
                self.biases.append(ElementWiseLinear(fout, weight=False, bias=True))
            elif norm == "batch":
                self.norms.append(nn.BatchNorm1d(heads * fout))
            elif norm == "none":
                self.biases.append(ElementWiseLinear(heads * fout, weight=False, bias=True))
        self.input_drop, self.dropout = nn.Dropout(input_drop), nn.Dropout(dropout)
        self.activation, self.residual = activation, residual
        self.fuse_layers = True  # one autograd node per hidden layer when the options allow (bot_amd/nn/fused.py)

    def forward(self, graph, feat):
        from . import fused
        h = graph.to_internal(feat)  # identity unless the graph was renumbered (bot_amd.reorder_graph)
        if not fused.take_input_dropped():  # bot_amd.train assembles the input with the dropout applied (one pass) and says so
            h = self.input_drop(h)
        h_last = None
        infer = self.fuse_layers and not self.training and not torch.is_grad_enabled() and (h.is_cuda or fused.FORCE)
        for i in range(self.n_layers):
            last = i == self.n_layers - 1
            if infer:  # evaluate(): one GEMM + one fused sweep per layer, nothing kept for a backward (bot_amd/nn/fused.py, f3)
                epi = self.biases[-1] if last else (self.norms[i] if len(self.norms) else self.biases[i])
                if fused.can_infer(self.convs[i], epi, self.activation, graph, self.residual, last):
                    h = fused.gat_infer_layer(self.convs[i], epi, graph, h, relu=not last, first=i == 0)
                    if last:
                        return graph.to_original(h)
                    continue
            norm = None if last else (self.norms[i] if len(self.norms) else False)
            if (self.fuse_layers and norm is not False and (h.is_cuda or fused.FORCE)
                    and fused.can_fuse(self.convs[i], norm, self.activation, graph, self.training, self.residual)):
                # does the NEXT layer read this one's output only as the halves its epilogue stashes?  then the fp32 copy is not stored
                y_needed = True
                if not last and norm is not None:
                    nxt_last = i + 1 == self.n_layers - 1
                    nxt_norm = None if nxt_last else (self.norms[i + 1] if len(self.norms) else False)
                    y_needed = nxt_norm is False or not fused.halves_only_consumer(
                        self.convs[i + 1], nxt_norm, self.activation, graph, self.training, self.residual, h.shape[0], h.is_cuda)
                h = fused.gat_hidden_layer(self.convs[i], norm, graph, h, self.dropout.p, self.training, y_needed=y_needed)
                if last:
                    h = h.view(h.shape[0], self.convs[i]._num_heads, -1)
                else:
                    continue
            else:
                h = self.convs[i](graph, h)
            if i < self.n_layers - 1:
                if self.residual and h_last is not None:
                    h = h + h_last
                h_last = h
                h = h.flatten(1)
                if len(self.norms):
                    h = _epilogue(h, self.norms[i], self.activation, self.dropout, self.training)
                else:
                    h = self.dropout(self.activation(self.biases[i](h)))
        # (one output head: the mean over it is the head itself, bit for bit - a view, not a reduction launch and its backward)
        return graph.to_original(self.biases[-1](h.view(h.shape[0], -1) if h.shape[1] == 1 else h.mean(1)))
