"""1-D vertex-partitioned full-batch training across the GPUs of one node (one process per GPU,
`torch.distributed`; backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference is single-device (no distributed code at all); this mode is what
BASELINE.json's north star adds for graphs that outgrow one GPU.  Design:

* Rank p owns a contiguous id range [lo_p, hi_p) chosen so every rank holds about the same number
  of in-edges.  It keeps ALL in-edges of its vertices, so destination rows never need a reduction
  across ranks; the local graph is a block in DGL's sense: owned vertices first, then the *halo*
  (remote sources, sorted by global id = grouped by owner), `bot_amd.Graph(num_dst_nodes=n_owned)`.
  Local edges keep their global edge-id order, so per-destination sums run in the same order as on
  one GPU.
* Per layer, forward: ONE all-to-all(v) of the projected source rows that other ranks need
  (`Graph.extend` -> `HaloPlan.extend`): rows are packed by the HIP row-gather kernel and land
  directly behind the owned rows.  Backward: the reverse all-to-all of the halo-row gradients, folded
  into the owned rows peer by peer in rank order with the HIP scatter-add kernel (each peer's rows are
  sorted-unique: one writer per row, deterministic).  xGMI is point-to-point — one all-to-all keeps
  all 7 links busy at once, which a ring collective would not.
* Replicated: parameters, optimizer state.  Per step: one flat all-reduce(sum) of the parameter
  gradients; BatchNorm statistics over ALL nodes (models.py:698,727 normalise over the node axis)
  via all-reduced sums (`SyncBatchNorm1d`); the loss is the global mean (local sum / global count).
"""
from __future__ import annotations

import os

from dataclasses import dataclass

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _C, halo
from . import train as T
from .graph import Graph


# ------------------------------------------------------------------------------------------------ collectives with autograd
class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.clone()
        dist.all_reduce(y, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


def all_reduce_sum(x, group=None):
    return _AllReduceSum.apply(x, group)


class _HaloExchange(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_own, plan):
        n_own = x_own.shape[0]
        x2 = x_own.reshape(n_own, x_own[0].numel() if n_own else int(torch.tensor(x_own.shape[1:]).prod()))
        F = x2.shape[1]
        ext = torch.empty((n_own + plan.n_halo, F), dtype=x2.dtype, device=x2.device)
        ext[:n_own] = x2
        send = _C.gather_rows(x2, plan.send_rows) if plan.n_send else x2.new_empty((0, F))
        halo.a2a(ext[n_own:], send, plan.recv_splits, plan.send_splits, plan.group)
        ctx.plan, ctx.shape = plan, x_own.shape
        return ext.view((n_own + plan.n_halo,) + tuple(x_own.shape[1:]))

    @staticmethod
    def backward(ctx, g_ext):
        plan = ctx.plan
        n_own = ctx.shape[0]
        g2 = g_ext.reshape(g_ext.shape[0], -1)
        g_own = g2[:n_own].clone()
        back = torch.empty((plan.n_send, g2.shape[1]), dtype=g2.dtype, device=g2.device)
        halo.a2a(back, g2[n_own:].contiguous(), plan.send_splits, plan.recv_splits, plan.group)
        off = 0
        for cnt in plan.send_splits:  # peer by peer, rank order: fixed summation order
            if cnt:
                _C.scatter_add_rows(g_own, plan.send_rows[off:off + cnt], back[off:off + cnt])
            off += cnt
        return g_own.view(ctx.shape), None


@dataclass
class HaloPlan:
    send_rows: torch.Tensor      # int32 [n_send]: local ids of owned rows to ship, grouped by destination rank
    send_splits: list            # rows per destination rank
    recv_splits: list            # halo rows per owner rank (sum = n_halo)
    group: object = None
    n_global: int = 0            # vertices of the whole graph (the row count BatchNorm normalises over, models.py:698,727)

    @property
    def n_send(self):
        return int(sum(self.send_splits))

    @property
    def n_halo(self):
        return int(sum(self.recv_splits))

    def extend(self, x_own):
        return _HaloExchange.apply(x_own, self)

    def to(self, device):
        return HaloPlan(self.send_rows.to(device), self.send_splits, self.recv_splits, self.group, self.n_global)


# ------------------------------------------------------------------------------------------------ partitioning (host, integer)
def partition_bounds(in_degrees: torch.Tensor, world: int, balance: str = "auto") -> list:
    """Contiguous id ranges, world+1 boundaries.  balance="edges": ~equal in-edge counts (the SpMM/attention work);
    "nodes": equal vertex counts (the GEMM / BatchNorm work, and identical GEMM shapes on every rank, so one set of tuned
    kernel selections serves all of them); "auto": equal vertex counts when that leaves the in-edge counts within 5 % of
    each other — true for graphs whose ids carry no degree order, like the randomly relabelled benchmark graphs — else edges."""
    n = in_degrees.numel()
    cum = torch.cumsum(in_degrees.to(torch.int64), 0)
    total = int(cum[-1]) if n else 0
    if balance in ("nodes", "auto") and n:
        b = [(n * k + world - 1) // world if k < world else n for k in range(world + 1)]
        b[0] = 0
        edges = [int(cum[b[k + 1] - 1]) - (int(cum[b[k] - 1]) if b[k] > 0 else 0) if b[k + 1] > b[k] else 0 for k in range(world)]
        if balance == "nodes" or max(edges) <= 1.05 * total / world + 1:
            return b
    targets = torch.tensor([total * k // world for k in range(1, world)], dtype=torch.int64)
    cuts = torch.searchsorted(cum, targets, right=False) + 1 if world > 1 else torch.zeros(0, dtype=torch.int64)
    b = [0] + [int(min(max(c, 0), n)) for c in cuts.tolist()] + [n]
    for i in range(1, len(b)):
        b[i] = max(b[i], b[i - 1])
    return b


@dataclass
class Partition:
    graph: Graph                  # local block: owned vertices first, halo sources behind
    rank: int
    world: int
    lo: int
    hi: int
    halo_global: torch.Tensor     # int64 [n_halo] global ids of the halo sources
    n_edges: int
    edge_ids: torch.Tensor = None    # int64 [n_edges] global ids of the local edges, ascending = local edge-id order
                                     # (edge features of the block: efeat[edge_ids]; configs 4/5, src/ogbn-proteins/models.py:244)
    node_ids: torch.Tensor = None    # int64 [n_owned]: ORIGINAL ids of the owned vertices when the partitioner renumbered them
                                     # (None: lo + arange(n_owned))
    feat: torch.Tensor = None
    labels: torch.Tensor = None
    train_idx: torch.Tensor = None   # LOCAL ids of owned training nodes (same for val/test)
    val_idx: torch.Tensor = None
    test_idx: torch.Tensor = None

    @property
    def n_owned(self):
        return self.hi - self.lo


def build_partition(src, dst, num_nodes: int, rank: int, world: int, device=None, group=None, bounds=None,
                    chunk=None) -> Partition:
    """Cut rank's block out of the whole graph (edge list in edge-id order).  Pure integer work with torch ops on the device
    the edge list lives on — on the GPU box every rank holds the (identically seeded) edge list in its OWN HBM, so no host ever
    holds `world` copies — and no communication: ranks derive matching halo / send lists from the same edge list.  Only the
    edges that touch this rank are processed beyond the two owner look-ups: its in-edges (the local block and the halo it
    receives) and the out-edges of its vertices that end elsewhere (what it sends)."""
    dev = src.device
    device = dev if device is None else torch.device(device)
    if bounds is None:
        bounds = partition_bounds(torch.bincount(dst, minlength=num_nodes).cpu(), world)
    bt = torch.tensor(bounds[1:], dtype=torch.int64, device=dev)
    own_src = torch.searchsorted(bt, src, right=True)   # owner rank of every edge's source
    own_dst = torch.searchsorted(bt, dst, right=True)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_own = hi - lo
    # local block: all in-edges of my vertices, in global edge-id order
    m = own_dst == rank
    ls, ld = src[m], dst[m] - lo
    remote = own_src[m] != rank
    halo_global = torch.unique(ls[remote])                # sorted by global id = grouped by owner rank
    recv_splits = torch.bincount(torch.searchsorted(bt, halo_global, right=True), minlength=world).tolist()
    # what I send: my vertices that are sources of edges ending on another rank, per destination rank, sorted by id
    out = (own_src == rank) & ~m
    key = torch.unique(own_dst[out] * num_nodes + src[out])
    send_rows = (key % num_nodes - lo).to(torch.int32)
    send_splits = torch.bincount(torch.div(key, num_nodes, rounding_mode="floor"), minlength=world).tolist()
    local_src = torch.where(remote, n_own + torch.searchsorted(halo_global, ls), ls - lo)
    g = Graph(local_src, ld, n_own + halo_global.numel(), num_dst_nodes=n_own, chunk=chunk)
    g.global_out_degrees = torch.bincount(src[own_src == rank] - lo, minlength=n_own).to(torch.int64)
    g.halo = HaloPlan(send_rows.contiguous(), send_splits, recv_splits, group, n_global=int(num_nodes))
    g = g.to(device)
    g.create_formats_()
    return Partition(g, rank, world, lo, hi, halo_global.to(device), int(ls.numel()), edge_ids=torch.nonzero(m).squeeze(1).to(device))


def partition_dataset(ds, rank: int, world: int, device, group=None, partitioner: str = "contiguous") -> Partition:
    """Partition a `bot_amd.synth.Dataset` and move rank's slice to `device`.

    partitioner="contiguous": ranges of the graph's own ids (for the benchmark graphs these are random: ~ (P-1)/P of every
    rank's sources are remote).  "community": an edge-cut-aware 1-D partition — the vertices are first
    renumbered by `bot_amd.graph.reorder_permutation(…, "community")` (label propagation on the device, communities
    contiguous, hubs first inside a community) and the ranges are cut in THAT order, so most edges of a community stay inside
    one rank; `Partition.node_ids` keeps the original ids of the owned rows.  A graph that was already renumbered by
    `preprocess(reorder=...)` is partitioned in its internal order."""
    from .graph import reorder_permutation
    g = ds.graph
    s, d = g.edges()
    n = g.number_of_nodes()
    perm = g.node_perm                                   # internal -> original (None: identity)
    if partitioner == "community":
        p2, labels = reorder_permutation(g, "community")  # new -> current
        if int(torch.bincount(labels).max()) * 4 <= n:    # a real community structure (same criterion as reorder_graph) ...
            inv2 = torch.empty_like(p2)
            inv2[p2] = torch.arange(n, dtype=p2.dtype, device=p2.device)
            s, d = inv2[s], inv2[d]
            perm = p2 if perm is None else perm[p2]
        # ... otherwise the labels flooded into one community and the order is just degree-descending: ranges of it are badly
        # unbalanced in vertices (hubs first) and cut as many edges as the given numbering, so that one is kept
    elif partitioner != "contiguous":
        raise ValueError(f"unknown partitioner {partitioner!r}")
    p = build_partition(s, d, n, rank, world, device, group)
    if perm is None:
        rows = slice(p.lo, p.hi)
        to_new = None
    else:
        perm = perm.to(ds.feat.device)
        rows = perm[p.lo:p.hi]
        p.node_ids = rows.to(device)
        to_new = torch.empty_like(perm)
        to_new[perm] = torch.arange(n, dtype=perm.dtype, device=perm.device)
    p.feat = ds.feat[rows].to(device)
    p.labels = ds.labels[rows].to(device)
    for name in ("train_idx", "val_idx", "test_idx"):
        idx = getattr(ds, name)
        if to_new is not None:
            idx = to_new[idx.to(to_new.device)]
        idx = idx[(idx >= p.lo) & (idx < p.hi)] - p.lo
        setattr(p, name, torch.sort(idx).values.to(device))
    return p


def halo_statistics(src, dst, num_nodes: int, world: int, bounds=None) -> dict:
    """Halo rows every rank would receive under contiguous ranges of the given numbering (host-side planning aid)."""
    if bounds is None:
        bounds = partition_bounds(torch.bincount(dst, minlength=num_nodes).cpu(), world)
    bt = torch.tensor(bounds[1:], dtype=torch.int64, device=src.device)
    od, os_ = torch.searchsorted(bt, dst, right=True), torch.searchsorted(bt, src, right=True)
    cross = od != os_
    key = torch.unique(od[cross] * num_nodes + src[cross])
    halo = torch.bincount(torch.div(key, num_nodes, rounding_mode="floor"), minlength=world)
    return {"halo_rows_per_rank": halo.tolist(), "cut_edges": int(cross.sum()), "edges": int(src.numel()),
            "owned_rows_per_rank": [bounds[k + 1] - bounds[k] for k in range(world)]}


# ------------------------------------------------------------------------------------------------ model pieces
class SyncBatchNorm1d(nn.BatchNorm1d):
    """BatchNorm1d whose batch statistics cover the nodes of ALL ranks (the reference normalises over the
    whole node set, models.py:698,727).  Same parameters / buffers / state_dict keys as nn.BatchNorm1d.
    Device-agnostic (plain torch ops + differentiable all-reduce), so the gloo tests cover it."""

    group = None
    _bot_sync = True  # bot_amd.ops.bn_relu_dropout: all-reduce the column statistics across ranks

    def forward(self, x):
        if not self.training or not (dist.is_available() and dist.is_initialized()):
            return super().forward(x)
        stats = torch.cat([x.sum(0), (x * x).sum(0), x.new_full((1,), float(x.shape[0]))])
        stats = all_reduce_sum(stats, self.group)
        C = x.shape[1]
        cnt = stats[-1]
        mean = stats[:C] / cnt
        var = stats[C:2 * C] / cnt - mean * mean
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - mom).add_(mean.detach(), alpha=mom)
                self.running_var.mul_(1 - mom).add_(var.detach() * (cnt / (cnt - 1)), alpha=mom)
        y = (x - mean) * torch.rsqrt(var + self.eps)
        return y * self.weight + self.bias if self.affine else y


def wrap_model(model: nn.Module, group=None) -> nn.Module:
    """Swap every nn.BatchNorm1d for SyncBatchNorm1d in place (parameters and buffers are shared)."""
    for name, child in list(model.named_children()):
        if type(child) is nn.BatchNorm1d:
            sbn = SyncBatchNorm1d(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            sbn.load_state_dict(child.state_dict())
            sbn.to(next(iter(child.state_dict().values())).device)
            if child.affine:
                sbn.weight, sbn.bias = child.weight, child.bias
            sbn.group = group
            sbn.train(child.training)
            if isinstance(model, nn.ModuleList):
                model[int(name)] = sbn
            else:
                setattr(model, name, sbn)
        else:
            wrap_model(child, group)
    return model


def all_reduce_grads(model: nn.Module, group=None):
    """One flat all-reduce(sum) of every parameter gradient (5.8 MB at BASELINE config 2)."""
    params = [p for p in model.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, group=group)
    off, views = 0, []
    for p in params:
        n = p.numel()
        views.append(flat[off:off + n].view_as(p))
        off += n
    torch._foreach_copy_([p.grad for p in params], views)       # one multi-tensor launch (a copy per parameter was 27 launches a step)


def seed_rank_streams(base_seed: int, rank: int):
    """Give every rank its own dropout / edge-drop / label-mask stream (call AFTER the replicated model was built from the
    common seed): the fused kernels draw their Philox seeds from torch's CPU generator and count elements from 0 on every
    rank, so identically seeded ranks would drop the same local positions."""
    torch.manual_seed((int(base_seed) * 1000003 + 7919 * (int(rank) + 1)) & 0x7FFFFFFFFFFFFFFF)


def _global_mean(y, w, group):
    """sum(y * w) / global sum(w) as this rank's additive share of the global mean (its backward gives the right scale)."""
    from .ops import sum_all
    cnt = sum_all(w).reshape(1)
    dist.all_reduce(cnt, group=group)
    return sum_all(y * w) / cnt[0]


def forward_backward(model, part: Partition, *, use_labels=True, mask_rate=0.5, n_label_iters=0, loss="logit", n_classes=None,
                     mask=None, group=None):
    """Partitioned counterpart of `bot_amd.train.forward_backward` (run.py:252-284): same step, the loss is the
    mean over the prediction nodes of ALL ranks, parameter gradients are summed over ranks."""
    tr = part.train_idx
    if (T.FUSED_STEP and n_label_iters == 0 and part.feat.dtype == torch.float32 and loss in ("logit", "loge", "savage")
            and os.environ.get("BOT_DIST_FUSED_STEP", "1") != "0"):
        # the single-GPU step's glue (label split, input assembly + input dropout, per-node loss with its gradient: four launches for ~45
        # tensor ops, bot_amd.train) with the prediction-node COUNT summed over the ranks before the loss reads it (round 6: at one rank
        # the partitioned step paid 0.2 ms and ~40 launches for the tensor-op form, profiles/r06_partitioned_1rank_kernel_diff.txt)
        local, pred, _ = T._fused_forward_backward(model, part.graph, part.feat, part.labels, tr, use_labels=use_labels, mask_rate=mask_rate, loss=loss,
                                                   n_classes=n_classes, mask=mask, count_reduce=lambda c: dist.all_reduce(c, group=group))
        all_reduce_grads(model, group)
        total = local.detach().clone()
        dist.all_reduce(total, group=group)
        return total, pred
    if mask is None:
        mask = torch.rand(tr.shape, device=tr.device) < mask_rate
    # Fixed shapes throughout (no `tr[mask]`: boolean indexing makes the host wait for the device twice per step, which is
    # what bounds a rank once its GPU work is a few ms): the label columns of the masked-out training nodes are written as
    # zeros, and the loss runs over ALL owned training nodes with 0/1 weights.  Same sets, same mean as run.py:256-281.
    feat = part.feat
    tr_labels = part.labels[tr, 0]
    if use_labels:
        onehot = torch.zeros([feat.shape[0], n_classes], device=feat.device, dtype=feat.dtype)
        onehot[tr, tr_labels] = mask.to(feat.dtype)
        feat = torch.cat([feat, onehot], dim=-1)
        w = (~mask).to(feat.dtype)
    else:
        w = mask.to(feat.dtype)
    pred = model(part.graph, feat)
    if n_label_iters > 0 and use_labels:
        # label reuse (run.py:274-279): the label columns of every node WITHOUT an input label — the masked-out training nodes,
        # validation and test nodes — are overwritten with the previous prediction's softmax, then the model runs again
        m = mask.unsqueeze(1)
        for _ in range(n_label_iters):
            pred = pred.detach()
            prob = torch.softmax(pred, dim=-1)
            feat[tr, -n_classes:] = torch.where(m, feat[tr, -n_classes:], prob[tr])
            for idx in (part.val_idx, part.test_idx):
                feat[idx, -n_classes:] = prob[idx]
            pred = model(part.graph, feat)
    wn = torch.zeros(pred.shape[0], device=pred.device, dtype=pred.dtype)   # weighted mean over all owned rows: see bot_amd.train
    wn[tr] = w
    y = T.per_node_loss(pred, part.labels.clamp(0, pred.shape[1] - 1), loss)   # placeholder labels outside the set: bot_amd.train
    local = _global_mean(torch.where(wn > 0, y, torch.zeros_like(y)), wn, group)
    local.backward()
    all_reduce_grads(model, group)
    total = local.detach().clone()
    dist.all_reduce(total, group=group)
    return total, pred


def train_step(model, part: Partition, optimizer, **kw):
    model.train()
    optimizer.zero_grad()
    loss, pred = forward_backward(model, part, **kw)
    optimizer.step()
    return loss, pred


@torch.no_grad()
def evaluate(model, part: Partition, *, use_labels=True, n_label_iters=0, loss="logit", n_classes=None, group=None):
    """Partitioned counterpart of `bot_amd.train.evaluate` (run.py:290-322): eval-mode forward with every training label as
    input, optional label reuse; returns (train_acc, val_acc, test_acc, train_loss, val_loss, test_loss, pred of the owned rows),
    accuracies and losses being GLOBAL means (all-reduced sums and counts)."""
    import contextlib
    from .nn import fused
    model.eval()
    feat = part.feat
    n_static = feat.shape[1]
    if use_labels:
        feat = T.add_labels(feat, part.labels, part.train_idx, n_classes)
    with (fused.label_reuse(n_static) if (use_labels and n_label_iters > 0) else contextlib.nullcontext()):
        pred = model(part.graph, feat)
        for _ in range(n_label_iters if use_labels else 0):
            prob = torch.softmax(pred, dim=-1)
            for idx in (part.val_idx, part.test_idx):
                feat[idx, -n_classes:] = prob[idx]
            pred = model(part.graph, feat)
    stats = []
    for idx in (part.train_idx, part.val_idx, part.test_idx):
        y = T.per_node_loss(pred[idx], part.labels[idx], loss)
        hit = (torch.argmax(pred[idx], dim=1) == part.labels[idx, 0]).to(pred.dtype)
        stats += [y.sum(), hit.sum(), y.new_tensor(float(idx.numel()))]
    stats = torch.stack(stats)
    dist.all_reduce(stats, group=group)
    st = stats.tolist()
    losses = tuple(st[3 * i] / max(st[3 * i + 2], 1.0) for i in range(3))
    accs = tuple(st[3 * i + 1] / max(st[3 * i + 2], 1.0) for i in range(3))
    return accs + losses + (pred,)


def step_generic(model_call, part: Partition, node_loss, idx=None, weights=None, model=None, optimizer=None, group=None):
    """One partitioned train step for stacks with their own calling convention (the edge-feature GATs of configs 4 / 5:
    `model(graph)` reading `graph.ndata['feat']` / `graph.edata['feat']`, src/ogbn-proteins/gat.py:123-140).  `model_call()`
    returns the predictions of the owned rows; `node_loss(pred[idx], labels[idx])` a per-node loss [len(idx)]; the loss is the
    global mean over `idx` (default: the owned training nodes) with optional 0/1 `weights`."""
    idx = part.train_idx if idx is None else idx
    if optimizer is not None:
        model.train()
        optimizer.zero_grad()
    pred = model_call()
    y = node_loss(pred[idx], part.labels[idx])
    w = torch.ones_like(y) if weights is None else weights
    local = _global_mean(y, w, group)
    local.backward()
    if model is not None:
        all_reduce_grads(model, group)
    if optimizer is not None:
        optimizer.step()
    total = local.detach().clone()
    dist.all_reduce(total, group=group)
    return total, pred
