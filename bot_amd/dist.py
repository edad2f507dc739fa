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

from dataclasses import dataclass

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _C
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
        dist.all_to_all_single(ext[n_own:], send, plan.recv_splits, plan.send_splits, group=plan.group)
        ctx.plan, ctx.shape = plan, x_own.shape
        return ext.view((n_own + plan.n_halo,) + tuple(x_own.shape[1:]))

    @staticmethod
    def backward(ctx, g_ext):
        plan = ctx.plan
        n_own = ctx.shape[0]
        g2 = g_ext.reshape(g_ext.shape[0], -1)
        g_own = g2[:n_own].clone()
        back = torch.empty((plan.n_send, g2.shape[1]), dtype=g2.dtype, device=g2.device)
        dist.all_to_all_single(back, g2[n_own:].contiguous(), plan.send_splits, plan.recv_splits, group=plan.group)
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

    @property
    def n_send(self):
        return int(sum(self.send_splits))

    @property
    def n_halo(self):
        return int(sum(self.recv_splits))

    def extend(self, x_own):
        return _HaloExchange.apply(x_own, self)

    def to(self, device):
        return HaloPlan(self.send_rows.to(device), self.send_splits, self.recv_splits, self.group)


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
    feat: torch.Tensor = None
    labels: torch.Tensor = None
    train_idx: torch.Tensor = None   # LOCAL ids of owned training nodes (same for val/test)
    val_idx: torch.Tensor = None
    test_idx: torch.Tensor = None

    @property
    def n_owned(self):
        return self.hi - self.lo


def build_partition(src, dst, num_nodes: int, rank: int, world: int, device="cpu", group=None, bounds=None,
                    chunk=None) -> Partition:
    """Cut rank's block out of the whole graph (edge list in edge-id order, on the host).  Pure integer
    work, no communication: every rank derives the same halo lists from the same edge list."""
    src, dst = src.cpu(), dst.cpu()
    if bounds is None:
        bounds = partition_bounds(torch.bincount(dst, minlength=num_nodes), world)
    bt = torch.tensor(bounds[1:], dtype=torch.int64)
    own_src = torch.searchsorted(bt, src, right=True)   # owner rank of every edge's source
    own_dst = torch.searchsorted(bt, dst, right=True)
    lo, hi = bounds[rank], bounds[rank + 1]
    n_own = hi - lo
    # all cross-partition (destination owner, source vertex) pairs, deduplicated and sorted
    cross = own_src != own_dst
    key = torch.unique(own_dst[cross] * num_nodes + src[cross])
    k_dst, k_src = torch.div(key, num_nodes, rounding_mode="floor"), key % num_nodes
    k_own = torch.searchsorted(bt, k_src, right=True)
    mine = k_dst == rank                                  # what I receive: sorted by global id = grouped by owner
    halo_global = k_src[mine]
    recv_splits = torch.bincount(k_own[mine], minlength=world).tolist()
    out = k_own == rank                                   # what I send: sorted by (destination rank, global id)
    send_rows = (k_src[out] - lo).to(torch.int32)
    send_splits = torch.bincount(k_dst[out], minlength=world).tolist()
    # local block
    m = own_dst == rank
    ls, ld = src[m], dst[m] - lo
    local_src = torch.where((ls >= lo) & (ls < hi), ls - lo, n_own + torch.searchsorted(halo_global, ls))
    g = Graph(local_src, ld, n_own + halo_global.numel(), num_dst_nodes=n_own, chunk=chunk)
    g.global_out_degrees = torch.bincount(src, minlength=num_nodes)[lo:hi].to(torch.int64)
    g.halo = HaloPlan(send_rows.contiguous(), send_splits, recv_splits, group)
    g = g.to(device)
    g.create_formats_()
    return Partition(g, rank, world, lo, hi, halo_global, int(ls.numel()), edge_ids=torch.nonzero(m).squeeze(1))


def partition_dataset(ds, rank: int, world: int, device, group=None) -> Partition:
    """Partition a `bot_amd.synth.Dataset` (whole graph on the host) and move rank's slice to `device`."""
    s, d = ds.graph.edges()
    n = ds.graph.number_of_nodes()
    p = build_partition(s, d, n, rank, world, device, group)
    p.feat = ds.feat[p.lo:p.hi].to(device)
    p.labels = ds.labels[p.lo:p.hi].to(device)
    for name in ("train_idx", "val_idx", "test_idx"):
        idx = getattr(ds, name).cpu()
        idx = idx[(idx >= p.lo) & (idx < p.hi)] - p.lo
        setattr(p, name, idx.to(device))
    return p


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
                mom = self.momentum if self.momentum is not None else 0.1
                self.running_mean.mul_(1 - mom).add_(mean.detach(), alpha=mom)
                self.running_var.mul_(1 - mom).add_(var.detach() * (cnt / (cnt - 1)), alpha=mom)
                self.num_batches_tracked += 1
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
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


def forward_backward(model, part: Partition, *, use_labels=True, mask_rate=0.5, loss="logit", n_classes=None, mask=None,
                     group=None):
    """Partitioned counterpart of `bot_amd.train.forward_backward` (run.py:252-284): same step, the loss is the
    mean over the prediction nodes of ALL ranks, parameter gradients are summed over ranks."""
    tr = part.train_idx
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
    y = T.per_node_loss(pred[tr], part.labels[tr], loss)
    cnt = w.sum().reshape(1)
    dist.all_reduce(cnt, group=group)
    local = (y * w).sum() / cnt[0]
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
