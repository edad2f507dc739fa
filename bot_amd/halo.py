"""Partitioned mode (bot_amd.dist): neighbour sums whose halo exchange is OVERLAPPED with the work that does not need it.

A rank's block orders its source rows [owned | halo] and `Graph.halo_split` holds its in-edges split by source class.  The
aggregations here start the all-to-all of the halo rows asynchronously (`start`), sweep the in-edges whose source is owned
meanwhile, and add the halo-source edges once the rows have landed; the backward sweeps the halo rows' out-edges first, sends
those gradients home asynchronously, sweeps the owned rows and folds the returned rows in afterwards.  `torch.distributed`
runs the collective on its own stream and `work.wait()` orders the consumer behind it: no host synchronisation, capturable.
Per destination the sum runs over the owned-source edges first and the halo-source edges second — not the single-GPU order, by
design: values agree to rounding, not bitwise (tests/test_dist_gloo.py compares against one process at 1e-4 / 1e-5).

Users: the merged-GEMM GAT layers (bot_amd.nn.fused._GATHidden, which also fuses the BatchNorm epilogue around it), and through
`copy_u_sum` / `u_mul_e_sum` below GraphConv (models.py:374,381), the modular GATConv (models.py:547) and the edge-feature
GATConvs (ogbn-proteins/models.py:146, ogbn-products/models.py:145).  BOT_HALO_OVERLAP=0: the one-exchange form
(`Graph.extend` in front of the plain operators)."""
from __future__ import annotations

import os

import torch

from . import _C

OVERLAP = os.environ.get("BOT_HALO_OVERLAP", "1") != "0"
CALLS = 0           # aggregations that took the overlapped form here (tests assert the path was taken)


def enabled(graph) -> bool:
    return OVERLAP and graph.halo is not None


# Every halo all-to-all of the product goes through `a2a`, which counts the bytes this rank sends and receives (host arithmetic on the
# split sizes: no device read).  bench.py reports the counter's per-step delta as config.partition.exchange_bytes_per_rank_per_step;
# tests/test_dist_gloo.py holds it to the bytes the collective actually saw and to the partition plan's row counts x the layers' widths.
BYTES = {"sent": 0, "received": 0, "calls": 0}


def a2a(out, inp, out_splits, in_splits, group, async_op=False):
    """dist.all_to_all_single(out, inp, out_splits, in_splits) on rows of equal width, counted."""
    import torch.distributed as dist
    BYTES["sent"] += inp.numel() * inp.element_size()
    BYTES["received"] += out.numel() * out.element_size()
    BYTES["calls"] += 1
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def ship_rows(plan, own2d, async_op=False):
    """The rows of own2d [n_own, W] (row stride allowed) that other ranks need -> (halo [n_halo, W], work handle or None, send buffer:
    keep it referenced until the work is waited on)."""
    import torch.distributed as dist
    W = own2d.shape[1]
    send = _C.gather_rows(own2d, plan.send_rows) if plan.n_send else own2d.new_empty((0, W))
    halo = torch.empty((plan.n_halo, W), dtype=own2d.dtype, device=own2d.device)
    work = a2a(halo, send, plan.recv_splits, plan.send_splits, plan.group, async_op=async_op)
    return halo, (work if async_op else None), send


def return_rows(plan, dhalo, async_op=False):
    """Reverse direction: gradients of the halo rows [n_halo, W] go back to their owners -> (back [n_send, W], work or None)."""
    import torch.distributed as dist
    back = torch.empty((plan.n_send, dhalo.shape[1]), dtype=dhalo.dtype, device=dhalo.device)
    work = a2a(back, dhalo, plan.send_splits, plan.recv_splits, plan.group, async_op=async_op)
    return back, (work if async_op else None)


def fold_back(plan, own2d, back):
    """own2d[send_rows] += back, peer by peer in rank order (each peer's rows are sorted-unique: one writer per row, deterministic)."""
    off = 0
    for cnt in plan.send_splits:
        if cnt:
            _C.scatter_add_rows(own2d, plan.send_rows[off:off + cnt], back[off:off + cnt])
        off += cnt
    return own2d


class Transfer:
    """Halo rows in flight: `wait()` -> [n_halo, W] once the consumer's stream may read them."""
    __slots__ = ("halo", "work", "keep")

    def __init__(self, halo, work, keep):
        self.halo, self.work, self.keep = halo, work, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = self.keep = None
        return self.halo


def _flat2(t):
    w = 1
    for s in t.shape[1:]:
        w *= int(s)
    return t.reshape(t.shape[0], w)


def _pad4(x2):
    F = x2.shape[1]
    return x2 if F % 4 == 0 or F < 5 else torch.nn.functional.pad(x2, (0, 4 - F % 4))


def start(graph, x_own) -> Transfer:
    """Begin shipping the rows of `x_own` [n_own, ...] that other ranks need.  Call it as early as the rows exist (before the
    attention scores are formed) and hand the result to `u_mul_e_sum` / `copy_u_sum`, which take `x_own` itself as their
    differentiable input.  2-D inputs travel padded to a multiple of 4 columns, the width the sweeps use (ops._pad4)."""
    x2 = _flat2(x_own.detach())
    if x_own.dim() == 2:
        x2 = _pad4(x2)
    return Transfer(*ship_rows(graph.halo, x2, async_op=True))


def _part_weights(sub, pos, a2, n_src, H, D):
    """Weights of one part of the split CSC: through the position map `pos` (the row kernels take it), or — where the part is dense
    enough for the L2-blocked sweep, which reads its weights in the part's own order — gathered into an array of their own
    (E*H floats against the E*H*D the sweep gathers)."""
    if a2 is None:
        return None, None
    from . import blocked
    if blocked.plan_for(sub, n_src, H, D) is not None:
        return _C.gather_rows(a2, pos), None
    return a2, pos


class _HaloSum(torch.autograd.Function):
    """out[v] = sum over the in-edges (u -> v) of a_e * x[u] (a None: plain sum) (+ addend[v]) on a partition block, x given for
    the OWNED rows only.  a: [E, H] in CSC position order."""

    @staticmethod
    def forward(ctx, g, x, a, addend, transfer):
        global CALLS
        CALLS += 1
        plan, sp = g.halo, g.halo_split
        n_own = x.shape[0]
        two_d = x.dim() == 2
        x3 = _pad4(x).unsqueeze(1) if two_d else (x if x.dim() == 3 else _flat2(x).unsqueeze(1))
        H, D = x3.shape[1], x3.shape[2]
        if transfer is None:
            transfer = start(g, x)
        a2 = None if a is None else _flat2(a).contiguous()
        ad3 = None
        if addend is not None:
            ad3 = _pad4(addend).unsqueeze(1) if two_d else (addend if addend.dim() == 3 else _flat2(addend).unsqueeze(1))
        w, wp = _part_weights(sp["csc_own"], sp["csc_own_pos"], a2, n_own, H, D)
        out = _C.spmm(sp["csc_own"], x3, w, wp, addend=ad3)                             # owned-source edges
        halo3 = transfer.wait().view(plan.n_halo, H, D)
        if sp["csc_halo"].nnz:                                                          # halo-source edges on top
            w, wp = _part_weights(sp["csc_halo"], sp["csc_halo_pos"], a2, plan.n_halo, H, D)
            out = _C.spmm(sp["csc_halo"], halo3, w, wp, addend=out)
        ctx.g, ctx.xshape, ctx.two_d, ctx.ashape = g, x.shape, two_d, (None if a is None else a.shape)
        ctx.has_addend = addend is not None
        ctx.save_for_backward(x3, halo3, a2)
        if two_d:
            F = x.shape[1]
            return out.view(n_own, -1)[:, :F].contiguous() if D != F else out.view(n_own, F)
        return out.view((n_own,) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        plan, sp = g.halo, g.halo_split
        x3, halo3, a2 = ctx.saved_tensors
        n_own, H, D = x3.shape
        dout = dout.contiguous()
        d3 = _pad4(dout).unsqueeze(1) if ctx.two_d else (dout if dout.dim() == 3 else _flat2(dout).unsqueeze(1))
        d3 = d3.contiguous()
        da = None
        fused = a2 is not None and ctx.needs_input_grad[2] and D <= _C.spmm_dot_max_d(d3)
        if fused:
            da = torch.empty((g.csc.nnz, H), dtype=d3.dtype, device=d3.device)       # both parts fill their own positions
        # halo rows first: their gradients travel home while the owned rows are swept
        if plan.n_halo == 0:
            dhalo = d3.new_empty((0, H, D))
        elif fused:
            dhalo, _ = _C.spmm_dot(sp["csr_halo"], d3, a2, sp["csr_halo_c2c"], halo3, dot=da)
        else:
            dhalo = _C.spmm(sp["csr_halo"], d3, a2, None if a2 is None else sp["csr_halo_c2c"])
        back, work = return_rows(plan, dhalo.view(plan.n_halo, H * D), async_op=True)
        if fused:
            dx, _ = _C.spmm_dot(sp["csr_own"], d3, a2, sp["csr_own_c2c"], x3, dot=da)
        else:
            dx = _C.spmm(sp["csr_own"], d3, a2, None if a2 is None else sp["csr_own_c2c"])
            if a2 is not None and ctx.needs_input_grad[2]:                             # D beyond the fused launch: two sweeps
                da = _C.sddmm_dot(g.csc, torch.cat([x3, halo3]), d3, None)
        work.wait()
        fold_back(plan, dx.view(n_own, H * D), back)
        if ctx.two_d:
            F = ctx.xshape[1]
            dx = dx.view(n_own, -1)[:, :F].contiguous() if D != F else dx.view(n_own, F)
        else:
            dx = dx.view(ctx.xshape)
        return None, dx, (None if da is None else da.view(ctx.ashape)), (dout if ctx.has_addend else None), None


def copy_u_sum(graph, x_own, transfer=None):
    """`ops.copy_u_sum(graph, graph.extend(x_own))` with the exchange overlapped (models.py:374,381)."""
    return _HaloSum.apply(graph, x_own, None, None, transfer)


def u_mul_e_sum(graph, x_own, a, addend=None, transfer=None):
    """`ops.u_mul_e_sum(graph, graph.extend(x_own), a, order="csc", addend=addend)` with the exchange overlapped (models.py:547;
    ogbn-proteins/models.py:146,159-160).  a: [E,H,1] or [E,H] in CSC position order."""
    return _HaloSum.apply(graph, x_own, a, addend, transfer)
