"""TEST INFRASTRUCTURE ONLY — restatement of the `dgl 0.5.*` sparse operators the reference calls.

The reference (AiRyunn/BoT) holds no arithmetic of its own for these: every graph operation is a
call into the un-vendored dependency `dgl 0.5.*` (reference README.md:9).  PARITY UNPINNED for
everything in this file except `copy_u_sum` + degrees (pinned by the GraphConv docstring rows,
src/no-sampling/models.py:186-209, see tests/test_oracle_golden.py).  Each function cites the
reference call site whose semantics it restates.

Everything here is written with differentiable torch CPU ops over the COO edge list, so backward
passes come from torch autograd applied to the *forward* definition — deliberately independent of
the hand-derived backward formulas the HIP kernels implement.

Conventions: `src`, `dst` are int64 [E] tensors in edge-id order (edge e goes src[e] -> dst[e]);
node tensors are [N, ...]; edge tensors are [E, ...] in edge-id order.
"""
from __future__ import annotations

import torch


def in_degrees(dst: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """`graph.in_degrees()` — models.py:335,388,478,551.  int64 [N], bit-exact."""
    return torch.bincount(dst, minlength=num_nodes).to(torch.int64)


def out_degrees(src: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """`graph.out_degrees()` — models.py:352,501; ogbn-proteins/gat.py:64.  int64 [N], bit-exact."""
    return torch.bincount(src, minlength=num_nodes).to(torch.int64)


def copy_u_sum(src, dst, num_dst: int, x: torch.Tensor) -> torch.Tensor:
    """`update_all(fn.copy_src('h','m'), fn.sum('m','h'))` — models.py:374,381.

    out[v] = sum over in-edges (u->v), in ascending edge-id order, of x[u].
    """
    out = torch.zeros((num_dst,) + tuple(x.shape[1:]), dtype=x.dtype)
    return out.index_add(0, dst, x[src])


def u_mul_e_sum(src, dst, num_dst: int, x: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """`update_all(fn.u_mul_e('ft','a','m'), fn.sum('m','ft'))` — models.py:547;
    ogbn-proteins/models.py:146; ogbn-products/models.py:147.

    x is [N, H, D], a is [E, H, 1] (broadcast over D).  out[v,h,:] = sum_e a[e,h] * x[src[e],h,:].
    """
    out = torch.zeros((num_dst,) + tuple(x.shape[1:]), dtype=x.dtype)
    return out.index_add(0, dst, x[src] * a)


def copy_e_sum(dst, num_dst: int, w: torch.Tensor) -> torch.Tensor:
    """`update_all(fn.copy_e('feat','feat_copy'), fn.sum('feat_copy','feat'))` — ogbn-proteins/gat.py:58."""
    out = torch.zeros((num_dst,) + tuple(w.shape[1:]), dtype=w.dtype)
    return out.index_add(0, dst, w)


def copy_u(src, x: torch.Tensor) -> torch.Tensor:
    """`apply_edges(fn.copy_u('el','e'))` — models.py:525.  e[eid] = x[src[eid]]."""
    return x[src]


def u_add_v(src, dst, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """`apply_edges(fn.u_add_v('el','er','e'))` — models.py:523.  e[eid] = x[src[eid]] + y[dst[eid]]."""
    return x[src] + y[dst]


def edge_softmax(dst, num_dst: int, e: torch.Tensor, eids: torch.Tensor | None = None) -> torch.Tensor:
    """`dgl.ops.edge_softmax(graph, e)` — models.py:544 — and the `eids=` form — models.py:537.

    For every destination v and every trailing index, softmax over the in-edges of v.  With `eids`
    given, `e` holds values for those edges only (shape [len(eids), ...]); the softmax runs over
    the edge-induced subgraph that keeps all nodes, and the result is returned in the order of `eids`.
    """
    d = dst if eids is None else dst[eids]
    idx = d.view((-1,) + (1,) * (e.dim() - 1)).expand_as(e)
    m = torch.full((num_dst,) + tuple(e.shape[1:]), float("-inf"), dtype=e.dtype)
    m = m.scatter_reduce(0, idx, e.detach(), reduce="amax", include_self=True)
    ex = torch.exp(e - m[d])
    s = torch.zeros((num_dst,) + tuple(e.shape[1:]), dtype=e.dtype).index_add(0, d, ex)
    return ex / s[d]


# ---------------------------------------------------------------------------------------------
# graph transforms used by preprocess() — run.py:138,143.  Integer work: bit-exact.
# ---------------------------------------------------------------------------------------------

def to_bidirected(src, dst, num_nodes: int):
    """`dgl.to_bidirected(graph)` — run.py:138.

    DGL 0.5 documents it as add-reverse-edges followed by `to_simple` (duplicate (u,v) pairs
    collapsed).  The edge order after `to_simple` is the sorted order of the (src, dst) pairs
    [upstream-DGL, recalled; unpinned — the reference never prints edge ids].
    """
    s = torch.cat([src, dst])
    d = torch.cat([dst, src])
    key = torch.unique(s * num_nodes + d)  # sorted ascending == lexicographic (src, dst)
    return key // num_nodes, key % num_nodes


def remove_self_loop(src, dst):
    """`graph.remove_self_loop()` — run.py:143.  Keeps the relative order of the surviving edges."""
    keep = src != dst
    return src[keep], dst[keep]


def add_self_loop(src, dst, num_nodes: int):
    """`graph.add_self_loop()` — run.py:143.  Appends (i,i) for i = 0..N-1; their edge ids are the last N."""
    loops = torch.arange(num_nodes, dtype=src.dtype)
    return torch.cat([src, loops]), torch.cat([dst, loops])


def preprocess_edges(src, dst, num_nodes: int):
    """Edge-list part of `preprocess(graph)` — run.py:133-148."""
    s, d = to_bidirected(src, dst, num_nodes)
    s, d = remove_self_loop(s, d)
    return add_self_loop(s, d, num_nodes)


def build_csc(src, dst, num_nodes: int):
    """In-edge (destination-major) compressed form, stable in edge id.

    Returns (indptr int64 [N+1], indices int64 [E] = source of each in-edge, eid int64 [E])."""
    eid = torch.argsort(dst, stable=True)
    indptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    indptr[1:] = torch.cumsum(torch.bincount(dst, minlength=num_nodes), 0)
    return indptr, src[eid], eid


def build_csr(src, dst, num_nodes: int):
    """Out-edge (source-major) compressed form, stable in edge id: (indptr, indices = dst, eid)."""
    return build_csc(dst, src, num_nodes)
