"""TEST INFRASTRUCTURE ONLY — a minimal stand-in for the `dgl` / `ogb` import surface of the reference.

Purpose: let `oracle/gen_golden.py` import the reference's *own* `src/no-sampling/models.py`,
`src/no-sampling/run.py` and `src/ogbn-proteins/models.py` from /root/reference, unmodified, in the
authoring container (where `dgl 0.5.*` cannot be installed) and execute their module logic to
produce the golden vectors committed under tests/golden/.  The graph arithmetic behind the surface
is `oracle/ref_ops.py` (PARITY UNPINNED — see that file); the module logic that runs on top of it
is the reference's, verbatim.

It never travels into the product and is never used on the GPU box (nothing there imports the
reference).  The closed surface implemented is exactly what the reference touches
(models.py:3-13, run.py:11-30, ogbn-proteins/models.py:4-16):

  graph:   local_scope, in_degrees, out_degrees, number_of_edges, number_of_nodes,
           number_of_dst_nodes, is_block, device, srcdata/dstdata/ndata/edata, update_all,
           apply_edges, remove_self_loop, add_self_loop, create_formats_, to
  dgl:     to_bidirected, graph, function.{copy_src,copy_u,copy_e,u_add_v,u_mul_e,sum},
           ops.edge_softmax, utils.expand_as_pair, _ffi.base.DGLError, base.ALL, random.seed,
           nn.pytorch (+ .utils.Identity), data.<8 dataset class names>
  ogb:     nodeproppred.{DglNodePropPredDataset, Evaluator}
"""
from __future__ import annotations

import contextlib
import sys
import types

import torch

from . import ref_ops


class _Msg:
    def __init__(self, kind, a, b, out):
        self.kind, self.a, self.b, self.out = kind, a, b, out


class _Reduce:
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


class StandinGraph:
    """Homogeneous, non-block graph holding a COO edge list in edge-id order."""

    is_block = False

    def __init__(self, src, dst, num_nodes):
        self._src = src.to(torch.int64)
        self._dst = dst.to(torch.int64)
        self._n = int(num_nodes)
        self.ndata = {}
        self.edata = {}

    # --- dict views: one frame for src and dst on a homogeneous graph
    @property
    def srcdata(self):
        return self.ndata

    @property
    def dstdata(self):
        return self.ndata

    @property
    def device(self):
        return self._src.device

    def to(self, device):
        return self

    def number_of_nodes(self):
        return self._n

    def number_of_dst_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.numel())

    def in_degrees(self):
        return ref_ops.in_degrees(self._dst, self._n)

    def out_degrees(self):
        return ref_ops.out_degrees(self._src, self._n)

    def edges(self):
        return self._src, self._dst

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata, self.edata = nd, ed

    def create_formats_(self):
        return None

    def remove_self_loop(self):
        s, d = ref_ops.remove_self_loop(self._src, self._dst)
        g = StandinGraph(s, d, self._n)
        g.ndata = dict(self.ndata)
        return g

    def add_self_loop(self):
        s, d = ref_ops.add_self_loop(self._src, self._dst, self._n)
        g = StandinGraph(s, d, self._n)
        g.ndata = dict(self.ndata)
        return g

    def apply_edges(self, msg):
        if msg.kind == "copy_u":
            self.edata[msg.out] = ref_ops.copy_u(self._src, self.ndata[msg.a])
        elif msg.kind == "u_add_v":
            self.edata[msg.out] = ref_ops.u_add_v(self._src, self._dst, self.ndata[msg.a], self.ndata[msg.b])
        else:
            raise NotImplementedError(msg.kind)

    def update_all(self, msg, red):
        assert red.msg == msg.out
        if msg.kind == "copy_u":
            out = ref_ops.copy_u_sum(self._src, self._dst, self._n, self.ndata[msg.a])
        elif msg.kind == "u_mul_e":
            out = ref_ops.u_mul_e_sum(self._src, self._dst, self._n, self.ndata[msg.a], self.edata[msg.b])
        elif msg.kind == "copy_e":
            out = ref_ops.copy_e_sum(self._dst, self._n, self.edata[msg.a])
        else:
            raise NotImplementedError(msg.kind)
        self.ndata[red.out] = out


def _edge_softmax(graph, logits, eids=None, norm_by="dst"):
    assert norm_by == "dst"
    if eids is not None and not torch.is_tensor(eids):
        eids = None  # dgl.base.ALL
    return ref_ops.edge_softmax(graph._dst, graph._n, logits, eids)


def _expand_as_pair(x, g=None):
    if isinstance(x, tuple):
        return x
    return x, x


def _to_bidirected(g, copy_ndata=False):
    s, d = ref_ops.to_bidirected(g._src, g._dst, g._n)
    return StandinGraph(s, d, g._n)


def install():
    """Register the stand-in modules in sys.modules (idempotent). Refuses to shadow a real dgl."""
    if "dgl" in sys.modules and not getattr(sys.modules["dgl"], "__bot_standin__", False):
        raise RuntimeError("a real `dgl` is already imported; the stand-in must not shadow it")

    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    dgl = mod("dgl")
    dgl.__bot_standin__ = True
    dgl.__path__ = []
    dgl.to_bidirected = _to_bidirected
    dgl.graph = lambda edges, num_nodes=None: StandinGraph(edges[0], edges[1], num_nodes)
    dgl.DGLGraph = StandinGraph

    fn = mod("dgl.function")
    fn.copy_src = lambda src, out: _Msg("copy_u", src, None, out)
    fn.copy_u = lambda u, out: _Msg("copy_u", u, None, out)
    fn.copy_e = lambda e, out: _Msg("copy_e", e, None, out)
    fn.u_add_v = lambda u, v, out: _Msg("u_add_v", u, v, out)
    fn.u_mul_e = lambda u, e, out: _Msg("u_mul_e", u, e, out)
    fn.sum = lambda msg, out: _Reduce(msg, out)
    dgl.function = fn

    ops = mod("dgl.ops")
    ops.edge_softmax = _edge_softmax
    dgl.ops = ops

    utils = mod("dgl.utils")
    utils.expand_as_pair = _expand_as_pair
    dgl.utils = utils

    ffi = mod("dgl._ffi")
    ffi.__path__ = []
    ffi_base = mod("dgl._ffi.base")

    class DGLError(Exception):
        pass

    ffi_base.DGLError = DGLError
    ffi.base = ffi_base
    dgl._ffi = ffi

    base = mod("dgl.base")
    base.ALL = "__ALL__"
    base.DGLError = DGLError
    dgl.base = base

    rnd = mod("dgl.random")
    rnd.seed = lambda s: None
    dgl.random = rnd

    nn_ = mod("dgl.nn")
    nn_.__path__ = []
    nnp = mod("dgl.nn.pytorch")
    nnp.__path__ = []
    nnu = mod("dgl.nn.pytorch.utils")
    nnu.Identity = torch.nn.Identity
    nnp.utils = nnu
    nn_.pytorch = nnp
    dgl.nn = nn_

    data = mod("dgl.data")
    for name in (
        "AmazonCoBuyComputerDataset AmazonCoBuyPhotoDataset CiteseerGraphDataset CoauthorCSDataset "
        "CoraFullDataset CoraGraphDataset PubmedGraphDataset RedditDataset"
    ).split():
        setattr(data, name, type(name, (), {}))
    dgl.data = data

    dl = mod("dgl.dataloading")
    dgl.dataloading = dl

    ogb = mod("ogb")
    ogb.__path__ = []
    npp = mod("ogb.nodeproppred")
    npp.DglNodePropPredDataset = type("DglNodePropPredDataset", (), {})
    npp.Evaluator = type("Evaluator", (), {})
    ogb.nodeproppred = npp
    return dgl
