"""TEST INFRASTRUCTURE ONLY — autograd wrappers around oracle/_build/liboracle.so (the C restatement of
DGL's CPU kernels, oracle/csrc/oracle_ops.c).  Mirrors how DGL's PyTorch backend wires its C++
kernels into autograd [upstream-DGL, recalled: python/dgl/backend/pytorch/sparse.py]: SpMM backward
= SpMM on the reversed graph + SDDMM dot; edge_softmax backward = a*da - a*sum(a*da).

Used as a second oracle (tests/test_oracle_c.py) and as the timed CPU baseline of bench.py
(`cpu_baseline.kind = "port"`).  PARITY UNPINNED like oracle/ref_ops.py.  Never used by bot_amd.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_int64, c_void_p

import torch

from . import ref_ops as R

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build")
_LIB = os.path.join(_DIR, "liboracle.so")
_libs = {}


def lib(dtype=torch.float32):
    """liboracle.so (float32) or liboracle_f64.so (the same loops in double, for ranking two fp32 runs against an exact one)."""
    if dtype not in _libs:
        path = {torch.float32: _LIB, torch.float64: os.path.join(_DIR, "liboracle_f64.so")}[dtype]
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle/csrc`")
        _lib = _libs[dtype] = ctypes.CDLL(path)
        P, I = c_void_p, c_int64
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
        for name, args in {
            "oracle_spmm": [P, P, P, I, P, P, I, I, P],
            "oracle_sddmm_dot": [P, P, P, I, P, P, I, I, P],
            "oracle_u_add_v": [P, P, I, P, P, I, P],
            "oracle_edge_softmax_fwd": [P, P, I, P, P, I, P],
            "oracle_edge_softmax_bwd": [P, P, I, P, P, I, P],
            "oracle_segment_sum": [P, P, I, P, I, P],
        }.items():
            getattr(_lib, name).argtypes = args
            getattr(_lib, name).restype = None
    return _libs[dtype]


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n: int):
    """Thread count of the OpenMP kernels (torch's own pool is set with torch.set_num_threads)."""
    for dt in (torch.float32, torch.float64):
        if dt is torch.float32 or os.path.exists(os.path.join(_DIR, "liboracle_f64.so")):
            lib(dt).oracle_set_num_threads(int(n))


class CGraph:
    """COO + both compressed directions with int64 ids (DGL's default idtype), built once."""

    def __init__(self, src, dst, num_nodes):
        self.src, self.dst, self.num_nodes = src.contiguous(), dst.contiguous(), int(num_nodes)
        self.csc = tuple(t.contiguous() for t in R.build_csc(src, dst, num_nodes))  # rows = dst
        self.csr = tuple(t.contiguous() for t in R.build_csr(src, dst, num_nodes))  # rows = src

    @property
    def num_edges(self):
        return int(self.src.numel())


def _p(t):
    return None if t is None else t.data_ptr()


def _spmm(direction, n, x3, w2):
    ip, idx, eid = direction
    H, D = x3.shape[1], x3.shape[2]
    out = torch.empty((n, H, D), dtype=x3.dtype)
    lib(x3.dtype).oracle_spmm(_p(ip), _p(idx), _p(eid), n, _p(x3), _p(w2), H, D, _p(out))
    return out


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, x, a):
        x3 = x.reshape(x.shape[0], 1 if a is None else a.shape[1], -1).contiguous()
        a2 = None if a is None else a.reshape(a.shape[0], -1).contiguous()
        ctx.g, ctx.xs, ctx.as_ = g, x.shape, None if a is None else a.shape
        ctx.save_for_backward(x3, a2)
        return _spmm(g.csc, g.num_nodes, x3, a2).view((g.num_nodes,) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, dout):
        g = ctx.g
        x3, a2 = ctx.saved_tensors
        d3 = dout.reshape(x3.shape).contiguous()
        dx = _spmm(g.csr, g.num_nodes, d3, a2).view(ctx.xs)
        da = None
        if a2 is not None:
            ip, idx, eid = g.csc
            da = torch.empty_like(a2)
            lib(x3.dtype).oracle_sddmm_dot(_p(ip), _p(idx), _p(eid), g.num_nodes, _p(x3), _p(d3), x3.shape[1], x3.shape[2], _p(da))
            da = da.view(ctx.as_)
        return None, dx, da


def copy_u_sum(g, x):
    return _SpMM.apply(g, x, None)


def u_mul_e_sum(g, x, a):
    return _SpMM.apply(g, x, a)


class _UAddV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, x, y):
        ctx.g, ctx.xs, ctx.has_y = g, x.shape, y is not None
        x2 = x.reshape(x.shape[0], -1).contiguous()
        y2 = None if y is None else y.reshape(y.shape[0], -1).contiguous()
        out = torch.empty((g.num_edges, x2.shape[1]), dtype=x2.dtype)
        lib(x2.dtype).oracle_u_add_v(_p(g.src), _p(g.dst), g.num_edges, _p(x2), _p(y2), x2.shape[1], _p(out))
        return out.view((g.num_edges,) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, de):
        g = ctx.g
        de2 = de.reshape(de.shape[0], -1).contiguous()
        W = de2.shape[1]
        outs = []
        for direction, want in ((g.csr, True), (g.csc, ctx.has_y)):
            if not want:
                outs.append(None)
                continue
            ip, _, eid = direction
            o = torch.empty((g.num_nodes, W), dtype=de2.dtype)
            lib(de2.dtype).oracle_segment_sum(_p(ip), _p(eid), g.num_nodes, _p(de2), W, _p(o))
            outs.append(o.view(ctx.xs))
        return None, outs[0], outs[1]


def u_add_v(g, x, y):
    return _UAddV.apply(g, x, y)


def copy_u(g, x):
    return _UAddV.apply(g, x, None)


class _EdgeSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, e, keep):
        ip, _, eid = g.csc
        e2 = e.reshape(e.shape[0], -1).contiguous()
        a = torch.empty_like(e2)
        lib(e2.dtype).oracle_edge_softmax_fwd(_p(ip), _p(eid), g.num_nodes, _p(e2), _p(keep), e2.shape[1], _p(a))
        ctx.g, ctx.shape = g, e.shape
        ctx.save_for_backward(a)
        return a.view(e.shape)

    @staticmethod
    def backward(ctx, da):
        g = ctx.g
        (a,) = ctx.saved_tensors
        ip, _, eid = g.csc
        da2 = da.reshape(a.shape).contiguous()
        de = torch.empty_like(a)
        lib(a.dtype).oracle_edge_softmax_bwd(_p(ip), _p(eid), g.num_nodes, _p(a), _p(da2), a.shape[1], _p(de))
        return None, de.view(ctx.shape), None


def edge_softmax(g, e, keep=None):
    return _EdgeSoftmax.apply(g, e, keep)
