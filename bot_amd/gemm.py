"""The layer's dense projections as fp32 GEMMs on the fp16 matrix cores (include/bot_gnn.h: halves_scale / halves_split /
gemm_halves; kernels in csrc/halves.hip, hipBLASLt call in csrc/gemm.cpp).

An fp32 matrix is held as two fp16 halves per entry, x = (h1 + h2) / s (s a power of two found on the device), laid out as
[h1 | h1 | 2^11 h2] for left operands and [h1 | h2 | 2^-11 h1] for right operands, so that ONE fp16 GEMM over the three-fold
reduction axis forms a1 b1 + a1 b2 + a2 b1 with fp32 accumulation (the 2^11 keeps the second half of a left operand in fp16's
normal range for rows down to 2^-28 of the matrix maximum, csrc/halves.hip "Dynamic range").  Against an fp64 product the result is as close as hipBLASLt's fp32
GEMM (1e-6 of the largest entry at the config-2 shapes) and ~3x faster: the fp32 MFMA rate, not HBM, bounds these GEMMs.

`BOT_GEMM=f32` switches every projection back to the stock fp32 GEMM.
"""
from __future__ import annotations

import os
import weakref

import torch

from . import _C

MODE = os.environ.get("BOT_GEMM", "halves")
NT_KERNEL = os.environ.get("BOT_GEMM_NT", "halves3")   # NT products (forward, input gradient): "halves3" = csrc/halves3.hip (each operand
                                                       # half staged once, three MFMAs per fragment pair), "lib" = hipBLASLt over the 3x-concatenated axis
TN_KERNEL = os.environ.get("BOT_GEMM_TN", "halves3")   # weight gradients (reduction over the node rows): "halves3" = the hand-written split-K
                                                       # kernel of csrc/halves3.hip, "lib" = batched hipBLASLt products over row chunks + combine
# Left operands WITHOUT their duplicate h1 piece ([h1 | 2^11 h2], "order 2"): only the library's concatenated-axis NT GEMM reads the
# duplicate, so with both hand-written kernels in place wide operands are written with two pieces (a third less to write and to hold:
# 0.5 GB per split of the config-2 gradient buffer).  Narrow operands (piece < NODUP_MIN_PIECE) keep three.  (Round 5 tried 64 - with
# fragment-major right operands no product is left that needs the duplicate: S-products 428-432 -> 434-440 ms, S-arxiv equal; no gain, kept
# at 256: profiles/r05_nodup_ab.txt.  BOT_NODUP_MIN_PIECE overrides.)
LEFT_NODUP = NT_KERNEL == "halves3" and TN_KERNEL == "halves3" and os.environ.get("BOT_HALVES_DUP", "0") != "1"
NODUP_MIN_PIECE = int(os.environ.get("BOT_NODUP_MIN_PIECE", "256"))
TN_MIN_OUT = 512 * 1024                                # smaller results (the 40-class output layer) go to the kernel's grouped form with more row splits (TN_NARROW), else the library
NT_MIN_COLS = 192                                      # narrower outputs (the 40-class output layer) leave most of a 256-column tile empty: library
LINEAR_BLOCKS = os.environ.get("BOT_LINEAR_BLOCKS", "1") != "0"   # merged projections hand their column blocks' gradients over without a `cat`
FORCE = False              # tests set this to run the halves path over the emulated (CPU) backend at any row count
MIN_ROWS = 8192            # below this many rows the fp32 GEMM is launch-bound anyway
PIECE_ALIGN = 64
SHIFT = 2048.0             # csrc/common.h kHalvesShift: a left operand's third piece is 2^11 h2
CHUNK_ROWS = 8192          # row chunk of the weight-gradient reduction (one batch entry each)


def enabled(x) -> bool:
    return MODE == "halves" and x.dim() == 2 and x.dtype == torch.float32 and (FORCE or (x.is_cuda and x.shape[0] >= MIN_ROWS))


class Halves:
    """fp16 halves of an fp32 matrix [n, F]: `buf` [n, 3 * piece] fp16 (order 2: [n, 2 * piece]), `scale` [2] = (s, 1/s) on the device.
    order 0: left [h1 | h1 | 2^11 h2]; 1: right [h1 | h2 | 2^-11 h1]; 2: left without the duplicate, [h1 | 2^11 h2]; 3: right, FRAGMENT-MAJOR
    (bot_halves_split_frag_f16: `buf` is [2 ceil(n / 16) piece / 32, 512], read by the hand-written NT kernel only)."""
    __slots__ = ("buf", "scale", "n", "F", "piece", "order", "bn_link")

    def __init__(self, buf, scale, n, F, piece, order):
        self.buf, self.scale, self.n, self.F, self.piece, self.order = buf, scale, n, F, piece, order
        self.bn_link = None      # BnLink: the halves were written by a fused BatchNorm epilogue whose backward wants its reduce pass as a by-product

    @property
    def left(self):
        return self.order != 1

    @property
    def h2_off(self):
        """column of the second half (2^11 h2) of a left operand"""
        return self.piece if self.order == 2 else 2 * self.piece


# The reduce pass of a fused BatchNorm / ReLU / dropout epilogue's backward (1 GB read per hidden layer at config 2) as a by-product of the
# product that WRITES that backward's incoming gradient - the consumer layer's `d h = d out . W^T` (include/bot_gnn.h "v18").  The epilogue
# hangs a BnLink on the halves it writes; the consumer keeps it with its context, passes it to `mm_nt` in its backward, and the epilogue's
# backward claims the partials if the gradient it receives IS that product's output (same storage, same version, same shape: a gradient
# autograd accumulated from several consumers is another tensor, or a bumped version, and takes the pass).
BN_BYPRODUCT = os.environ.get("BOT_BN_BWD_BYPRODUCT", "1") != "0"
BN_BYPRODUCT_CALLS = 0


class BnLink:
    __slots__ = ("x", "mean", "invstd", "w", "b", "p", "seed", "relu", "stats", "key", "ref")

    def __init__(self, x, mean, invstd, w, b, p, seed, relu=True):
        self.x, self.mean, self.invstd, self.w, self.b, self.p, self.seed, self.relu = x, mean, invstd, w, b, p, seed, bool(relu)
        self.stats = self.key = self.ref = None

    def stats_for(self, m, n, k):
        """A by-product request for the product [m, n] of piece width k, None when it cannot carry one."""
        if not BN_BYPRODUCT or self.x is None:
            return None
        st = _C.BnBwdStats(self.x, self.mean, self.invstd, self.w, self.b, self.relu, self.p, self.seed)
        return st if st.fits(m, n, k) else None

    def deliver(self, st, dh):
        # (`ref` keeps the product's output alive until the claim: its address cannot be handed to another tensor in between)
        self.stats, self.key, self.ref = st, (dh.data_ptr(), dh._version, tuple(dh.shape), tuple(dh.stride())), dh

    def claim(self, dy):
        """The delivered partials if `dy` is the tensor they were computed from (one use), else None."""
        st, key = self.stats, self.key
        self.stats = self.key = self.ref = None
        if st is not None and key == (dy.data_ptr(), dy._version, tuple(dy.shape), tuple(dy.stride())):
            global BN_BYPRODUCT_CALLS
            BN_BYPRODUCT_CALLS += 1
            return st
        return None


def left_order(piece: int) -> int:
    """The layout new LEFT operands of this piece width are written in (0: with the duplicate piece, 2: without)."""
    return 2 if (LEFT_NODUP and piece >= NODUP_MIN_PIECE) else 0


def split(x, order: int, scale=None) -> Halves:
    """order 0: left operand [h1 | h1 | 2^11 h2]; order 1: right operand [h1 | h2 | 2^-11 h1] (both reduce over their columns).
    `scale`: the (s, 1/s) pair when the caller already has it (max|x| delivered by the kernels that wrote x)."""
    n, F = x.shape
    piece = (F + PIECE_ALIGN - 1) // PIECE_ALIGN * PIECE_ALIGN
    if order == 0:
        order = left_order(piece)
    if scale is None:
        scale = _C.halves_scale(x)
    return Halves(_C.halves_split(x, scale, order, piece), scale, n, F, piece, order)


RIGHT_FRAG = os.environ.get("BOT_RIGHT_FRAG", "1") != "0" and os.environ.get("BOT_NT_KERNEL", "") != "128x64"


def split_right(w, scale=None) -> Halves:
    """The right operand of `mm_nt` for a weight w [p, F]: fragment-major (order 3) when the product will run on the hand-written NT kernel
    (the same predicate `mm_nt` routes by, evaluated from w's shape: p >= NT_MIN_COLS or a left operand of this width has no duplicate
    piece), else the row-major order 1 the library formulation reads."""
    p, F = w.shape
    piece = (F + PIECE_ALIGN - 1) // PIECE_ALIGN * PIECE_ALIGN
    # scale: the (s, 1/s) pair of a matrix with the same entries (the forward split this weight's transpose: its maximum is this one's)
    if RIGHT_FRAG and NT_KERNEL == "halves3" and (w.is_cuda or FORCE) and (p >= NT_MIN_COLS or left_order(piece) == 2):
        if scale is None:
            scale = _C.halves_scale(w)
        return Halves(_C.halves_split_frag(w, scale, piece), scale, p, F, piece, 3)
    return split(w, 1, scale=scale)


_STASH = {}
STATS = {"stashed": 0, "taken": 0, "split": 0}   # how often an epilogue's halves were reused (tests, tools)


_MAX_STASH = 4    # live entries kept (each holds an operand buffer): a second model or a recompute may interleave; older ones are dropped


def stash(y, h: Halves):
    """The halves of `y` were produced together with it (fused BatchNorm epilogue): the next projection takes them from here
    instead of splitting y again.  Keyed per tensor identity (a stash of another tensor no longer destroys this one), entries
    of dead tensors are pruned, at most _MAX_STASH live ones are kept (a dropped entry costs a re-split, or - for a halves-only hidden
    state - the loud error of `take`); consumed by `take`."""
    for k in [k for k, (ref, _) in _STASH.items() if ref() is None]:
        del _STASH[k]
    while len(_STASH) >= _MAX_STASH:
        del _STASH[next(iter(_STASH))]
    _STASH[(y.data_ptr(), y._version, tuple(y.shape))] = (weakref.ref(y), h)   # y itself must still be alive when the halves are taken
    STATS["stashed"] += 1


# data_ptr of a handle's one-element base -> the base, WEAKLY: the entry lives exactly as long as the base does - i.e. as long as any
# view of it (the handle, `._base`) is alive - so a live handle is never mistaken for data (the bounded FIFO that this replaces
# could evict a live handle's base, after which `take` would have split the zeros placeholder into an all-zero operand), and an address
# recycled after the base died is not mistaken for a handle
_HANDLES = weakref.WeakValueDictionary()


def make_handle(like, n: int, F: int):
    """The stand-in for a hidden state that exists as fp16 halves only (bot_amd.nn.fused._epilogue_forward): zeros of shape [n, F] on
    ONE element (strides 0, 0) — initialised memory, the autograd edge and the key of the stashed halves.  Registered by the address
    of its base, so that `take` tells it from a caller's own broadcast tensor; the handle also carries the mark `_bot_handle`."""
    base = like.new_zeros(1)
    _HANDLES[base.data_ptr()] = base
    h = base.expand(n, F)
    h._bot_handle = True
    return h


def is_handle(x) -> bool:
    return x.dim() == 2 and x.stride(0) == 0 and x.stride(1) == 0 and (getattr(x, "_bot_handle", False) or x.data_ptr() in _HANDLES)


_SCALES = {}


def stash_scale(y, scale):
    """max|y| was delivered by the kernel that wrote y (include/bot_gnn.h "Maxima as by-products"): the (s, 1/s) pair for the split
    of y that follows — the inference layers, whose output is split by the next layer.  One entry, consumed by `split_with_stash`."""
    _SCALES.clear()
    _SCALES[(y.data_ptr(), y._version, tuple(y.shape))] = (weakref.ref(y), scale)


def split_with_stash(x, order: int) -> Halves:
    ref, scale = _SCALES.pop((x.data_ptr(), x._version, tuple(x.shape)), (None, None))
    return split(x, order, scale=scale if (scale is not None and ref() is not None) else None)


def take(x, order: int):
    ref, h = _STASH.pop((x.data_ptr(), x._version, tuple(x.shape)), (None, None))
    if h is not None and ref() is not None and h.left == (order != 1):    # a dead y: the address was recycled for another tensor
        STATS["taken"] += 1
        return h
    if is_handle(x):
        # a HANDLE (bot_amd.nn.fused._epilogue_forward: the hidden state was stored as halves only) whose halves are gone: its
        # values are placeholders, never an operand
        raise RuntimeError("bot_amd.gemm.take: this hidden state exists only as fp16 halves and they are no longer stashed "
                           "(it was produced for exactly one halves-GEMM consumer); set BOT_SKIP_Y=0 to store fp32 hidden states")
    STATS["split"] += 1
    return split(x, order)


def epilogue_piece(F: int, x) -> int | None:
    """Piece width for halves written by the BatchNorm epilogue of a [n, F] tensor, None when the fused form does not apply."""
    if MODE != "halves" or not (FORCE or (x.is_cuda and x.shape[0] >= MIN_ROWS)) or F % 2 or x.stride(0) % 2:
        return None
    return (F + PIECE_ALIGN - 1) // PIECE_ALIGN * PIECE_ALIGN


def _alpha(a: Halves, b: Halves, n=None):
    """1 / (s_a s_b) on the device, one value per output column when `n` is given (hipBLASLt's device-vector alpha): ONE launch."""
    if n is None:
        return a.scale[1:] * b.scale[1:]
    return torch.mul(a.scale[1:].expand(n), b.scale[1:].expand(n))


def mm_nt(a: Halves, b: Halves, out=None, link: BnLink | None = None):
    """a [n, F] (order 0) times b [p, F]^T (order 1) -> fp32 [n, p].  link: the result is the gradient arriving at that epilogue (BnLink)."""
    assert a.left and b.order in (1, 3) and a.F == b.F and a.piece == b.piece
    if NT_KERNEL == "halves3" and (b.order == 3 or b.n >= NT_MIN_COLS or a.order == 2):      # (an operand without the duplicate piece, or a fragment-major one, has no library form)
        st = link.stats_for(a.n, b.n, a.piece) if (link is not None and out is None) else None
        res = _C.gemm_halves3_nt(a.buf, b.buf, a.scale, b.scale, a.piece, b.piece, a.piece, out=out, a2_off=a.h2_off, b_frag=b.order == 3, n=b.n, bn=st)
        if st is not None:
            link.deliver(st, res)
        return res
    assert a.order == 0 and b.order == 1
    return _C.gemm_halves(a.buf, b.buf, _alpha(a, b, b.n), trans_b=True, out=out)


TN_NARROW = os.environ.get("BOT_GEMM_TN_NARROW", "1") != "0"
_TN_TILES = {}


def _tn_tiles(K: int, P: int):
    """The regular tile grid of a [K, P] result as a tile list of bot_gemm_halves3_tn_grouped_f32 (None: more than its 16 tiles)."""
    key = (K, P)
    if key not in _TN_TILES:
        pt = 128 if (P + 127) // 128 * ((K + 191) // 192) <= 16 else 192
        tiles = [(192 * i, min(192, K - 192 * i), pt * j, min(pt, P - pt * j), 192 * i * P + pt * j, P, 0)
                 for i in range((K + 191) // 192) for j in range((P + pt - 1) // pt)]
        _TN_TILES[key] = tiles if len(tiles) <= 16 else None
    return _TN_TILES[key]


def tn(x: Halves, d: Halves):
    """x^T d for two LEFT-operand layouts x [N, K], d [N, P] (order 0 or 2 each): the weight gradient, a reduction over the N rows.
    Row chunks of CHUNK_ROWS are batch entries (x1^T [d1 | d2] and x2^T d1 per chunk), the partial products are added
    afterwards — faster than one long-K GEMM and a pairwise-style summation (bot_amd.ops.weight_grad)."""
    assert x.left and d.left and x.n == d.n
    N, K, P, KP, PP = x.n, x.F, d.F, x.piece, d.piece
    if TN_KERNEL == "halves3" and KP * PP >= TN_MIN_OUT:      # enough 192 x 192 output tiles x row splits to fill the chip
        return _C.gemm_halves3_tn(x.buf, d.buf, x.scale, d.scale, KP, PP, K, P, x2_off=x.h2_off, d2_off=d.h2_off)
    if TN_KERNEL == "halves3" and TN_NARROW and N >= 8192:
        # a narrow result (the 40-class output layer's [750, 240]): the grouped form of the kernel takes a tile LIST, so the same grid of
        # tiles (192 x 128 when P allows) runs with as many row splits as fill the chip (8 tiles x 32 splits here, not x 8)
        tiles = _tn_tiles(K, P)
        if tiles is not None:
            out = torch.empty((K, P), dtype=torch.float32, device=x.buf.device)
            return _C.gemm_halves3_tn_grouped(x.buf, d.buf, x.scale, d.scale, x.h2_off, d.h2_off, out, tiles)
    alpha = _alpha(x, d, 2 * PP)                # one value per output column; the narrower products take a prefix
    S = max(1, N // CHUNK_ROWS)
    R = N // S
    ldx, ldd = x.buf.stride(0), d.buf.stride(0)
    x1, x2 = x.buf[:, :K], x.buf[:, x.h2_off:x.h2_off + K]
    d12, d1 = d.buf[:, d.h2_off - PP:d.h2_off + PP], d.buf[:, :PP]       # [h1 | 2^11 h2]: the last two pieces, or all of an order-2 buffer

    def part(xa, db, n, rows, batch, r0):
        return _C.gemm_halves(xa[r0:], db[r0:], alpha[:n], trans_a=True, m=K, n=n, k=rows, batch=batch,
                              strides=(rows * ldx, rows * ldd, 0))

    a = part(x1, d12, 2 * PP, R, S, 0)
    b = part(x2, d1, PP, R, S, 0)
    ra = rb = None
    if S * R < N:
        ra = part(x1, d12, 2 * PP, N - S * R, 1, S * R)
        rb = part(x2, d1, PP, N - S * R, 1, S * R)
    # both operands are LEFT layouts here: x1^T d1 + (x1^T [2^11 d2] + [2^11 x2]^T d1) / 2^11, the chunks added in chunk order —
    # one launch (bot_halves_tn_combine_f32) instead of two library reductions and the element-wise passes
    return _C.halves_tn_combine(a, b, P, ra, rb)



class _Matmul(torch.autograd.Function):
    """y = x w (kp: w is [K, P]) or y = x w^T (w is [P, K], nn.Linear's layout), forward and both gradients on the halves."""

    @staticmethod
    def forward(ctx, x, w, kp):
        xh = take(x, 0)
        ctx.kp, ctx.meta, ctx.bn_link = kp, (xh.n, xh.F, xh.piece, xh.order), xh.bn_link
        ctx.save_for_backward(xh.buf, xh.scale, w)
        ws = split_right(w.t().contiguous() if kp else w)
        ctx.wscale = ws.scale
        return mm_nt(xh, ws)

    @staticmethod
    def backward(ctx, dy):
        buf, scale, w = ctx.saved_tensors
        kp = ctx.kp
        dh = split(dy.contiguous(), 0)
        dx = mm_nt(dh, split_right(w if kp else w.t().contiguous(), scale=ctx.wscale), link=ctx.bn_link) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = tn(Halves(buf, scale, *ctx.meta), dh)               # [K, P]
            dw = dw if kp else dw.t().contiguous()
        return dx, dw, None


class _MergedLinear(torch.autograd.Function):
    """(x w^T)[:, block_i] for a weight [P, K] made of row blocks (several Linears on one input, bot_amd.nn.edge_gat): the column blocks of
    ONE GEMM come back as separate tensors, and in the backward their gradients go side by side into ONE halves operand — each block split
    in place into its column range under a common scale (bot_halves_split_cols_f16) — instead of being copied into one [n, P] buffer
    first (autograd's `cat` for a `split`: 9.5 GB read + written per layer at S-products)."""

    @staticmethod
    def forward(ctx, x, w, sizes):
        xh = take(x, 0)
        ctx.sizes, ctx.meta, ctx.bn_link = sizes, (xh.n, xh.F, xh.piece, xh.order), xh.bn_link
        ctx.save_for_backward(xh.buf, xh.scale, w)
        ws = split_right(w)
        ctx.wscale = ws.scale
        y = mm_nt(xh, ws)
        return tuple(torch.split(y, sizes, dim=1))

    @staticmethod
    def backward(ctx, *grads):
        buf, scale, w = ctx.saved_tensors
        sizes = ctx.sizes
        n, P = buf.shape[0], sum(sizes)
        piece = (P + PIECE_ALIGN - 1) // PIECE_ALIGN * PIECE_ALIGN
        grads = [None if g is None else (g if g.is_contiguous() else g.contiguous()) for g in grads]
        slots = _C.absmax_slots(buf.device)
        for g in grads:
            if g is not None:
                _C.absmax_into(g, slots)
        dscale = _C.halves_scale_from_slots(slots)
        order = left_order(piece)
        pieces = 2 if order == 2 else 3
        dbuf = torch.empty((n, pieces * piece), dtype=torch.float16, device=buf.device)
        off = 0
        for i, (g, wd) in enumerate(zip(grads, sizes)):
            width = wd if i + 1 < len(sizes) else piece - off                        # the last block also zeroes the operand's padding
            if g is None:
                for k in range(pieces):
                    dbuf[:, k * piece + off:k * piece + off + width].zero_()
            else:
                _C.halves_split_cols(g, dscale, order, dbuf, piece, off, width)
            off += wd
        dh = Halves(dbuf, dscale, n, P, piece, order)
        dx = mm_nt(dh, split_right(w.t().contiguous(), scale=ctx.wscale), link=ctx.bn_link) if ctx.needs_input_grad[0] else None
        dw = tn(Halves(buf, scale, *ctx.meta), dh).t().contiguous() if ctx.needs_input_grad[1] else None
        return dx, dw, None


def linear_blocks(x, weight, sizes):
    """The column blocks `sizes` of x [N, K] @ weight[P, K]^T as separate tensors through one GEMM and one halves operand in the backward;
    None when the shapes do not pay or a block would start on an odd column (the caller keeps `linear` + `torch.split`)."""
    if not LINEAR_BLOCKS or not worth(x, weight.shape[1], weight.shape[0]) or any(s % 2 for s in sizes[:-1]):
        return None
    return _MergedLinear.apply(x, weight, tuple(sizes))


def worth(x, K: int, P: int) -> bool:
    """Shapes for which the halves pay: enough rows, and a weight big enough that the GEMM (not the split passes) dominates."""
    return enabled(x) and min(K, P) >= 32 and K * P >= 128 * 128


def matmul(x, w):
    """x [N, K] @ w [K, P] (GraphConv's weight layout, src/no-sampling/models.py:371)."""
    if worth(x, w.shape[0], w.shape[1]):
        return _Matmul.apply(x, w, True)
    return torch.matmul(x, w)


def linear(x, weight):
    """x [N, K] @ weight[P, K]^T (nn.Linear's layout); None when the shapes do not pay (the caller keeps its own path)."""
    if worth(x, weight.shape[1], weight.shape[0]):
        return _Matmul.apply(x, weight, False)
    return None
