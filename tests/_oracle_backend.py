"""CPU stand-in for `bot_amd._C` kernel wrappers, for the `-m "not gpu"` suite ONLY.

The product has no CPU path.  To exercise the host logic above the C ABI (autograd wrappers,
permutation plumbing, layer modules, partitioned mode) in a container without a GPU, tests
monkeypatch the tensor-level wrappers of `bot_amd._C` with these functions, which restate the
contract written in include/bot_gnn.h with plain torch ops on CPU tensors.  GPU tests never install
this; they call the real kernels through the same wrappers.
"""
import torch

from oracle import ref_ops as R


def _rows(d):
    deg = (d.indptr[1:] - d.indptr[:-1]).long()
    return torch.repeat_interleave(torch.arange(d.n_rows), deg)


def _perm(p, n):
    return torch.arange(n) if p is None else p.long()


def degrees(d):
    return (d.indptr[1:] - d.indptr[:-1]).to(torch.int64)


def spmm(d, x, w=None, wperm=None, out=None, addend=None):
    xs = x[d.indices.long()]
    if w is not None:
        xs = xs * w[_perm(wperm, d.nnz)].unsqueeze(-1)
    res = torch.zeros((d.n_rows,) + tuple(x.shape[1:]), dtype=x.dtype).index_add(0, _rows(d), xs)
    if addend is not None:
        res = res + addend
    if out is not None:
        out.copy_(res)
        return out
    return res


def spmm_dot_max_d(x):
    return 1024


def spmm_dot(d, x, w, wperm, y, out=None, dot=None, absmax=None):
    res = spmm(d, x, w, wperm)
    if out is not None:
        out.copy_(res)
    else:
        out = res
    _fold_absmax(absmax, res)
    val = (x[d.indices.long()] * y[_rows(d)]).sum(-1)
    if dot is None:
        dot = torch.empty_like(val)
    dot[_perm(wperm, d.nnz)] = val
    return out, dot


def spmm_bcast(d, x, w, wperm=None, head_outer=True):
    ww = w[_perm(wperm, d.nnz)]                                  # [nnz, H]
    xs = x[d.indices.long()]                                     # [nnz, D]
    res = torch.zeros(d.n_rows, w.shape[1], x.shape[1]).index_add(0, _rows(d), ww.unsqueeze(-1) * xs.unsqueeze(1))
    return res.permute(1, 0, 2).contiguous() if head_outer else res


def spmm_bcast_halves(d, x, w, wperm, hscale, hout, col0, hsh, h2_off, hpiece):
    """include/bot_gnn.h bot_spmm_bcast_halves_f16: the aggregated slab as a halves operand ([h1 | 2^11 h2] of hscale[0] * out)."""
    res = spmm_bcast(d, x, w, wperm, head_outer=True)           # [H, n, D]
    D = x.shape[1]
    for h in range(res.shape[0]):
        z = res[h].float() * float(hscale[0])
        h1 = z.half()
        c = col0 + h * hsh
        hout[:, c:c + hpiece] = 0
        hout[:, c + h2_off:c + h2_off + hpiece] = 0
        hout[:, c:c + D] = h1
        hout[:, c + h2_off:c + h2_off + D] = ((z - h1.float()) * 2048.0).half()
    return hout


def spmm_dot_bcast(d, x, w, wperm, y, out=None):
    xs = x[:, d.indices.long(), :].permute(1, 0, 2)               # [nnz, H, D]
    ww = w[_perm(wperm, d.nnz)]
    res = torch.zeros(d.n_rows, x.shape[2]).index_add(0, _rows(d), (ww.unsqueeze(-1) * xs).sum(1))
    if out is not None:
        out.copy_(res)
    else:
        out = res
    val = (xs * y[_rows(d)].unsqueeze(1)).sum(-1)                # [nnz, H]
    dot = torch.empty_like(val)
    dot[_perm(wperm, d.nnz)] = val
    return out, dot


def sddmm_dot(d, x, y, operm=None, out=None):
    val = (x[d.indices.long()] * y[_rows(d)]).sum(-1)
    res = torch.empty_like(val)
    res[_perm(operm, d.nnz)] = val
    return res


def sddmm_dot_bcast(d, x, y, operm=None):
    rows = _rows(d)
    dots = (x[d.indices.long()].unsqueeze(0) * y[:, rows]).sum(-1).t()      # [nnz, H]
    out = torch.empty_like(dots)
    out[_perm(operm, d.nnz)] = dots
    return out


def u_add_v(src, dst, x, y=None):
    out = x[src.long()]
    if y is not None:
        out = out + y[dst.long()]
    return out


def _logits(d, el, er, ee, eperm, slope, H):
    z = torch.zeros(d.nnz, H)
    if el is not None:
        z = z + el[d.indices.long()]
    if er is not None:
        z = z + er[_rows(d)]
    if ee is not None:
        z = z + ee[_perm(eperm, d.nnz)]
    return z, torch.where(z > 0, z, z * slope)


def gat_attn_fwd(d, el, er, ee, eperm, keep, slope, H, aperm, zsign=None, drop=None):  # zsign: never allocated on CPU (zsign_buffer)
    assert drop is None or drop[0] == 0.0, "the CPU emulation has no Philox stream: attention-dropout parity is checked on the GPU only"
    z, e = _logits(d, el, er, ee, eperm, slope, H)
    rows = _rows(d)
    a = torch.zeros_like(e)
    if keep is not None:
        kept = keep[_perm(eperm, d.nnz)].bool()
        pos = torch.nonzero(kept).squeeze(1)
        a[pos] = R.edge_softmax(rows, d.n_rows, e[pos], pos)
    else:
        a = R.edge_softmax(rows, d.n_rows, e)
    out = torch.empty_like(a)
    out[_perm(aperm, d.nnz)] = a
    return out if drop is None else (out, out)


def gat_attn_bwd(d, el, er, ee, eperm, slope, H, a, da, aperm, zperm, want_der, zsign=None, drop=None):
    assert drop is None or drop[0] == 0.0
    rows = _rows(d)
    ap = _perm(aperm, d.nnz)
    a_, da_ = a[ap], da[ap]
    t = torch.zeros(d.n_rows, H).index_add(0, rows, a_ * da_)
    de = a_ * (da_ - t[rows])
    if slope != 1.0:
        z, _ = _logits(d, el, er, ee, eperm, slope, H)
        de = torch.where(z > 0, de, de * slope)
    dz = torch.empty_like(de)
    dz[_perm(zperm, d.nnz)] = de
    der = torch.zeros(d.n_rows, H).index_add(0, rows, de) if want_der else None
    return dz, der


def gat_infer(d, x, el=None, er=None, ee=None, ew=None, slope=0.2, addend=None, scale=None, shift=None, relu=False, out=None, absmax=None):
    """include/bot_gnn.h bot_gat_infer_f32, restated with plain torch ops."""
    H, D = x.shape[1], x.shape[2]
    rows = _rows(d)
    if el is not None:
        _, e = _logits(d, el.reshape(-1, H), None if er is None else er.reshape(-1, H), None if ee is None else ee.reshape(-1, H),
                       None, slope, H)
        a = R.edge_softmax(rows, d.n_rows, e)
    else:
        a = torch.ones(d.nnz, H)
    if ew is not None:
        a = a * ew.reshape(-1, 1)
    r = torch.zeros(d.n_rows, H, D).index_add(0, rows, x[d.indices.long()] * a.unsqueeze(-1))
    if addend is not None:
        r = r + addend
    r = r.reshape(d.n_rows, H * D)
    if scale is not None:
        r = r * scale
    if shift is not None:
        r = r + shift
    if relu:
        r = torch.relu(r)
    r = r.view(d.n_rows, H, D)
    _fold_absmax(absmax, r)
    if out is not None:
        out.copy_(r)
        return out
    return r


def segment_sum(d, vals, perm=None):
    return torch.zeros(d.n_rows, vals.shape[1]).index_add(0, _rows(d), vals[_perm(perm, d.nnz)])


def gather_rows(x, rows, out=None):
    res = x[rows.long()].clone()
    if out is not None:
        out.copy_(res)
        return out
    return res


def scatter_add_rows(x, rows, vals):
    x[rows.long()] += vals
    return x


def colstats(x):
    mean = x.mean(0)
    return mean, ((x - mean) ** 2).sum(0)


def colsum(x):
    return x.sum(0)


def bn_stats(x, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    n = x.shape[0]
    mean = x.mean(0)
    m2 = ((x - mean) ** 2).sum(0)
    invstd = torch.rsqrt(m2 / n + eps)
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
        running_var.mul_(1 - momentum).add_(m2 / max(n - 1, 1), alpha=momentum)
    if num_batches_tracked is not None:
        num_batches_tracked += 1
    return mean, invstd


def _bn_gate(x, mean, invstd, weight, bias, relu, p):
    assert p == 0.0, "the CPU emulation has no Philox stream: dropout parity is checked on the GPU only"
    xh = (x - mean) * invstd
    o = xh * (weight if weight is not None else 1.0) + (bias if bias is not None else 0.0)
    return xh, o


def bn_act_fwd(x, mean, invstd, weight, bias, relu, p, seed, halves=None, want_y=True):
    _, o = _bn_gate(x, mean, invstd, weight, bias, relu, p)
    y = torch.relu(o) if relu else o
    if halves is not None:                      # bot_bn_act_fwd_halves_f32: the output also as fp16 halves (want_y False: only them)
        hscale, piece = halves[:2]
        return (y if want_y else None), halves_split(y, hscale, 2 if len(halves) > 2 and halves[2] == 2 else 0, piece)
    return y


# ---- fp16 halves (include/bot_gnn.h: bot_halves_scale_f32 / bot_halves_split_f16 / bot_gemm_halves_f32), restated with torch CPU ops
def _pow2_scale(m):
    import math
    s = 1.0
    if m > 0.0 and math.isfinite(m):
        f, e = math.frexp(m)
        if f == 0.5:
            e -= 1
        s = 2.0 ** min(60, 14 - e)
    return torch.tensor([s, 1.0 / s], dtype=torch.float32)


def halves_scale(x):
    return _pow2_scale(float(x.abs().max()) if x.numel() else 0.0)


def halves_split_cols(x, scale, order, buf, piece, col, width):
    n, F = x.shape
    tmp = halves_split(torch.nn.functional.pad(x.float(), (0, width - F)), scale, order, width)      # [n, 3 * width] (order 2: two pieces)
    for k in range(2 if order == 2 else 3):
        buf[:, k * piece + col:k * piece + col + width] = tmp[:, k * width:(k + 1) * width]
    return buf


def halves_split_heads(x, scale, H, D, DP, out=None):
    n = x.shape[0]
    buf = torch.zeros((n, 2 * H * DP), dtype=torch.float16) if out is None else out
    for h in range(H):
        halves_split_cols(x[:, h * D:(h + 1) * D], scale, 2, buf, H * DP, h * DP, DP)
    return buf


def halves_tn_combine(a, b, P, rem_a=None, rem_b=None):
    if a.dim() == 2:
        a, b = a.unsqueeze(0), b.unsqueeze(0)
    PP = a.shape[2] // 2
    s1, s2, s3 = a[:, :, :P].sum(0), a[:, :, PP:PP + P].sum(0), b[:, :, :P].sum(0)
    if rem_a is not None:
        s1, s2, s3 = s1 + rem_a[:, :P], s2 + rem_a[:, PP:PP + P], s3 + rem_b[:, :P]
    return s1 + (s2 + s3) * (1.0 / 2048.0)


# maxima as by-products (include/bot_gnn.h): int32 words holding the bit patterns of non-negative floats; one slot suffices here
def absmax_slots(device):
    return torch.zeros(64, dtype=torch.int32, device=device)


def _fold_absmax(slots, t):
    if slots is not None and t.numel():
        m = t.detach().float().abs()
        m = m[~torch.isnan(m)].max() if bool((~torch.isnan(m)).any()) else m.new_zeros(())
        slots[0] = torch.maximum(slots[0], m.reshape(1).view(torch.int32)[0])


def absmax_into(x, slots):
    _fold_absmax(slots, x)
    return slots


def halves_scale_from_slots(slots, mult=None, cap=None, cap_ratio=1.0):
    """include/bot_gnn.h bot_halves_scale_from_slots_f32 / _slots2_f32"""
    sc = _pow2_scale(float(slots.max().reshape(1).view(torch.float32)[0]) * (1.0 if mult is None else float(mult)))
    if cap is not None:
        s = min(float(sc[0]), float(cap[0]) * cap_ratio)
        sc = torch.tensor([s, 1.0 / s], dtype=torch.float32)
    return sc


def halves_tail(segments, scale, out, h2_off):
    """include/bot_gnn.h bot_halves_tail_f16"""
    for col, width, src in segments:
        if src is None:
            out[:, col:col + width] = 0
            out[:, h2_off + col:h2_off + col + width] = 0
        else:
            z = src.float() * float(scale[0])
            h1 = z.half()
            out[:, col:col + width] = h1
            out[:, h2_off + col:h2_off + col + width] = ((z - h1.float()) * 2048.0).half()
    return out


def spmm_dot_halves_fits(x, y, hout, hsh, h2_off):
    return x.shape[1] >= 2


def spmm_dot_halves(d, x, w, wperm, y, hscale, hout, hsh, h2_off, dot=None):
    """include/bot_gnn.h bot_spmm_dot_halves_f16"""
    res, dot = spmm_dot(d, x, w, wperm, y, dot=dot)
    H, D = res.shape[1], res.shape[2]
    for h in range(H):
        z = res[:, h].float() * float(hscale[0])
        h1 = z.half()
        hout[:, h * hsh:h * hsh + D] = h1
        hout[:, h2_off + h * hsh:h2_off + h * hsh + D] = ((z - h1.float()) * 2048.0).half()
    return dot


def halves_split(x, scale, order, piece, out=None):
    n, F = x.shape
    z = x.float() * (float(scale[0]) if scale is not None else 1.0)
    h1 = z.half()
    r = z - h1.float()
    pieces = 2 if order == 2 else 3
    buf = torch.zeros((n, pieces * piece), dtype=torch.float16) if out is None else out
    if out is not None:
        buf[:, :pieces * piece] = 0
    buf[:, :F] = h1
    # left operands [h1 | h1 | 2^11 h2] (order 2: without the duplicate, [h1 | 2^11 h2]), right operands [h1 | h2 | 2^-11 h1]
    # (csrc/halves.hip "Dynamic range")
    if order == 2:
        buf[:, piece:piece + F] = (r * 2048.0).half()
        return buf
    buf[:, piece:piece + F] = h1 if order == 0 else r.half()
    buf[:, 2 * piece:2 * piece + F] = (r * 2048.0).half() if order == 0 else (h1.float() / 2048.0).half()
    return buf


def gemm_halves(a, b, alpha, *, trans_a=False, trans_b=False, out=None, batch=1, strides=(0, 0, 0), m=None, n=None, k=None, beta=0.0,
                ldc=None):
    if m is None:
        m, k = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    if n is None:
        n = b.shape[-2] if trans_b else b.shape[-1]
    if out is None:
        out = torch.empty((m, n) if batch == 1 else (batch, m, n), dtype=torch.float32)
    alpha = alpha.reshape(-1)
    alpha = alpha.expand(n) if alpha.numel() == 1 else alpha
    sa, sb, sc = strides
    if batch > 1 and sc == 0:
        sc = out.stride(0)
    lda, ldb = a.stride(-2), b.stride(-2)
    ldc = out.stride(-2) if ldc is None else ldc

    def view(t, rows, cols, ld, off):
        return torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset() + off)

    for i in range(batch):
        A = view(a, k, m, lda, i * sa).t() if trans_a else view(a, m, k, lda, i * sa)
        B = view(b, n, k, ldb, i * sb).t() if trans_b else view(b, k, n, ldb, i * sb)
        C = view(out, m, n, ldc, i * sc)
        res = (A.float() @ B.float()) * alpha           # fp16 x fp16 products are exact in fp32; fp32 accumulation
        C.copy_(res if beta == 0.0 else res + beta * C)
    return out


def halves_split_frag(x, scale, piece):
    """include/bot_gnn.h bot_halves_split_frag_f16"""
    n, F = x.shape
    tiles, T = (n + 15) // 16, piece // 32
    z = torch.zeros(tiles * 16, piece)
    z[:n, :F] = x.float() * (float(scale[0]) if scale is not None else 1.0)
    h1 = z.half()
    h2 = (z - h1.float()).half()

    def frag(h):       # [tiles, 16 rows, T, 4 groups, 8] -> [tiles, T, 4 groups, 16 rows, 8]: lane l = row + 16 group
        return h.view(tiles, 16, T, 4, 8).permute(0, 2, 3, 1, 4).reshape(tiles * T, 512)
    return torch.cat([frag(h1), frag(h2)])


def _unfrag(buf, n, piece):
    """the row-major [h1 | h2] of a fragment-major right operand"""
    tiles, T = (n + 15) // 16, piece // 32

    def rows(b):
        return b.view(tiles, T, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(tiles * 16, piece)[:n]
    return torch.cat([rows(buf[:tiles * T]), rows(buf[tiles * T:])], dim=1)


class BnBwdStats:
    """bot_amd._C.BnBwdStats / include/bot_gnn.h bot_bn_bwd_stats_t: the reduce pass of an epilogue's backward as a by-product of the NT product
    that writes its incoming gradient - partials per 256-row tile, finished in tile order."""

    def __init__(self, x, mean, invstd, weight, bias, relu, p, seed, want_max=True):
        self.x, self.mean, self.invstd, self.weight, self.bias, self.relu, self.p, self.seed = x, mean, invstd, weight, bias, relu, p, seed
        self.n, self.F = x.shape
        self.nblk = (self.n + 255) // 256
        self.part = torch.zeros((self.nblk, 2, self.F))
        self.pmax = torch.zeros((self.nblk, 2, self.F)) if want_max else None

    def fits(self, m, n, k):
        return m == self.n and n == self.F and self.F % 2 == 0 and k % 64 == 0

    def fill(self, dy):
        xh, o = _bn_gate(self.x, self.mean, self.invstd, self.weight, self.bias, self.relu, self.p)
        g = torch.where(o > 0, dy, torch.zeros_like(dy)) if self.relu else dy
        for t in range(self.nblk):
            r = slice(256 * t, min(256 * (t + 1), self.n))
            self.part[t, 0], self.part[t, 1] = g[r].sum(0), (g[r] * xh[r]).sum(0)
            if self.pmax is not None:
                self.pmax[t, 0], self.pmax[t, 1] = g[r].abs().max(0).values, xh[r].abs().max(0).values

    def sums(self):
        s = self.part.double().sum(0)
        return s[0].float(), s[1].float()

    def finish(self, batch_stats, total_count, slots):
        sg, sgx = self.sums()
        self.bound(sg if batch_stats else None, sgx if batch_stats else None, total_count, slots)
        return sg, sgx

    def bound(self, sum_g, sum_gx, total_count, slots):
        return bn_bwd_bound(self.pmax.max(0).values, self.n, sum_g, sum_gx, total_count, self.weight, self.invstd, slots)


def gemm_halves3_nt(a, b, scale_a, scale_b, piece_a, piece_b, k, out=None, mode=0, a2_off=None, scale_a2=None, k_split=0, b_frag=False, n=None, bn=None):
    """include/bot_gnn.h bot_gemm_halves3_nt_f32 / _nt2_f32: a1 b1^T + a1 b2^T + (2^11 a2) (2^-11 b1)^T from a LEFT and a RIGHT operand buffer;
    scale_a2: a's columns from k_split on carry a second scale (the accumulators are rescaled by the ratio in front of them)."""
    a2_off = 2 * piece_a if a2_off is None else a2_off
    if b_frag:
        b = _unfrag(b, n, piece_b)

    def part(lo, hi):
        a1, a2 = a[:, lo:hi].float(), a[:, a2_off + lo:a2_off + hi].float()
        b1, b2 = b[:, lo:hi].float(), b[:, piece_b + lo:piece_b + hi].float()
        b1s = (b[:, lo:hi] * torch.tensor(2.0 ** -11, dtype=torch.float16)).float()   # the kernel's v_pk_mul_f16: exact or rounded into fp16 subnormals
        return a1 @ b1.t() + a1 @ b2.t() + a2 @ b1s.t()
    if scale_a2 is None:
        res = part(0, k) * (scale_a[1] * scale_b[1])
    else:
        res = (part(0, k_split) * (scale_a2[0] * scale_a[1]) + part(k_split, k)) * (scale_a2[1] * scale_b[1])
    if bn is not None:
        assert bn.fits(res.shape[0], res.shape[1], k)
        bn.fill(res)
    if out is None:
        return res
    out.copy_(res)
    return out


def _three(a1, a2s, b1, b2s, right, acc=torch.float32):
    """a1 b1^T-style three-term products of fp16 blocks as the kernels form them: left x right (b2 plain, the 2^-11 on b1) or left x left
    (both second halves carry 2^11, the 2^-11 on each first half); `acc`: the accumulation type (the GPU suite checks against float64)."""
    sh = torch.tensor(2.0 ** -11, dtype=torch.float16)
    f = lambda t: t.to(acc)
    if right:
        return f(a1) @ f(b1).t() + f(a1) @ f(b2s).t() + f(a2s) @ f(b1 * sh).t()
    return f(a1).t() @ f(b1) + f(a1 * sh).t() @ f(b2s) + f(a2s).t() @ f(b1 * sh)


def gemm_halves3_nt_grouped(a, b, scale_a, scale_b, a2_off, b2_off, out, groups, k_seg, mode=0, col_scale=None, col_shift=None, relu=False, absmax=None,
                            stats=None):
    """include/bot_gnn.h bot_gemm_halves3_nt_grouped_f32 (accumulated in out's dtype)"""
    acc = out.dtype
    flat = out.as_strided((out.untyped_storage().nbytes() // out.element_size() - out.storage_offset(),), (1,))
    ld, m = out.stride(-2), a.shape[0]
    alpha = scale_a[1] * scale_b[1]
    for (b_row0, n_valid, a_col0, a_col1, k_steps, c_off) in groups:
        k0 = 32 * min(k_seg, k_steps)
        k1 = 32 * k_steps - k0
        rows = slice(b_row0, b_row0 + n_valid)
        res = torch.zeros(m, n_valid, dtype=acc)
        if k0:
            res += _three(a[:, a_col0:a_col0 + k0], a[:, a_col0 + a2_off:a_col0 + a2_off + k0], b[rows, :k0], b[rows, b2_off:b2_off + k0], True, acc)
        if k1:
            res += _three(a[:, a_col1 + k0:a_col1 + k0 + k1], a[:, a_col1 + a2_off + k0:a_col1 + a2_off + k0 + k1], b[rows, k0:k0 + k1],
                          b[rows, b2_off + k0:b2_off + k0 + k1], True, acc)
        res = res * alpha
        if col_scale is not None:
            res = res * col_scale[c_off:c_off + n_valid].to(acc)
        if col_shift is not None:
            res = res + col_shift[c_off:c_off + n_valid].to(acc)
        if relu:
            res = torch.relu(res)
        _fold_absmax(absmax, res)
        flat.as_strided((m, n_valid), (ld, 1), c_off).copy_(res)
        if stats is not None:           # per 256-row tile: its first row (the pivot), sums of (v - pivot), (v - pivot)^2, column extremes
            part, minmax, pivot = stats     # (bot_gemm_halves3_nt_grouped2_f32, ABI 19)
            for t in range((m + 255) // 256):
                v = res[256 * t:256 * (t + 1)].float()
                pivot[t, c_off:c_off + n_valid] = v[0]
                dlt = v - v[0]
                part[t, 0, c_off:c_off + n_valid], part[t, 1, c_off:c_off + n_valid] = dlt.sum(0), (dlt * dlt).sum(0)
                minmax[t, 0, c_off:c_off + n_valid], minmax[t, 1, c_off:c_off + n_valid] = v.min(0).values, v.max(0).values
    return out


def gemm_halves3_tn_grouped(x, d, scale_x, scale_d, x2_off, d2_off, out, tiles, mode=0):
    """include/bot_gnn.h bot_gemm_halves3_tn_grouped_f32 (accumulated in out's dtype)"""
    flat = out.reshape(-1)
    alpha = scale_x[1] * scale_d[1]
    for (x_col0, k_valid, d_col0, p_valid, out_off, ldo, transposed) in tiles:
        res = _three(x[:, x_col0:x_col0 + k_valid], x[:, x_col0 + x2_off:x_col0 + x2_off + k_valid], d[:, d_col0:d_col0 + p_valid],
                     d[:, d_col0 + d2_off:d_col0 + d2_off + p_valid], False, out.dtype)
        flat.as_strided((k_valid, p_valid), (1, ldo) if transposed else (ldo, 1), out_off).copy_(res * alpha)
    return out


def gemm_halves3_tn(x, d, scale_x, scale_d, piece_x, piece_d, k, p, mode=0, x2_off=None, d2_off=None, scale_d2=None, p_split=0):
    """include/bot_gnn.h bot_gemm_halves3_tn_f32: x1^T d1 + x1^T d2 + x2^T d1 of two LEFT operand buffers ([h1 | h1 | 2^11 h2], or
    [h1 | 2^11 h2] with the second-half offset = the piece width)."""
    sh = torch.tensor(2.0 ** -11, dtype=torch.float16)
    x2_off = 2 * piece_x if x2_off is None else x2_off
    d2_off = 2 * piece_d if d2_off is None else d2_off
    x1, x2s = x[:, :k], x[:, x2_off:x2_off + k]
    d1, d2s = d[:, :p], d[:, d2_off:d2_off + p]
    res = x1.float().t() @ d1.float() + (x1 * sh).float().t() @ d2s.float() + x2s.float().t() @ (d1 * sh).float()
    if scale_d2 is None:
        return res * (scale_x[1] * scale_d[1])
    alpha = torch.full((p,), float(scale_d[1]))
    alpha[p_split:] = float(scale_d2[1])
    return res * (scale_x[1] * alpha)


# ---- the train step's glue (include/bot_gnn.h v14: label_split / build_input / node_loss / rmsprop_step), restated with torch CPU ops.
# Dropout and the random split draw from torch's generator here (the kernels' Philox streams are a GPU matter: the GPU suite checks
# their rates and reproducibility); with p = 0 and a given mask everything is exact.
def label_split(train_idx, labels, mask, mask_rate, seed, use_labels, code, wn):
    keep = mask.bool() if mask is not None else torch.rand(train_idx.shape, generator=torch.Generator().manual_seed(int(seed) % (2 ** 31))) < mask_rate
    pred = ~keep if use_labels else keep
    if code is not None:
        code[train_idx] = torch.where(keep, labels.reshape(labels.shape[0], -1)[train_idx, 0], torch.full_like(train_idx, -1)).to(torch.int32)
    wn[train_idx] = pred.to(torch.float32)
    return pred.sum().to(torch.float32).reshape(1)


def build_input(feat, code, n_classes, p, seed):
    n, F = feat.shape
    onehot = torch.zeros(n, n_classes)
    if n_classes:
        rows = torch.nonzero(code >= 0)[:, 0]
        onehot[rows, code[rows].long()] = 1.0
    x = torch.cat([feat, onehot], 1)
    if p > 0:
        keep = torch.rand(x.shape, generator=torch.Generator().manual_seed(int(seed) % (2 ** 31))) >= p
        x = torch.where(keep, x / (1.0 - p), torch.zeros_like(x))
    return x


def node_loss(x, labels, wn, count, kind, eps, want_grad=True):
    import math
    n, C = x.shape
    lab = labels.reshape(labels.shape[0], -1)[:, 0].clamp(0, C - 1)
    xd = x.detach()
    lse = torch.logsumexp(xd, 1)
    ce = lse - xd.gather(1, lab[:, None])[:, 0]
    if kind == "loge":
        y, dy = torch.log(eps + ce) - math.log(eps), 1.0 / (eps + ce)
    elif kind == "savage":
        e = torch.exp(-ce)
        y, dy = (1 - e) ** 2, 2 * (1 - e) * e
    else:
        y, dy = ce, torch.ones_like(ce)
    on = wn > 0
    n_pad = (n + 63) // 64 * 64
    yo = torch.zeros(n_pad)
    yo[:n] = torch.where(on, y, torch.zeros_like(y))
    dx = None
    if want_grad:
        sm = torch.exp(xd - lse[:, None])
        sm[torch.arange(n), lab] -= 1.0
        dx = torch.where(on[:, None], sm * (dy / count[0])[:, None], torch.zeros_like(sm))
    return yo, dx


def rmsprop_step(params, grads, square_avgs, lr, alpha, eps, weight_decay, lr_dev=None):
    lr = float(lr_dev) if lr_dev is not None else lr
    for p, g, sq in zip(params, grads, square_avgs):
        if weight_decay != 0:
            g = g.add(p, alpha=weight_decay)
        sq.mul_(alpha).addcmul_(g, g, value=1 - alpha)
        p.addcdiv_(g, sq.sqrt().add_(eps), value=-lr)


def bn_stats_halves(x, eps, momentum, running_mean, running_var, num_batches_tracked, weight, bias, p):
    mean, invstd = bn_stats(x, eps, momentum, running_mean, running_var, num_batches_tracked)
    dev = torch.maximum((x.max(0).values - mean).abs(), (x.min(0).values - mean).abs()) * invstd
    bound = ((weight.abs() if weight is not None else 1.0) * dev + (bias.abs() if bias is not None else 0.0)) / (1.0 - p)
    return mean, invstd, _pow2_scale(float(bound.max()))


def bn_stats_halves_partials(part, minmax, pivot, n, eps, momentum, running_mean, running_var, num_batches_tracked, weight, bias, p):
    """include/bot_gnn.h bot_bn_stats_halves_partials_f32"""
    nblk = part.shape[0]
    nb = torch.full((nblk, 1), 256.0, dtype=torch.float64)
    nb[-1] = n - 256 * (nblk - 1)
    P0 = pivot[0].double()
    d = pivot.double() - P0             # every block re-based onto the first block's pivot (csrc/dense.hip colstats_tiles_final_kernel)
    S = (part[:, 0].double() + nb * d).sum(0)
    Q = (part[:, 1].double() + 2 * d * part[:, 0].double() + nb * d * d).sum(0)
    mean = (P0 + S / n).float()
    m2 = (Q - S * S / n).clamp(min=0).float()
    invstd = torch.rsqrt(m2 / n + eps)
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
        running_var.mul_(1 - momentum).add_(m2 / max(n - 1, 1), alpha=momentum)
    if num_batches_tracked is not None:
        num_batches_tracked += 1
    mn, mx = minmax[:, 0].min(0).values, minmax[:, 1].max(0).values
    dev = torch.maximum((mx - mean).abs(), (mn - mean).abs()) * invstd
    bound = ((weight.abs() if weight is not None else 1.0) * dev + (bias.abs() if bias is not None else 0.0)) / (1.0 - p)
    return mean, invstd, _pow2_scale(float(bound.max()))


def bn_act_bwd_reduce(dy, x, mean, invstd, weight, bias, relu, p, seed, want_max=False):
    xh, o = _bn_gate(x, mean, invstd, weight, bias, relu, p)
    g = torch.where(o > 0, dy, torch.zeros_like(dy)) if relu else dy
    if want_max:        # the "workspace" of the restatement: the two column-maxima rows
        return g.sum(0), (g * xh).sum(0), torch.stack([g.abs().max(0).values, xh.abs().max(0).values])
    return g.sum(0), (g * xh).sum(0)


def bn_bwd_bound(ws, n, sum_g, sum_gx, total_count, weight, invstd, slots):
    """include/bot_gnn.h bot_bn_bwd_bound_f32"""
    t = ws[0].clone()
    if sum_g is not None:
        t = t + sum_g.abs() / total_count + ws[1] * sum_gx.abs() / total_count
    _fold_absmax(slots, (weight.abs() if weight is not None else 1.0) * invstd * t * 1.0001)
    return slots


def bn_act_bwd_apply_halves(dy, x, mean, invstd, weight, bias, relu, p, seed, sum_g, sum_gx, total_count, hscale, hout, hD, hDP, out=None, h2_off=None):
    res = bn_act_bwd_apply(dy, x, mean, invstd, weight, bias, relu, p, seed, sum_g, sum_gx, total_count, out=out)
    if h2_off is None:
        halves_split_heads(res, hscale, x.shape[1] // hD, hD, hDP, out=hout)
    else:       # a column range of a wider operand: head blocks of hD columns every hDP, the second half h2_off columns behind
        for h in range(x.shape[1] // hD):
            z = res[:, h * hD:(h + 1) * hD].float() * float(hscale[0])
            h1 = z.half()
            hout[:, h * hDP:h * hDP + hD] = h1
            hout[:, h2_off + h * hDP:h2_off + h * hDP + hD] = ((z - h1.float()) * 2048.0).half()
    return hout


def bn_act_bwd_apply(dy, x, mean, invstd, weight, bias, relu, p, seed, sum_g, sum_gx, total_count, out=None, absmax=None):
    xh, o = _bn_gate(x, mean, invstd, weight, bias, relu, p)
    g = torch.where(o > 0, dy, torch.zeros_like(dy)) if relu else dy
    w = weight if weight is not None else 1.0
    res = w * invstd * g if sum_g is None else w * invstd * (g - sum_g / total_count - xh * sum_gx / total_count)
    _fold_absmax(absmax, res)
    if out is not None:
        out.copy_(res)
        return out
    return res


def edge_mlp_supported(I, J, H):
    return I == 8 and J == 16 and 1 <= H <= 8


def edge_mlp_fwd(ef, W1, b1, W2):
    return torch.relu(ef @ W1.t() + b1) @ W2.t()


def edge_mlp_bwd(ef, W1, b1, W2, dz):
    pre = ef @ W1.t() + b1
    r = torch.relu(pre)
    du = (dz @ W2) * (pre > 0)
    return du.t() @ ef, du.sum(0), dz.t() @ r


def random_keep(n, n_keep, seed, device):
    """Same distribution as the HIP radix select (a uniformly random n_keep-subset), not the same bits."""
    gen = torch.Generator().manual_seed(seed & 0x7FFFFFFFFFFFFFFF)
    keep = torch.zeros(n, dtype=torch.uint8)
    keep[torch.randperm(n, generator=gen)[n - n_keep:]] = 1
    return keep.to(device)


NAMES = ["BnBwdStats", "bn_stats_halves_partials", "halves_split_frag", "halves_tail", "spmm_dot_halves", "spmm_dot_halves_fits", "gemm_halves3_tn", "bn_bwd_bound", "bn_act_bwd_apply_halves", "halves_split_heads", "gemm_halves3_nt_grouped", "gemm_halves3_tn_grouped", "spmm_bcast_halves", "label_split", "build_input", "node_loss", "rmsprop_step", "gemm_halves3_nt", "halves_split_cols", "halves_tn_combine", "absmax_slots", "absmax_into", "halves_scale_from_slots", "halves_scale", "halves_split", "gemm_halves", "bn_stats_halves", "colsum", "bn_stats", "sddmm_dot_bcast", "gat_infer", "random_keep", "spmm_bcast", "spmm_dot_bcast", "edge_mlp_supported", "edge_mlp_fwd", "edge_mlp_bwd", "spmm_dot", "spmm_dot_max_d", "colstats", "bn_act_fwd", "bn_act_bwd_reduce", "bn_act_bwd_apply", "degrees", "spmm", "sddmm_dot", "u_add_v", "gat_attn_fwd", "gat_attn_bwd", "segment_sum", "gather_rows",
         "scatter_add_rows"]


def install(monkeypatch):
    from bot_amd import _C
    for n in NAMES:
        monkeypatch.setattr(_C, n, globals()[n])


def install_direct():
    """For spawned worker processes of the multi-process tests (no pytest monkeypatch there)."""
    from bot_amd import _C
    for n in NAMES:
        setattr(_C, n, globals()[n])
