// Node-axis BatchNorm + ReLU + dropout, fused (gfx950).  SURVEY §8 row f1: the epilogue of every hidden
// layer of the reference stacks (src/no-sampling/models.py:636-639, :726-731) is
//     h = BatchNorm1d(h) over ALL nodes;  h = relu(h);  h = dropout(h)
// on a [N, H*D] tensor (508 MB at BASELINE config 2).  Done with stock elementwise kernels that is ~10 HBM
// round trips per layer per step; here it is 2 reads + 1 write forward and 3 reads + 1 write backward:
//
//   colstats      : per-column mean and M2 = sum (x-mean)^2, two-stage (partials per row block, then a
//                   fixed-order combine in double) — no atomics, bitwise reproducible.
//   bn_act_fwd    : y = drop(relu((x-mean)*invstd*w + b)); the dropout keep-mask is a counter-based
//                   Philox4x32-10 stream keyed by (seed, element group), so the backward regenerates it and
//                   nothing but x is saved.
//   bn_act_bwd_red: column sums of g and g*xhat, g = dy * keep/(1-p) * [bn > 0]   (same two-stage shape)
//   bn_act_bwd_app: dx = w*invstd*(g - sum_g/n - xhat*sum_gx/n)
//
// The split at the column statistics is deliberate: in the vertex-partitioned mode the (mean, M2, count) triples
// and the (sum_g, sum_gx) pairs are what gets all-reduced between the two kernels.
// HBM roofline: algorithmic bytes = 4*n*F * {1 (stats), 2 (fwd), 2 (bwd reduce), 3 (bwd apply)}.
#include <hip/hip_fp16.h>

#include "common.h"

#include <stdlib.h>

namespace bot {

constexpr int kTX = 64;   // lanes across columns (x VEC floats each)
constexpr int kTY = 4;    // rows per block iteration
constexpr int kRowBlocks = 256;

// keep/(1-p) factors of the VEC elements starting at column c of row r.  The mask is a function of (seed, r, c) ONLY — never
// of the load width a launch happens to pick: element (r, c) takes word c % 4 of the Philox block with counter
// r * ceil(F/4) + c/4, so the forward and the two backward kernels agree even when their operands are aligned differently
// (a VEC < 4 launch simply uses a part of each block).
template <int VEC>
__device__ __forceinline__ void drop_factors(uint64_t seed, int64_t r, int c, int64_t nquad, float p, float scale, float (&f)[VEC]) {
    uint32_t w[4];
    Philox::gen(seed, (uint64_t)(r * nquad + (c >> 2)), w);
#pragma unroll
    for (int t = 0; t < VEC; ++t) f[t] = ((w[(c & 3) + t] >> 8) * (1.0f / 16777216.0f)) >= p ? scale : 0.f;
}

// ---- quad access (round 2).  A VEC = 4 launch may run on rows whose length is 4k + 2 and on operands that are only 8-byte
// aligned: `nv` of the 4 columns at c exist (4, or 2 in the tail quad); an operand whose base and row stride are 16-byte
// multiples (`wide`) moves a full quad as one float4, any other as two float2 — so e.g. F = 750 runs its 16-byte-aligned
// operands (the SpMM / GEMM-slab outputs) at full width and one Philox block per 4 elements instead of per 2.
template <int VEC>
__device__ __forceinline__ void load_cols(float (&v)[VEC], const float* p, bool wide, int nv) {
    if constexpr (VEC == 4) {
        if (wide && nv == 4) {
            vload<4>(v, p);
        } else {
            const float2 a = *reinterpret_cast<const float2*>(p);
            v[0] = a.x, v[1] = a.y, v[2] = 0.f, v[3] = 0.f;
            if (nv == 4) {
                const float2 b = *reinterpret_cast<const float2*>(p + 2);
                v[2] = b.x, v[3] = b.y;
            }
        }
    } else {
        vload<VEC>(v, p);
    }
}
template <int VEC>
__device__ __forceinline__ void store_cols(float* p, const float (&v)[VEC], bool wide, int nv) {
    if constexpr (VEC == 4) {
        if (wide && nv == 4) {
            vstore<4>(p, v);
        } else {
            *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
            if (nv == 4) *reinterpret_cast<float2*>(p + 2) = make_float2(v[2], v[3]);
        }
    } else {
        vstore<VEC>(p, v);
    }
}
// per-column parameter vector (length F, any alignment): element c + t, `dflt` past the end or when the vector is absent
template <int VEC>
__device__ __forceinline__ void load_param(float (&v)[VEC], const float* a, int c, int F, float dflt) {
#pragma unroll
    for (int t = 0; t < VEC; ++t) v[t] = (a && c + t < F) ? a[c + t] : dflt;
}

// The Philox key of a launch: the host-side `seed`, plus — when the caller passes a device word — that word times an odd
// constant.  A captured hipGraph bakes `seed` into the launch; bumping the device word between replays (one tiny captured add)
// gives every replay a fresh mask while forward and backward of the SAME replay still regenerate the same one.
__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const uint64_t* off) {
    return off ? seed + off[0] * 0x9E3779B97F4A7C15ull : seed;
}

struct BnArgs {
    const float* x;
    int64_t ldx;
    int64_t n;
    int32_t F;
    const float* mean;
    const float* invstd;
    const float* w;
    const float* b;
    int32_t relu;
    float p;
    uint64_t seed;
    const uint64_t* seed_offset;  // optional device word mixed into the seed at run time (hipGraph replays: see eff_seed)
    // fwd
    __half* hout;          // optional: the output also as fp16 halves [h1 | h1 | 2^11 h2] or, `hpieces` == 2, [h1 | 2^11 h2] (halves.hip), scaled by hscale[0]
    int32_t hpieces;       // 3: with the duplicate h1 piece the library's concatenated-axis GEMM reads; 2: without (csrc/halves3.hip reads only two)
    int64_t ldh;
    int32_t piece;
    const float* hscale;
    float* y;
    int64_t ldy;
    // bwd
    const float* dy;
    int64_t lddy;
    const float* sum_g;
    const float* sum_gx;
    float inv_count;
    float* dx;
    int64_t lddx;
    float* part;  // [kRowBlocks][2][F]
    float* pmax;  // optional (bwd reduce): [kRowBlocks][2][F] per-block column maxima of |g| and |xhat| (the operands of bn_bwd_bound_kernel)
    int32_t hD, hDP, h2off;   // bwd apply with `hout`: dx as a LEFT halves operand [h1 | 2^11 h2] (second half h2off columns behind the first), the
                              // columns in blocks of hD (a head) that start every hDP >= hD columns of the operand (halves.hip halves_split_heads_kernel)
    uint32_t* absmax;      // optional by-product of the backward apply pass: max|dx| (common.h absmax_publish)
    bool wx, wy, wdy, wdx;  // quad launches: which operands have 16-byte aligned rows (load_cols / store_cols)
};

// part[rb][0][c] = sum_{rows of block rb} (x - pivot_c),  part[rb][1][c] = sum (x - pivot_c)^2,  pivot = x[0,c]
template <int VEC>
__global__ __launch_bounds__(kTX* kTY) void colstats_partial_kernel(const float* x, int64_t ldx, int64_t n, int32_t F, float* part,
                                                                   bool wx, bool pivot, float* minmax = nullptr) {
    __shared__ float lds[2][kTY][kTX * VEC];
    const int tx = threadIdx.x % kTX, ty = threadIdx.x / kTX;
    const int c = (blockIdx.x * kTX + tx) * VEC;
    const int nv = min(VEC, F - c);
    float s[VEC], q[VEC], piv[VEC], mn[VEC], mx[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) s[t] = q[t] = 0.f, mn[t] = INFINITY, mx[t] = -INFINITY;
    if (c < F) {
        load_cols<VEC>(piv, x + c, wx, nv);
        if (!pivot) {   // plain sums (colsum): a pivot far from the column mean would only inflate the partial sums
#pragma unroll
            for (int t = 0; t < VEC; ++t) piv[t] = 0.f;
        }
#pragma unroll 4
        for (int64_t r = (int64_t)blockIdx.y * kTY + ty; r < n; r += (int64_t)gridDim.y * kTY) {
            float v[VEC];
            load_cols<VEC>(v, x + r * ldx + c, wx, nv);
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                const float d = v[t] - piv[t];
                s[t] += d;
                q[t] = fmaf(d, d, q[t]);
                mn[t] = fminf(mn[t], v[t]), mx[t] = fmaxf(mx[t], v[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < VEC; ++t) lds[0][ty][tx * VEC + t] = s[t], lds[1][ty][tx * VEC + t] = q[t];
    __syncthreads();
    if (ty == 0 && c < F) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            if (t >= nv) break;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int j = 0; j < kTY; ++j) a += lds[0][j][tx * VEC + t], b += lds[1][j][tx * VEC + t];
            part[((int64_t)blockIdx.y * 2 + 0) * F + c + t] = a;
            part[((int64_t)blockIdx.y * 2 + 1) * F + c + t] = b;
        }
    }
    if (minmax) {   // per-column extremes of the rows of this block (the caller bounds the BatchNorm output with them)
        __syncthreads();
#pragma unroll
        for (int t = 0; t < VEC; ++t) lds[0][ty][tx * VEC + t] = mn[t], lds[1][ty][tx * VEC + t] = mx[t];
        __syncthreads();
        if (ty == 0 && c < F) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                if (t >= nv) break;
                float a = lds[0][0][tx * VEC + t], b = lds[1][0][tx * VEC + t];
#pragma unroll
                for (int j = 1; j < kTY; ++j) a = fminf(a, lds[0][j][tx * VEC + t]), b = fmaxf(b, lds[1][j][tx * VEC + t]);
                minmax[((int64_t)blockIdx.y * 2 + 0) * F + c + t] = a;
                minmax[((int64_t)blockIdx.y * 2 + 1) * F + c + t] = b;
            }
        }
    }
}

// bound[c] >= max_r |dropout(relu?(BatchNorm(x)))[r, c]| from the column extremes:  (|w| max(|max - mean|, |min - mean|) invstd + |b|) / (1 - p)
__global__ __launch_bounds__(kBlock) void bn_bound_kernel(int32_t F, const float* minmax, int nblk, const float* mean, const float* invstd,
                                                         const float* w, const float* b, float p, float* bound) {
    // 64 columns x 4 row-block groups per workgroup: group g folds the partials g, g+4, ... (independent loads, unrolled)
    __shared__ float lds[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float mn = INFINITY, mx = -INFINITY;
    if (c < F) {
#pragma unroll 8
        for (int k = grp; k < nblk; k += 4) mn = fminf(mn, minmax[((int64_t)k * 2 + 0) * F + c]), mx = fmaxf(mx, minmax[((int64_t)k * 2 + 1) * F + c]);
    }
    lds[0][grp][threadIdx.x & 63] = mn, lds[1][grp][threadIdx.x & 63] = mx;
    __syncthreads();
    if (grp != 0 || c >= F) return;
#pragma unroll
    for (int g = 1; g < 4; ++g) mn = fminf(mn, lds[0][g][threadIdx.x]), mx = fmaxf(mx, lds[1][g][threadIdx.x]);
    const float dev = fmaxf(fabsf(mx - mean[c]), fabsf(mn - mean[c])) * invstd[c];
    bound[c] = (fabsf(w ? w[c] : 1.f) * dev + fabsf(b ? b[c] : 0.f)) / (1.f - p);
}

// Second stage of the column reductions: 64 columns x 4 row-groups per workgroup; group g adds the partials
// b = g, g+4, g+8, ... in double, the four group sums are combined in fixed order through LDS (deterministic).
__device__ __forceinline__ void pair_reduce(const float* part, int nblk, int32_t F, int c, int grp, double (&lds)[2][4][64],
                                            double& S, double& Q) {
    double s = 0.0, q = 0.0;
    if (c < F) {
        int b = grp;
        for (; b + 28 < nblk; b += 32) {     // eight partial pairs in flight, added in the same order (one dependent load per term: 20 us a launch)
            float vs[8], vq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) vs[j] = part[((int64_t)(b + 4 * j) * 2 + 0) * F + c], vq[j] = part[((int64_t)(b + 4 * j) * 2 + 1) * F + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)vs[j], q += (double)vq[j];
        }
        for (; b < nblk; b += 4) s += (double)part[((int64_t)b * 2 + 0) * F + c], q += (double)part[((int64_t)b * 2 + 1) * F + c];
    }
    lds[0][grp][threadIdx.x & 63] = s;
    lds[1][grp][threadIdx.x & 63] = q;
    __syncthreads();
    S = Q = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) S += lds[0][g][threadIdx.x & 63], Q += lds[1][g][threadIdx.x & 63];
}

// mean[c] = pivot + S/n ; m2[c] = Q - S^2/n.  With `invstd` given this is the whole training-mode statistics step of
// nn.BatchNorm1d in one launch (it replaced nine elementwise launches per layer): invstd = rsqrt(m2/n + eps), and the running
// statistics move by `momentum` towards the batch mean / the UNBIASED batch variance; the step counter is bumped by one.
// `bound` (with minmax [nblk][2][F] = the column extremes per row block): bn_bound_kernel's stage in the same launch (round 5: the statistics'
// second stage was three launches in a row of ~14 us each).
__global__ __launch_bounds__(kBlock) void colstats_final_kernel(const float* x, int64_t n, int32_t F, const float* part, int nblk,
                                                               float* mean, float* m2, float* invstd, float eps, float momentum,
                                                               float* running_mean, float* running_var, int64_t* num_batches,
                                                               const float* minmax = nullptr, const float* bw = nullptr, const float* bb = nullptr,
                                                               float bp = 0.f, float* bound = nullptr) {
    __shared__ double lds[2][4][64];
    __shared__ float ldm[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    if (bound) {                         // (uniform) group g folds the extremes of blocks g, g + 4, ...
        float mn = INFINITY, mx = -INFINITY;
        if (c < F) {
#pragma unroll 8
            for (int k = grp; k < nblk; k += 4) mn = fminf(mn, minmax[((int64_t)k * 2 + 0) * F + c]), mx = fmaxf(mx, minmax[((int64_t)k * 2 + 1) * F + c]);
        }
        ldm[0][grp][threadIdx.x & 63] = mn, ldm[1][grp][threadIdx.x & 63] = mx;
    }
    double S, Q;
    pair_reduce(part, nblk, F, c, grp, lds, S, Q);
    if (grp == 0 && c < F) {
        const float mu = (float)((double)x[c] + S / (double)n);
        const float q = (float)fmax(Q - S * S / (double)n, 0.0);
        mean[c] = mu;
        if (m2) m2[c] = q;
        if (bound) {                     // (pair_reduce's barrier made ldm visible) bn_bound_kernel's expression on this column's statistics
            float mn = ldm[0][0][threadIdx.x], mx = ldm[1][0][threadIdx.x];
#pragma unroll
            for (int g = 1; g < 4; ++g) mn = fminf(mn, ldm[0][g][threadIdx.x]), mx = fmaxf(mx, ldm[1][g][threadIdx.x]);
            const float is = rsqrtf(q / (float)n + eps);
            const float dev = fmaxf(fabsf(mx - mu), fabsf(mn - mu)) * is;
            bound[c] = (fabsf(bw ? bw[c] : 1.f) * dev + fabsf(bb ? bb[c] : 0.f)) / (1.f - bp);
        }
        if (invstd) {
            invstd[c] = rsqrtf(q / (float)n + eps);
            if (running_mean) {
                running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mu;
                running_var[c] = running_var[c] * (1.f - momentum) + momentum * (q / (float)(n > 1 ? n - 1 : 1));
            }
        }
    }
    if (num_batches && blockIdx.x == 0 && threadIdx.x == 0) num_batches[0] += 1;
}

// colstats_final_kernel for partials whose pivot differs PER ROW BLOCK (v19: the grouped NT launch's by-product takes each 256-row tile's
// first value as that tile's pivot): block b holds n_b rows, pivot p_b, S_b = sum (v - p_b), Q_b = sum (v - p_b)^2 (fp32 sums of 256 well
// centred terms).  Every block is re-based onto the first block's pivot P0 in double - S'_b = S_b + n_b d, Q'_b = Q_b + 2 d S_b + n_b d^2,
// d = p_b - P0: exact algebra, and P0 is a value of the column, a few standard deviations from its mean whatever the mean is - then the
// blocks are added in the fixed (group, block) order of pair_reduce and finished like the pass form: mean = P0 + S / n, m2 = Q - S^2 / n.
// 16 columns x 64 row-block groups per workgroup (1024 threads; F = 750: 47 workgroups - with 64 columns x 4 groups the 12 workgroups were bound
// by what 12 CUs can pull from the L2: 146 us; a wave reads 4 blocks x 64 bytes per instruction)
constexpr int kTileCols = 16, kTileGroups = 64;
__global__ __launch_bounds__(kTileCols * kTileGroups) void colstats_tiles_final_kernel(const float* pivot, int64_t n, int32_t F, const float* part, int nblk,
                                                                               int block_rows, float* mean, float* invstd, float eps, float momentum,
                                                                               float* running_mean, float* running_var, int64_t* num_batches,
                                                                               const float* minmax, const float* bw, const float* bb, float bp,
                                                                               float* bound) {
    __shared__ double lds[2][kTileGroups][kTileCols];
    __shared__ float ldm[2][kTileGroups][kTileCols];
    const int lc = threadIdx.x % kTileCols, c = blockIdx.x * kTileCols + lc, grp = threadIdx.x / kTileCols;
    float mn = INFINITY, mx = -INFINITY;
    double s = 0.0, q = 0.0;
    const float P0f = c < F ? pivot[c] : 0.f;
    const double P0 = (double)P0f;
    const int last = nblk - 1;
    const double n_last = (double)(n - (int64_t)last * block_rows), n_full = (double)block_rows;
    if (c < F) {
        constexpr int U = 4;                    // four blocks in flight per thread
        int b = grp;
        for (; b + (U - 1) * kTileGroups < nblk; b += U * kTileGroups) {
            float vs[U], vq[U], vp[U], v0[U], v1[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int64_t o = ((int64_t)(b + kTileGroups * j) * 2) * F + c;
                vs[j] = part[o], vq[j] = part[o + F], vp[j] = pivot[(int64_t)(b + kTileGroups * j) * F + c], v0[j] = minmax[o], v1[j] = minmax[o + F];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const double nb = (b + kTileGroups * j == last) ? n_last : n_full, d = (double)vp[j] - P0;
                s += (double)vs[j] + nb * d;
                q += (double)vq[j] + 2.0 * d * (double)vs[j] + nb * d * d;
                mn = fminf(mn, v0[j]), mx = fmaxf(mx, v1[j]);
            }
        }
        for (; b < nblk; b += kTileGroups) {
            const int64_t o = ((int64_t)b * 2) * F + c;
            const double nb = (b == last) ? n_last : n_full, d = (double)pivot[(int64_t)b * F + c] - P0;
            const double sb = (double)part[o];
            s += sb + nb * d;
            q += (double)part[o + F] + 2.0 * d * sb + nb * d * d;
            mn = fminf(mn, minmax[o]), mx = fmaxf(mx, minmax[o + F]);
        }
    }
    lds[0][grp][lc] = s, lds[1][grp][lc] = q, ldm[0][grp][lc] = mn, ldm[1][grp][lc] = mx;
    __syncthreads();
    if (grp == 0 && c < F) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int g = 0; g < kTileGroups; ++g) S += lds[0][g][lc], Q += lds[1][g][lc];     // fixed (group, block) order: deterministic
#pragma unroll
        for (int g = 1; g < kTileGroups; ++g) mn = fminf(mn, ldm[0][g][lc]), mx = fmaxf(mx, ldm[1][g][lc]);
        const float mu = (float)(P0 + S / (double)n);
        const float m2 = (float)fmax(Q - S * S / (double)n, 0.0);
        const float is = rsqrtf(m2 / (float)n + eps);
        mean[c] = mu, invstd[c] = is;
        if (bound) {                     // bn_bound_kernel's expression on this column's statistics
            const float dev = fmaxf(fabsf(mx - mu), fabsf(mn - mu)) * is;
            bound[c] = (fabsf(bw ? bw[c] : 1.f) * dev + fabsf(bb ? bb[c] : 0.f)) / (1.f - bp);
        }
        if (running_mean) {
            running_mean[c] = running_mean[c] * (1.f - momentum) + momentum * mu;
            running_var[c] = running_var[c] * (1.f - momentum) + momentum * (m2 / (float)(n > 1 ? n - 1 : 1));
        }
    }
    if (num_batches && blockIdx.x == 0 && threadIdx.x == 0) num_batches[0] += 1;
}

// sum[c] = the unshifted partial sums added in double: the column SUM (a bias gradient).  Going through the pivot-shifted
// fp32 mean instead loses ~eps * |pivot| * n — 2e-7 on gradients whose true sum is 0 at n = 2.45 M.
__global__ __launch_bounds__(kBlock) void colsum_final_kernel(const float* part, int32_t F, int nblk, float* sum) {
    __shared__ double lds[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    double S, Q;
    pair_reduce(part, nblk, F, c, grp, lds, S, Q);
    if (grp == 0 && c < F) sum[c] = (float)S;
}

template <int VEC>
__global__ __launch_bounds__(kTX* kTY) void bn_act_fwd_kernel(BnArgs a) {
    const int tx = threadIdx.x % kTX, ty = threadIdx.x / kTX;
    const int c = (blockIdx.x * kTX + tx) * VEC;
    if constexpr (VEC == 4) {
        if (a.hout && c >= a.F) {   // zero padding of the three pieces
            if (c < a.piece)
                for (int64_t r = (int64_t)blockIdx.y * kTY + ty; r < a.n; r += (int64_t)gridDim.y * kTY) {
                    __half* o = a.hout + r * a.ldh + c;
                    const uint2 z = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(o) = z, *reinterpret_cast<uint2*>(o + a.piece) = z;
                    if (a.hpieces == 3) *reinterpret_cast<uint2*>(o + 2 * (int64_t)a.piece) = z;
                }
            return;
        }
    }
    if (c >= a.F) return;
    const int nv = min(VEC, a.F - c);
    const float hs = a.hout ? a.hscale[0] : 1.f;
    float mu[VEC], sc[VEC], sh[VEC], wv[VEC];
    load_param<VEC>(mu, a.mean, c, a.F, 0.f);
    load_param<VEC>(sc, a.invstd, c, a.F, 0.f);
    load_param<VEC>(wv, a.w, c, a.F, 1.f);
    load_param<VEC>(sh, a.b, c, a.F, 0.f);
#pragma unroll
    for (int t = 0; t < VEC; ++t) sc[t] *= wv[t];
    const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
    const int64_t nquad = (a.F + 3) / 4;
    const uint64_t seed = eff_seed(a.seed, a.seed_offset);
    constexpr int UR = 4;  // rows in flight per thread: loads first, then compute + store (x and y may alias for the compiler)
    const int64_t step = (int64_t)gridDim.y * kTY;
    for (int64_t r0 = (int64_t)blockIdx.y * kTY + ty; r0 < a.n; r0 += step * UR) {
        float v[UR][VEC];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t r = r0 + u * step;
            if (r < a.n) load_cols<VEC>(v[u], a.x + r * a.ldx + c, a.wx, nv);
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int64_t r = r0 + u * step;
            if (r >= a.n) break;
            float f[VEC];
            if (a.p > 0.f) drop_factors<VEC>(seed, r, c, nquad, a.p, scale, f);
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                float o = fmaf(v[u][t] - mu[t], sc[t], sh[t]);
                if (a.relu) o = fmaxf(o, 0.f);
                if (a.p > 0.f) o *= f[t];
                v[u][t] = o;
            }
            if (a.y) store_cols<VEC>(a.y + r * a.ldy + c, v[u], a.wy, nv);     // NULL: only the halves are wanted (bot_bn_act_fwd_halves_f32)
            if constexpr (VEC == 4) {
                if (a.hout) {
                    __half h1[4], h2[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float z = t < nv ? v[u][t] * hs : 0.f;
                        h1[t] = __float2half_rn(z);
                        h2[t] = __float2half_rn((z - __half2float(h1[t])) * kHalvesShift);   // left operand: [h1 | h1 | 2^11 h2]
                    }
                    __half* o = a.hout + r * a.ldh + c;
                    const uint2 hi = *reinterpret_cast<const uint2*>(h1), lo = *reinterpret_cast<const uint2*>(h2);
                    *reinterpret_cast<uint2*>(o) = hi;
                    if (a.hpieces == 3) *reinterpret_cast<uint2*>(o + a.piece) = hi, *reinterpret_cast<uint2*>(o + 2 * (int64_t)a.piece) = lo;
                    else *reinterpret_cast<uint2*>(o + a.piece) = lo;
                }
            }
        }
    }
}

// Row-segment form of the forward pass when the halves are written (round 3): one workgroup per kRowSeg CONSECUTIVE rows, a thread per 4
// columns of the whole padded row — the launch shape of halves_split_kernel.  The column-tiled kernel above (256 row blocks striding
// through the matrix) wrote the three pieces at 3.7 TB/s; 21 k small workgroups that each stream 24 KB in and 36 KB out reach 5.4 TB/s
// (0.349 -> 0.237 ms at [169 343, 752]; 1, 2, 4, 16 rows per workgroup and looping workgroups of 16 ... 256 rows all measured slower;
// the backward apply pass in the same shape measured 0.302 vs 0.305 ms: it writes ONE contiguous row, and stays column-tiled).
constexpr int kRowSeg = 8;
__global__ __launch_bounds__(256) void bn_act_fwd_rowseg_kernel(BnArgs a) {
    const int c = threadIdx.x * 4;
    if (c >= a.piece) return;
    const bool live = c < a.F;
    const int nv = live ? min(4, a.F - c) : 0;
    const float hs = a.hscale[0];
    float mu[4], sc[4], sh[4], wv[4];
    load_param<4>(mu, a.mean, c, a.F, 0.f);
    load_param<4>(sc, a.invstd, c, a.F, 0.f);
    load_param<4>(wv, a.w, c, a.F, 1.f);
    load_param<4>(sh, a.b, c, a.F, 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) sc[t] *= wv[t];
    const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
    const int64_t nquad = (a.F + 3) / 4;
    const uint64_t seed = eff_seed(a.seed, a.seed_offset);
    const int64_t r0 = (int64_t)blockIdx.x * kRowSeg;
    float v[kRowSeg][4];
#pragma unroll
    for (int u = 0; u < kRowSeg; ++u)
        if (live && r0 + u < a.n) load_cols<4>(v[u], a.x + (r0 + u) * a.ldx + c, a.wx, nv);
#pragma unroll
    for (int u = 0; u < kRowSeg; ++u) {
        const int64_t r = r0 + u;
        if (r >= a.n) break;
        __half h1[4], h2[4];
        float f[4];
        if (live && a.p > 0.f) drop_factors<4>(seed, r, c, nquad, a.p, scale, f);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float o = 0.f;
            if (t < nv) {
                o = fmaf(v[u][t] - mu[t], sc[t], sh[t]);
                if (a.relu) o = fmaxf(o, 0.f);
                if (a.p > 0.f) o *= f[t];
            }
            const float z = o * hs;
            h1[t] = __float2half_rn(z);
            h2[t] = __float2half_rn((z - __half2float(h1[t])) * kHalvesShift);   // left operand: [h1 | h1 | 2^11 h2]
            v[u][t] = o;
        }
        if (a.y && live) store_cols<4>(a.y + r * a.ldy + c, v[u], a.wy, nv);
        __half* o = a.hout + r * a.ldh + c;
        const uint2 hi = *reinterpret_cast<const uint2*>(h1), lo = *reinterpret_cast<const uint2*>(h2);
        *reinterpret_cast<uint2*>(o) = hi;
        if (a.hpieces == 3) *reinterpret_cast<uint2*>(o + a.piece) = hi, *reinterpret_cast<uint2*>(o + 2 * (int64_t)a.piece) = lo;
        else *reinterpret_cast<uint2*>(o + a.piece) = lo;
    }
}

// g = dy * keep/(1-p) * [bn > 0];  partial column sums of g and g*xhat
template <int VEC>
__global__ __launch_bounds__(kTX* kTY) void bn_act_bwd_reduce_kernel(BnArgs a) {
    __shared__ float lds[2][kTY][kTX * VEC];
    const int tx = threadIdx.x % kTX, ty = threadIdx.x / kTX;
    const int c = (blockIdx.x * kTX + tx) * VEC;
    float s[VEC], q[VEC], gm[VEC], xm[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) s[t] = q[t] = gm[t] = xm[t] = 0.f;
    const int nv = min(VEC, a.F - c);
    if (c < a.F) {
        float mu[VEC], is[VEC], sc[VEC], sh[VEC];
        load_param<VEC>(mu, a.mean, c, a.F, 0.f);
        load_param<VEC>(is, a.invstd, c, a.F, 0.f);
        load_param<VEC>(sc, a.w, c, a.F, 1.f);
        load_param<VEC>(sh, a.b, c, a.F, 0.f);
        const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
        const int64_t nquad = (a.F + 3) / 4;
        const uint64_t seed = eff_seed(a.seed, a.seed_offset);
#pragma unroll 4
        for (int64_t r = (int64_t)blockIdx.y * kTY + ty; r < a.n; r += (int64_t)gridDim.y * kTY) {
            float v[VEC], g[VEC], f[VEC];
            load_cols<VEC>(v, a.x + r * a.ldx + c, a.wx, nv);
            load_cols<VEC>(g, a.dy + r * a.lddy + c, a.wdy, nv);
            if (a.p > 0.f) drop_factors<VEC>(seed, r, c, nquad, a.p, scale, f);
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                const float xh = (v[t] - mu[t]) * is[t];
                float gg = g[t];
                if (a.p > 0.f) gg *= f[t];
                if (a.relu && !(fmaf(xh, sc[t], sh[t]) > 0.f)) gg = 0.f;
                s[t] += gg;
                q[t] = fmaf(gg, xh, q[t]);
                if (t < nv) gm[t] = fmaxf(gm[t], fabsf(gg)), xm[t] = fmaxf(xm[t], fabsf(xh));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < VEC; ++t) lds[0][ty][tx * VEC + t] = s[t], lds[1][ty][tx * VEC + t] = q[t];
    __syncthreads();
    if (ty == 0 && c < a.F) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            if (t >= nv) break;
            float u = 0.f, w = 0.f;
#pragma unroll
            for (int j = 0; j < kTY; ++j) u += lds[0][j][tx * VEC + t], w += lds[1][j][tx * VEC + t];
            a.part[((int64_t)blockIdx.y * 2 + 0) * a.F + c + t] = u;
            a.part[((int64_t)blockIdx.y * 2 + 1) * a.F + c + t] = w;
        }
    }
    if (a.pmax) {                        // (uniform) the same two-stage shape for the column maxima
        __syncthreads();
#pragma unroll
        for (int t = 0; t < VEC; ++t) lds[0][ty][tx * VEC + t] = gm[t], lds[1][ty][tx * VEC + t] = xm[t];
        __syncthreads();
        if (ty == 0 && c < a.F) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) {
                if (t >= nv) break;
                float u = 0.f, w = 0.f;
#pragma unroll
                for (int j = 0; j < kTY; ++j) u = fmaxf(u, lds[0][j][tx * VEC + t]), w = fmaxf(w, lds[1][j][tx * VEC + t]);
                a.pmax[((int64_t)blockIdx.y * 2 + 0) * a.F + c + t] = u;
                a.pmax[((int64_t)blockIdx.y * 2 + 1) * a.F + c + t] = w;
            }
        }
    }
}

// max_c |dx_c| bounded BEFORE dx exists:  dx = w invstd (g - mean g - xhat mean(g xhat))  =>
//   |dx_{r,c}| <= |w_c| invstd_c (max_r |g| + |sum_g_c| / n + max_r |xhat| |sum_gx_c| / n)       (sum_g == NULL, eval statistics: |w_c| invstd_c max_r |g|)
// with the column maxima of the reduce pass (pmax, nblk row blocks) and the FINAL column sums (after a cross-rank reduction, if any).
// The maximum over the columns goes to the by-product slots: halves_scale_from_slots turns it into the scale under which
// bn_act_bwd_apply writes dx directly as a halves operand (the format keeps 22 bits for entries down to 2^-28 of the scale: a bound that
// is loose by a few binades costs nothing).
__global__ __launch_bounds__(kBlock) void bn_bwd_bound_kernel(int32_t F, const float* pmax, int nblk, const float* sum_g, const float* sum_gx, float inv_count,
                                                             const float* w, const float* invstd, uint32_t* slots) {
    // 64 columns x 4 row-block groups per workgroup (the shape of pair_reduce), eight loads in flight per thread
    __shared__ float lds[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float gm = 0.f, xm = 0.f;
    if (c < F) {
        int k = grp;
        for (; k + 28 < nblk; k += 32) {
            float vg[8], vx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) vg[j] = pmax[((int64_t)(k + 4 * j) * 2 + 0) * F + c], vx[j] = pmax[((int64_t)(k + 4 * j) * 2 + 1) * F + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) gm = fmaxf(gm, vg[j]), xm = fmaxf(xm, vx[j]);
        }
        for (; k < nblk; k += 4) gm = fmaxf(gm, pmax[((int64_t)k * 2 + 0) * F + c]), xm = fmaxf(xm, pmax[((int64_t)k * 2 + 1) * F + c]);
    }
    lds[0][grp][threadIdx.x & 63] = gm, lds[1][grp][threadIdx.x & 63] = xm;
    __syncthreads();
    float b = 0.f;
    if (grp == 0 && c < F) {
#pragma unroll
        for (int g = 1; g < 4; ++g) gm = fmaxf(gm, lds[0][g][threadIdx.x & 63]), xm = fmaxf(xm, lds[1][g][threadIdx.x & 63]);
        float t = gm;
        if (sum_g) t += fabsf(sum_g[c]) * inv_count + xm * fabsf(sum_gx[c]) * inv_count;
        b = fabsf(w ? w[c] : 1.f) * invstd[c] * t * 1.0001f;          // (rounding of the three-term expression itself)
    }
    absmax_publish(wave_absmax(b), slots);
}

__global__ __launch_bounds__(kBlock) void pair_final_kernel(int32_t F, const float* part, int nblk, float* s0, float* s1) {
    __shared__ double lds[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    double A, B;
    pair_reduce(part, nblk, F, c, grp, lds, A, B);
    if (grp == 0 && c < F) {
        s0[c] = (float)A;
        s1[c] = (float)B;
    }
}

// v18: the second stage over MANY row blocks (one per 256-row GEMM tile: 662 at config 2, against <= 256 of the pass form) - 32 columns x
// 16 block groups per workgroup; group g adds blocks g, g + 16, ... in order (eight pairs in flight), the groups are combined in order.
// (8 KB of LDS: these launches run beside the side stream's weight-gradient product, whose workgroups leave 16 KB per CU)
constexpr int kWideGroups = 16, kWideCols = 32;     // (round 6 tried 16 columns x 32 groups = 47 workgroups: 91 -> 183 us per step beside the side stream's weight-gradient workgroups)
__global__ __launch_bounds__(kWideCols * kWideGroups) void pair_final_wide_kernel(int32_t F, const float* part, int nblk, float* s0, float* s1) {
    __shared__ double lds[2][kWideGroups][kWideCols];
    const int lc = threadIdx.x % kWideCols, c = blockIdx.x * kWideCols + lc, grp = threadIdx.x / kWideCols;
    double s = 0.0, q = 0.0;
    if (c < F) {
        int b = grp;
        for (; b + 7 * kWideGroups < nblk; b += 8 * kWideGroups) {
            float vs[8], vq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                vs[j] = part[((int64_t)(b + kWideGroups * j) * 2 + 0) * F + c], vq[j] = part[((int64_t)(b + kWideGroups * j) * 2 + 1) * F + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)vs[j], q += (double)vq[j];
        }
        for (; b < nblk; b += kWideGroups) s += (double)part[((int64_t)b * 2 + 0) * F + c], q += (double)part[((int64_t)b * 2 + 1) * F + c];
    }
    lds[0][grp][lc] = s, lds[1][grp][lc] = q;
    __syncthreads();
    if (grp == 0 && c < F) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int g = 0; g < kWideGroups; ++g) S += lds[0][g][lc], Q += lds[1][g][lc];
        s0[c] = (float)S, s1[c] = (float)Q;
    }
}

__global__ __launch_bounds__(kWideCols * kWideGroups) void bn_bwd_bound_wide_kernel(int32_t F, const float* pmax, int nblk, const float* sum_g, const float* sum_gx,
                                                                                   float inv_count, const float* w, const float* invstd, uint32_t* slots) {
    __shared__ float lds[2][kWideGroups][kWideCols];
    const int lc = threadIdx.x % kWideCols, c = blockIdx.x * kWideCols + lc, grp = threadIdx.x / kWideCols;
    float gm = 0.f, xm = 0.f;
    if (c < F) {
        int k = grp;
        for (; k + 7 * kWideGroups < nblk; k += 8 * kWideGroups) {
            float vg[8], vx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                vg[j] = pmax[((int64_t)(k + kWideGroups * j) * 2 + 0) * F + c], vx[j] = pmax[((int64_t)(k + kWideGroups * j) * 2 + 1) * F + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) gm = fmaxf(gm, vg[j]), xm = fmaxf(xm, vx[j]);
        }
        for (; k < nblk; k += kWideGroups) gm = fmaxf(gm, pmax[((int64_t)k * 2 + 0) * F + c]), xm = fmaxf(xm, pmax[((int64_t)k * 2 + 1) * F + c]);
    }
    lds[0][grp][lc] = gm, lds[1][grp][lc] = xm;
    __syncthreads();
    float b = 0.f;
    if (grp == 0 && c < F) {
#pragma unroll
        for (int g = 1; g < kWideGroups; ++g) gm = fmaxf(gm, lds[0][g][lc]), xm = fmaxf(xm, lds[1][g][lc]);
        float t = gm;
        if (sum_g) t += fabsf(sum_g[c]) * inv_count + xm * fabsf(sum_gx[c]) * inv_count;
        b = fabsf(w ? w[c] : 1.f) * invstd[c] * t * 1.0001f;          // (bn_bwd_bound_kernel's expression)
    }
    if (threadIdx.x < 64) absmax_publish(wave_absmax(b), slots);       // (the first wave holds groups 0 and 1: b is 0 outside group 0)
}

// pair_final_wide_kernel + bn_bwd_bound_wide_kernel for one rank (the bound uses the sums this launch has just formed)
__global__ __launch_bounds__(kWideCols * kWideGroups) void bn_bwd_finish_wide_kernel(int32_t F, const float* part, const float* pmax, int nblk, float* s0, float* s1,
                                                                                    float inv_count, bool batch_stats, const float* w, const float* invstd,
                                                                                    uint32_t* slots) {
    __shared__ double lds[2][kWideGroups][kWideCols];
    __shared__ float ldm[2][kWideGroups][kWideCols];
    const int lc = threadIdx.x % kWideCols, c = blockIdx.x * kWideCols + lc, grp = threadIdx.x / kWideCols;
    double s = 0.0, q = 0.0;
    float gm = 0.f, xm = 0.f;
    if (c < F) {
        int b = grp;
        for (; b + 3 * kWideGroups < nblk; b += 4 * kWideGroups) {      // (eight in flight instead of four: 41 -> 78 us beside a weight-gradient workgroup, round 6)
            float vs[4], vq[4], vg[4], vx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t o = ((int64_t)(b + kWideGroups * j) * 2) * F + c;
                vs[j] = part[o], vq[j] = part[o + F], vg[j] = pmax[o], vx[j] = pmax[o + F];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s += (double)vs[j], q += (double)vq[j], gm = fmaxf(gm, vg[j]), xm = fmaxf(xm, vx[j]);
        }
        for (; b < nblk; b += kWideGroups) {
            const int64_t o = ((int64_t)b * 2) * F + c;
            s += (double)part[o], q += (double)part[o + F], gm = fmaxf(gm, pmax[o]), xm = fmaxf(xm, pmax[o + F]);
        }
    }
    lds[0][grp][lc] = s, lds[1][grp][lc] = q, ldm[0][grp][lc] = gm, ldm[1][grp][lc] = xm;
    __syncthreads();
    float bnd = 0.f;
    if (grp == 0 && c < F) {
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int g = 0; g < kWideGroups; ++g) S += lds[0][g][lc], Q += lds[1][g][lc];
#pragma unroll
        for (int g = 1; g < kWideGroups; ++g) gm = fmaxf(gm, ldm[0][g][lc]), xm = fmaxf(xm, ldm[1][g][lc]);
        const float fs = (float)S, fq = (float)Q;
        s0[c] = fs, s1[c] = fq;
        float t = gm;
        if (batch_stats) t += fabsf(fs) * inv_count + xm * fabsf(fq) * inv_count;
        bnd = fabsf(w ? w[c] : 1.f) * invstd[c] * t * 1.0001f;        // (bn_bwd_bound_kernel's expression)
    }
    if (threadIdx.x < 64) absmax_publish(wave_absmax(bnd), slots);
}

template <int VEC>
__global__ __launch_bounds__(kTX* kTY) void bn_act_bwd_apply_kernel(BnArgs a) {
    const int tx = threadIdx.x % kTX, ty = threadIdx.x / kTX;
    const int c = (blockIdx.x * kTX + tx) * VEC;
    float amax = 0.f;
    if (c < a.F) {                       // (no early return: every lane takes part in the absmax reduction below)
        const int nv = min(VEC, a.F - c);
        float mu[VEC], is[VEC], sc[VEC], sh[VEC], mg[VEC], mgx[VEC];
        load_param<VEC>(mu, a.mean, c, a.F, 0.f);
        load_param<VEC>(is, a.invstd, c, a.F, 0.f);
        load_param<VEC>(sc, a.w, c, a.F, 1.f);
        load_param<VEC>(sh, a.b, c, a.F, 0.f);
        load_param<VEC>(mg, a.sum_g, c, a.F, 0.f);                    // NULL sums: eval mode (running statistics are constants)
        load_param<VEC>(mgx, a.sum_gx, c, a.F, 0.f);
#pragma unroll
        for (int t = 0; t < VEC; ++t) mg[t] *= a.inv_count, mgx[t] *= a.inv_count;
        const float scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
        const int64_t nquad = (a.F + 3) / 4;
        const uint64_t seed = eff_seed(a.seed, a.seed_offset);
        constexpr int UR = 4;
        const int64_t step = (int64_t)gridDim.y * kTY;
        // halves output (a.hout): destination column of each column pair of this thread (fixed over the rows) and whether the thread's
        // four columns lie in one head's block
        int hdst[(VEC + 1) / 2] = {};
        bool hquad = false;
        const float hs = a.hout ? a.hscale[0] : 1.f;
        if (a.hout) {
#pragma unroll
            for (int t = 0; t < VEC; t += 2) {
                const int cc = c + t, hd = cc / a.hD;
                hdst[t / 2] = hd * a.hDP + (cc - hd * a.hD);
            }
            if constexpr (VEC == 4) hquad = nv == 4 && c / a.hD == (c + 3) / a.hD;
        }
        for (int64_t r0 = (int64_t)blockIdx.y * kTY + ty; r0 < a.n; r0 += step * UR) {
            float v[UR][VEC], g[UR][VEC];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int64_t r = r0 + u * step;
                if (r < a.n) {
                    load_cols<VEC>(v[u], a.x + r * a.ldx + c, a.wx, nv);
                    load_cols<VEC>(g[u], a.dy + r * a.lddy + c, a.wdy, nv);
                }
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int64_t r = r0 + u * step;
                if (r >= a.n) break;
                float f[VEC];
                if (a.p > 0.f) drop_factors<VEC>(seed, r, c, nquad, a.p, scale, f);
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    const float xh = (v[u][t] - mu[t]) * is[t];
                    float gg = g[u][t];
                    if (a.p > 0.f) gg *= f[t];
                    if (a.relu && !(fmaf(xh, sc[t], sh[t]) > 0.f)) gg = 0.f;
                    v[u][t] = sc[t] * is[t] * (gg - mg[t] - xh * mgx[t]);
                    if (t < nv) amax = fmaxf(amax, fabsf(v[u][t]));
                }
                if (a.dx) store_cols<VEC>(a.dx + r * a.lddx + c, v[u], a.wdx, nv);
                if (a.hout) {            // dx as a halves operand, head blocks of hD columns every hDP (hD even: a column pair never straddles)
                    __half* ho = a.hout + r * a.ldh;
                    __half h1[VEC], h2[VEC];
#pragma unroll
                    for (int t = 0; t < VEC; ++t) {
                        const float z = (t < nv ? v[u][t] : 0.f) * hs;
                        h1[t] = __float2half_rn(z);
                        h2[t] = __float2half_rn((z - __half2float(h1[t])) * kHalvesShift);
                    }
                    if constexpr (VEC == 4) {
                        if (hquad) {         // the four columns are one head's: 8-byte stores (4-byte aligned)
                            *reinterpret_cast<uint2*>(ho + hdst[0]) = *reinterpret_cast<const uint2*>(h1);
                            *reinterpret_cast<uint2*>(ho + hdst[0] + a.h2off) = *reinterpret_cast<const uint2*>(h2);
                            continue;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < VEC; t += 2) {
                        if (t >= nv) break;
                        if (VEC >= 2 && t + 1 < nv) {
                            *reinterpret_cast<__half2*>(ho + hdst[t / 2]) = __halves2half2(h1[t], h1[t + 1]);
                            *reinterpret_cast<__half2*>(ho + hdst[t / 2] + a.h2off) = __halves2half2(h2[t], h2[t + 1]);
                        } else {
                            ho[hdst[t / 2]] = h1[t], ho[hdst[t / 2] + a.h2off] = h2[t];
                        }
                    }
                }
            }
        }
    }
    if (a.absmax) absmax_publish(wave_absmax(amax), a.absmax);
}

// Launch width of the BatchNorm kernels.  16 / 8 / 4-byte lanes when EVERY operand allows them (pick_vec); otherwise, for even
// F with 8-byte aligned operands, the quad form: VEC = 4 lanes whose operands are moved per `wide` flag (load_cols).
static inline bool rows16(const void* p, int64_t ld) { return p != nullptr && aligned(p, 16) && ld % 4 == 0; }
static int bn_vec(int32_t F, std::initializer_list<int64_t> strides, std::initializer_list<const void*> ptrs, bool* quad) {
    const int v = pick_vec(F, strides, ptrs);
    *quad = false;
    if (v == 4 || F % 2 != 0) return v;
    for (int64_t st : strides)
        if (st % 2 != 0) return v;
    for (const void* p : ptrs)
        if (p != nullptr && !aligned(p, 8)) return v;
    *quad = true;
    return 4;
}

static dim3 bn_grid(int32_t F, int vec, int64_t n) {
    const int64_t rb = (n + kTY - 1) / kTY;
    return dim3((unsigned)((F + kTX * vec - 1) / (kTX * vec)), (unsigned)(rb < kRowBlocks ? (rb > 0 ? rb : 1) : kRowBlocks));
}

}  // namespace bot

extern "C" {

int64_t bot_bn_workspace_floats(int32_t F) { return (int64_t)bot::kRowBlocks * 4 * F + F; }   // sums, extremes, bounds (bn_stats_halves)

int bot_colstats_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float* mean, float* m2, float* workspace,
                     bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && F >= 1 && ldx >= F, BOT_E_RANGE, "colstats: n=%lld F=%d ldx=%lld", (long long)n, F, (long long)ldx);
    BOT_REQUIRE(x && mean && m2 && workspace, BOT_E_NULL, "colstats: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    bool quad;
    const int vec = bn_vec(F, {ldx}, {x}, &quad);
    const bool wx = !quad || rows16(x, ldx);
    const dim3 grid = bn_grid(F, vec, n);
    if (vec == 4) hipLaunchKernelGGL((colstats_partial_kernel<4>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, wx, true);
    else if (vec == 2) hipLaunchKernelGGL((colstats_partial_kernel<2>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, true);
    else hipLaunchKernelGGL((colstats_partial_kernel<1>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, true);
    hipLaunchKernelGGL(colstats_final_kernel, dim3((F + 63) / 64), dim3(kBlock), 0, st, x, n, F, workspace, (int)grid.y,
                       mean, m2, (float*)nullptr, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (int64_t*)nullptr);
    return hip_status("colstats launch");
}

int bot_colsum_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float* sum, float* workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && F >= 1 && ldx >= F, BOT_E_RANGE, "colsum: n=%lld F=%d ldx=%lld", (long long)n, F, (long long)ldx);
    BOT_REQUIRE(x && sum && workspace, BOT_E_NULL, "colsum: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    bool quad;
    const int vec = bn_vec(F, {ldx}, {x}, &quad);
    const bool wx = !quad || rows16(x, ldx);
    const dim3 grid = bn_grid(F, vec, n);
    if (vec == 4) hipLaunchKernelGGL((colstats_partial_kernel<4>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, wx, false);
    else if (vec == 2) hipLaunchKernelGGL((colstats_partial_kernel<2>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, false);
    else hipLaunchKernelGGL((colstats_partial_kernel<1>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, false);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((F + 63) / 64), dim3(kBlock), 0, st, workspace, F, (int)grid.y, sum);
    return hip_status("colsum launch");
}

int bot_bn_stats_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float eps, float momentum, float* mean, float* invstd,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float* workspace,
                     bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && F >= 1 && ldx >= F, BOT_E_RANGE, "bn_stats: n=%lld F=%d ldx=%lld", (long long)n, F, (long long)ldx);
    BOT_REQUIRE(x && mean && invstd && workspace, BOT_E_NULL, "bn_stats: NULL pointer");
    BOT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), BOT_E_NULL, "bn_stats: running_mean and running_var go together");
    BOT_REQUIRE(eps >= 0.f && momentum >= 0.f && momentum <= 1.f, BOT_E_RANGE, "bn_stats: eps=%f momentum=%f", (double)eps, (double)momentum);
    hipStream_t st = (hipStream_t)stream;
    bool quad;
    const int vec = bn_vec(F, {ldx}, {x}, &quad);
    const bool wx = !quad || rows16(x, ldx);
    const dim3 grid = bn_grid(F, vec, n);
    if (vec == 4) hipLaunchKernelGGL((colstats_partial_kernel<4>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, wx, true);
    else if (vec == 2) hipLaunchKernelGGL((colstats_partial_kernel<2>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, true);
    else hipLaunchKernelGGL((colstats_partial_kernel<1>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, true);
    hipLaunchKernelGGL(colstats_final_kernel, dim3((F + 63) / 64), dim3(kBlock), 0, st, x, n, F, workspace, (int)grid.y,
                       mean, (float*)nullptr, invstd, eps, momentum, running_mean, running_var, num_batches_tracked);
    return hip_status("bn_stats launch");
}

int bot_bn_stats_halves_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float eps, float momentum, float* mean, float* invstd,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* weight,
                            const float* bias, float p, float* hscale, float* workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && F >= 1 && ldx >= F, BOT_E_RANGE, "bn_stats_halves: n=%lld F=%d ldx=%lld", (long long)n, F, (long long)ldx);
    BOT_REQUIRE(x && mean && invstd && workspace && hscale, BOT_E_NULL, "bn_stats_halves: NULL pointer");
    BOT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), BOT_E_NULL, "bn_stats_halves: running_mean and running_var go together");
    BOT_REQUIRE(eps >= 0.f && momentum >= 0.f && momentum <= 1.f && p >= 0.f && p < 1.f, BOT_E_RANGE, "bn_stats_halves: eps=%f momentum=%f p=%f",
                (double)eps, (double)momentum, (double)p);
    hipStream_t st = (hipStream_t)stream;
    bool quad;
    const int vec = bn_vec(F, {ldx}, {x}, &quad);
    const bool wx = !quad || rows16(x, ldx);
    const dim3 grid = bn_grid(F, vec, n);
    float* minmax = workspace + (int64_t)kRowBlocks * 2 * F;     // [row block][min, max][F]
    float* bound = workspace + (int64_t)kRowBlocks * 4 * F;      // [F]
    if (vec == 4) hipLaunchKernelGGL((colstats_partial_kernel<4>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, wx, true, minmax);
    else if (vec == 2) hipLaunchKernelGGL((colstats_partial_kernel<2>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, true, minmax);
    else hipLaunchKernelGGL((colstats_partial_kernel<1>), grid, dim3(kTX * kTY), 0, st, x, ldx, n, F, workspace, true, true, minmax);
    hipLaunchKernelGGL(colstats_final_kernel, dim3((F + 63) / 64), dim3(kBlock), 0, st, x, n, F, workspace, (int)grid.y,
                       mean, (float*)nullptr, invstd, eps, momentum, running_mean, running_var, num_batches_tracked, minmax, weight, bias, p, bound);
    launch_halves_scale(bound, F, hscale, st);
    return hip_status("bn_stats_halves launch");
}

// v17: bot_bn_stats_halves_f32 from column partials a PRODUCER of x delivered with it (the grouped NT GEMM's epilogue:
// bot_gemm_halves3_nt_grouped2_f32 `stats_*`) instead of a pass over x:  part [nblk][2][F] = per row block the sums of (x - pivot) and
// (x - pivot)^2, minmax [nblk][2][F] the column extremes, pivot [nblk][F] the shift the producer used in each 256-row block (v19: the block's
// own first value, written by the producer; colstats_tiles_final_kernel re-bases the blocks exactly).  nblk = ceil(n / 256).
int bot_bn_stats_halves_partials_f32(const float* part, const float* minmax, int32_t nblk, const float* pivot, int64_t n, int32_t F, float eps, float momentum,
                                     float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                     const float* weight, const float* bias, float p, float* hscale, float* bound_workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && F >= 1 && nblk >= 1, BOT_E_RANGE, "bn_stats_halves_partials: n=%lld F=%d nblk=%d", (long long)n, F, nblk);
    BOT_REQUIRE(part && minmax && pivot && mean && invstd && hscale && bound_workspace, BOT_E_NULL, "bn_stats_halves_partials: NULL pointer");
    BOT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), BOT_E_NULL, "bn_stats_halves_partials: running_mean and running_var go together");
    BOT_REQUIRE(eps >= 0.f && momentum >= 0.f && momentum <= 1.f && p >= 0.f && p < 1.f, BOT_E_RANGE, "bn_stats_halves_partials: eps=%f momentum=%f p=%f",
                (double)eps, (double)momentum, (double)p);
    hipStream_t st = (hipStream_t)stream;
    BOT_REQUIRE((int64_t)nblk == (n + 255) / 256, BOT_E_RANGE, "bn_stats_halves_partials: nblk=%d is not ceil(n / 256) for n=%lld", nblk, (long long)n);
    hipLaunchKernelGGL(colstats_tiles_final_kernel, dim3((F + kTileCols - 1) / kTileCols), dim3(kTileCols * kTileGroups), 0, st, pivot, n, F, part, (int)nblk, 256, mean, invstd, eps, momentum,
                       running_mean, running_var, num_batches_tracked, minmax, weight, bias, p, bound_workspace);
    launch_halves_scale(bound_workspace, F, hscale, st);
    return hip_status("bn_stats_halves_partials launch");
}

static int bn_act_fwd_impl(const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                           const float* weight, const float* bias, int32_t relu, float p, uint64_t seed,
                           const uint64_t* seed_offset, float* y, int64_t ldy, const float* hscale, uint16_t* hout, int64_t ldh,
                           int32_t piece, int32_t pieces, bot_stream_t stream) {
    using namespace bot;
    if (!y) ldy = ldx;                   // halves only (hout != NULL): y puts no constraint on the launch width
    BOT_REQUIRE(n >= 0 && F >= 1 && ldx >= F && ldy >= F, BOT_E_RANGE, "bn_act_fwd: n=%lld F=%d", (long long)n, F);
    BOT_REQUIRE(p >= 0.f && p < 1.f, BOT_E_RANGE, "bn_act_fwd: dropout p=%f must be in [0,1)", (double)p);
    if (n == 0) return 0;
    BOT_REQUIRE(x && mean && invstd && (y || hout), BOT_E_NULL, "bn_act_fwd: NULL pointer");
    BnArgs a{};
    a.x = x, a.ldx = ldx, a.n = n, a.F = F, a.mean = mean, a.invstd = invstd, a.w = weight, a.b = bias, a.relu = relu, a.p = p;
    a.seed = seed, a.seed_offset = seed_offset, a.y = y, a.ldy = ldy;
    a.hout = reinterpret_cast<__half*>(hout), a.ldh = ldh, a.piece = piece, a.hscale = hscale, a.hpieces = pieces;
    bool quad;
    const int vec = bn_vec(F, {ldx, ldy}, {x, y}, &quad);
    if (hout) {
        BOT_REQUIRE(hscale, BOT_E_NULL, "bn_act_fwd_halves: NULL scale");
        BOT_REQUIRE(vec == 4 && piece >= F && piece % 4 == 0 && piece <= (int64_t)((F + kTX * 4 - 1) / (kTX * 4)) * kTX * 4 && (pieces == 2 || pieces == 3) &&
                        ldh >= pieces * (int64_t)piece &&
                        ldh % 4 == 0 && aligned(hout, 8),
                    BOT_E_ALIGN, "bn_act_fwd_halves: needs the 4-column form (even F, 8-byte aligned rows) and piece = F rounded up to x64 "
                                 "(F=%d piece=%d ldh=%lld)", F, piece, (long long)ldh);
    }
    a.wx = !quad || rows16(x, ldx), a.wy = !quad || rows16(y, ldy);
    const dim3 grid = bn_grid(F, vec, n);
    hipStream_t st = (hipStream_t)stream;
    if (hout && piece <= 1024) {         // (vec == 4 was required above) the row-segment form: see bn_act_fwd_rowseg_kernel
        hipLaunchKernelGGL(bn_act_fwd_rowseg_kernel, dim3((unsigned)((n + kRowSeg - 1) / kRowSeg)), dim3((piece / 4 + 63) / 64 * 64), 0, st, a);
        return hip_status("bn_act_fwd rowseg launch");
    }
    if (vec == 4) hipLaunchKernelGGL((bn_act_fwd_kernel<4>), grid, dim3(kTX * kTY), 0, st, a);
    else if (vec == 2) hipLaunchKernelGGL((bn_act_fwd_kernel<2>), grid, dim3(kTX * kTY), 0, st, a);
    else hipLaunchKernelGGL((bn_act_fwd_kernel<1>), grid, dim3(kTX * kTY), 0, st, a);
    return hip_status("bn_act_fwd launch");
}

int bot_bn_act_fwd_f32(const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                       const float* weight, const float* bias, int32_t relu, float p, uint64_t seed,
                       const uint64_t* seed_offset, float* y, int64_t ldy, bot_stream_t stream) {
    return bn_act_fwd_impl(x, ldx, n, F, mean, invstd, weight, bias, relu, p, seed, seed_offset, y, ldy, nullptr, nullptr, 0, 0, 3, stream);
}

int bot_bn_act_fwd_halves_f32(const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                              const float* weight, const float* bias, int32_t relu, float p, uint64_t seed,
                              const uint64_t* seed_offset, float* y, int64_t ldy, const float* hscale, uint16_t* hout, int64_t ldh,
                              int32_t piece, int32_t pieces, bot_stream_t stream) {
    BOT_REQUIRE(hout, BOT_E_NULL, "bn_act_fwd_halves: NULL output");
    return bn_act_fwd_impl(x, ldx, n, F, mean, invstd, weight, bias, relu, p, seed, seed_offset, y, ldy, hscale, hout, ldh, piece, pieces, stream);
}

static int bn_act_bwd_reduce_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                                  const float* weight, const float* bias, int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, float* sum_g,
                                  float* sum_gx, float* workspace, bool want_max, bot_stream_t stream);

int bot_bn_act_bwd_reduce_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F,
                              const float* mean, const float* invstd, const float* weight, const float* bias, int32_t relu,
                              float p, uint64_t seed, const uint64_t* seed_offset, float* sum_g, float* sum_gx, float* workspace,
                              bot_stream_t stream) {
    return bn_act_bwd_reduce_impl(dy, lddy, x, ldx, n, F, mean, invstd, weight, bias, relu, p, seed, seed_offset, sum_g, sum_gx, workspace, false, stream);
}

int bot_bn_act_bwd_reduce_max_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F,
                                  const float* mean, const float* invstd, const float* weight, const float* bias, int32_t relu,
                                  float p, uint64_t seed, const uint64_t* seed_offset, float* sum_g, float* sum_gx, float* workspace,
                                  bot_stream_t stream) {
    return bn_act_bwd_reduce_impl(dy, lddy, x, ldx, n, F, mean, invstd, weight, bias, relu, p, seed, seed_offset, sum_g, sum_gx, workspace, true, stream);
}

int bot_bn_bwd_bound_f32(int32_t F, int64_t n, const float* workspace, const float* sum_g, const float* sum_gx, double total_count, const float* weight,
                         const float* invstd, uint32_t* absmax_slots, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(F >= 1 && n >= 1 && workspace && invstd && absmax_slots, BOT_E_NULL, "bn_bwd_bound: F=%d n=%lld or a NULL pointer", F, (long long)n);
    BOT_REQUIRE((sum_g == nullptr) == (sum_gx == nullptr) && (sum_g == nullptr || total_count >= 1.0), BOT_E_RANGE, "bn_bwd_bound: sums / total_count");
    const int nblk = (int)bn_grid(F, 1, n).y;
    static_assert(kBlock == 256, "bn_bwd_bound_kernel: 64 columns x 4 groups");
    hipLaunchKernelGGL(bn_bwd_bound_kernel, dim3((F + 63) / 64), dim3(kBlock), 0, (hipStream_t)stream, F, workspace + (int64_t)kRowBlocks * 2 * F, nblk,
                       sum_g, sum_gx, sum_g ? (float)(1.0 / total_count) : 0.f, weight, invstd, absmax_slots);
    return hip_status("bn_bwd_bound launch");
}

// v18: the two consumers of the reduce pass's partials, for partials a PRODUCER of dy delivered with it (bot_gemm_halves3_nt3_f32 `bn`):
// part / pmax [nblk][2][F] as bn_act_bwd_reduce_kernel leaves them, one row block per 256-row GEMM tile
int bot_bn_act_bwd_reduce_partials_f32(const float* part, int32_t nblk, int32_t F, float* sum_g, float* sum_gx, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(part && sum_g && sum_gx, BOT_E_NULL, "bn_act_bwd_reduce_partials: NULL pointer");
    BOT_REQUIRE(nblk >= 1 && F >= 1, BOT_E_RANGE, "bn_act_bwd_reduce_partials: nblk=%d F=%d", nblk, F);
    hipLaunchKernelGGL(pair_final_wide_kernel, dim3((F + kWideCols - 1) / kWideCols), dim3(kWideCols * kWideGroups), 0, (hipStream_t)stream, F, part, (int)nblk, sum_g, sum_gx);
    return hip_status("bn_act_bwd_reduce_partials launch");
}

// both second stages in one launch (no cross-rank reduction between them: the local sums are the final ones)
int bot_bn_bwd_partials_finish_f32(const float* part, const float* pmax, int32_t nblk, int32_t F, float* sum_g, float* sum_gx, int32_t batch_stats,
                                   double total_count, const float* weight, const float* invstd, uint32_t* absmax_slots, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(part && pmax && sum_g && sum_gx && invstd && absmax_slots, BOT_E_NULL, "bn_bwd_partials_finish: NULL pointer");
    BOT_REQUIRE(nblk >= 1 && F >= 1 && (!batch_stats || total_count >= 1.0), BOT_E_RANGE, "bn_bwd_partials_finish: nblk=%d F=%d total_count=%f", nblk, F, total_count);
    hipLaunchKernelGGL(bn_bwd_finish_wide_kernel, dim3((F + kWideCols - 1) / kWideCols), dim3(kWideCols * kWideGroups), 0, (hipStream_t)stream, F, part, pmax,
                       (int)nblk, sum_g, sum_gx, batch_stats ? (float)(1.0 / total_count) : 0.f, batch_stats != 0, weight, invstd, absmax_slots);
    return hip_status("bn_bwd_partials_finish launch");
}

int bot_bn_bwd_bound_partials_f32(int32_t F, const float* pmax, int32_t nblk, const float* sum_g, const float* sum_gx, double total_count, const float* weight,
                                  const float* invstd, uint32_t* absmax_slots, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(F >= 1 && nblk >= 1 && pmax && invstd && absmax_slots, BOT_E_NULL, "bn_bwd_bound_partials: F=%d nblk=%d or a NULL pointer", F, nblk);
    BOT_REQUIRE((sum_g == nullptr) == (sum_gx == nullptr) && (sum_g == nullptr || total_count >= 1.0), BOT_E_RANGE, "bn_bwd_bound_partials: sums / total_count");
    hipLaunchKernelGGL(bn_bwd_bound_wide_kernel, dim3((F + kWideCols - 1) / kWideCols), dim3(kWideCols * kWideGroups), 0, (hipStream_t)stream, F, pmax, (int)nblk, sum_g, sum_gx,
                       sum_g ? (float)(1.0 / total_count) : 0.f, weight, invstd, absmax_slots);
    return hip_status("bn_bwd_bound_partials launch");
}

static int bn_act_bwd_reduce_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                                  const float* weight, const float* bias, int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, float* sum_g,
                                  float* sum_gx, float* workspace, bool want_max, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && F >= 1 && ldx >= F && lddy >= F, BOT_E_RANGE, "bn_act_bwd_reduce: n=%lld F=%d", (long long)n, F);
    BOT_REQUIRE(dy && x && mean && invstd && sum_g && sum_gx && workspace, BOT_E_NULL, "bn_act_bwd_reduce: NULL pointer");
    BnArgs a{};
    a.x = x, a.ldx = ldx, a.n = n, a.F = F, a.mean = mean, a.invstd = invstd, a.w = weight, a.b = bias, a.relu = relu, a.p = p;
    a.seed = seed, a.seed_offset = seed_offset, a.dy = dy, a.lddy = lddy, a.part = workspace;
    a.pmax = want_max ? workspace + (int64_t)kRowBlocks * 2 * F : nullptr;        // (bot_bn_workspace_floats: 4 kRowBlocks F + F)
    bool quad;
    const int vec = bn_vec(F, {ldx, lddy}, {x, dy}, &quad);
    a.wx = !quad || rows16(x, ldx), a.wdy = !quad || rows16(dy, lddy);
    const dim3 grid = bn_grid(F, vec, n);
    hipStream_t st = (hipStream_t)stream;
    if (vec == 4) hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<4>), grid, dim3(kTX * kTY), 0, st, a);
    else if (vec == 2) hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<2>), grid, dim3(kTX * kTY), 0, st, a);
    else hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<1>), grid, dim3(kTX * kTY), 0, st, a);
    hipLaunchKernelGGL(pair_final_kernel, dim3((F + 63) / 64), dim3(kBlock), 0, st, F, workspace, (int)grid.y, sum_g,
                       sum_gx);
    return hip_status("bn_act_bwd_reduce launch");
}

static int bn_act_bwd_apply_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                                 const float* weight, const float* bias, int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, const float* sum_g,
                                 const float* sum_gx, double total_count, float* dx, int64_t lddx, uint32_t* absmax_slots, const float* hscale, uint16_t* hout,
                                 int64_t ldh, int32_t h2_off, int32_t hD, int32_t hDP, bot_stream_t stream);

int bot_bn_act_bwd_apply_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F,
                             const float* mean, const float* invstd, const float* weight, const float* bias, int32_t relu,
                             float p, uint64_t seed, const uint64_t* seed_offset, const float* sum_g, const float* sum_gx,
                             double total_count, float* dx, int64_t lddx, uint32_t* absmax_slots, bot_stream_t stream) {
    BOT_REQUIRE(dx || n == 0, BOT_E_NULL, "bn_act_bwd_apply: NULL pointer");
    return bn_act_bwd_apply_impl(dy, lddy, x, ldx, n, F, mean, invstd, weight, bias, relu, p, seed, seed_offset, sum_g, sum_gx, total_count, dx, lddx,
                                 absmax_slots, nullptr, nullptr, 0, 0, 0, 0, stream);
}

int bot_bn_act_bwd_apply_halves_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                                    const float* weight, const float* bias, int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset,
                                    const float* sum_g, const float* sum_gx, double total_count, float* dx, int64_t lddx, const float* hscale, uint16_t* hout,
                                    int64_t ldh, int32_t h2_off, int32_t hD, int32_t hDP, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(hscale && hout, BOT_E_NULL, "bn_act_bwd_apply_halves: NULL pointer");
    BOT_REQUIRE(hD >= 2 && hD % 2 == 0 && hDP >= hD && hDP % 2 == 0 && F % hD == 0 && h2_off % 2 == 0 && ldh % 2 == 0 && h2_off >= (int64_t)(F / hD) * hDP &&
                    ldh >= h2_off + (int64_t)(F / hD) * hDP && aligned(hout, 4), BOT_E_RANGE, "bn_act_bwd_apply_halves: F=%d hD=%d hDP=%d h2_off=%d ldh=%lld", F, hD, hDP,
                h2_off, (long long)ldh);
    return bn_act_bwd_apply_impl(dy, lddy, x, ldx, n, F, mean, invstd, weight, bias, relu, p, seed, seed_offset, sum_g, sum_gx, total_count, dx, lddx, nullptr,
                                 hscale, hout, ldh, h2_off, hD, hDP, stream);
}

static int bn_act_bwd_apply_impl(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                                 const float* weight, const float* bias, int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, const float* sum_g,
                                 const float* sum_gx, double total_count, float* dx, int64_t lddx, uint32_t* absmax_slots, const float* hscale, uint16_t* hout,
                                 int64_t ldh, int32_t h2_off, int32_t hD, int32_t hDP, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && F >= 1 && ldx >= F && lddy >= F && (dx == nullptr || lddx >= F), BOT_E_RANGE, "bn_act_bwd_apply: n=%lld F=%d", (long long)n, F);
    if (n == 0) return 0;
    BOT_REQUIRE(dy && x && mean && invstd && (dx || hout), BOT_E_NULL, "bn_act_bwd_apply: NULL pointer");
    BOT_REQUIRE((sum_g == nullptr) == (sum_gx == nullptr), BOT_E_NULL, "bn_act_bwd_apply: sum_g and sum_gx go together");
    BOT_REQUIRE(sum_g == nullptr || total_count >= 1.0, BOT_E_RANGE, "bn_act_bwd_apply: total_count=%f", total_count);
    BnArgs a{};
    a.x = x, a.ldx = ldx, a.n = n, a.F = F, a.mean = mean, a.invstd = invstd, a.w = weight, a.b = bias, a.relu = relu, a.p = p;
    a.seed = seed, a.seed_offset = seed_offset, a.dy = dy, a.lddy = lddy, a.sum_g = sum_g, a.sum_gx = sum_gx, a.inv_count = sum_g ? (float)(1.0 / total_count) : 0.f;
    a.dx = dx, a.lddx = lddx, a.absmax = absmax_slots;
    a.hout = reinterpret_cast<__half*>(hout), a.hscale = hscale, a.ldh = ldh, a.h2off = h2_off, a.hD = hD, a.hDP = hDP;
    if (!dx) lddx = ldx;        // (no fp32 output: its pitch must not narrow the launch width)
    bool quad;
    const int vec = bn_vec(F, {ldx, lddy, lddx}, {x, dy, dx}, &quad);
    a.wx = !quad || rows16(x, ldx), a.wdy = !quad || rows16(dy, lddy), a.wdx = !quad || rows16(dx, lddx);
    const dim3 grid = bn_grid(F, vec, n);       // (round 6: 384 ... 1024 row blocks instead of 256 change nothing, alone or beside the side stream)
    // (round 6, profiles/r06_apply_beside_tn.txt: beside the side stream's weight-gradient product this pass takes as long as that product -
    // 0.40 -> 0.95 ms - and the PAIR ends 40-90 us after the product alone would: s_setprio 3 for this pass's waves gave 0.74 ms here and
    // +60 us on the product, and nothing in the step, 10.52 vs 10.54 ms: the time moves to the kernels behind it)
    hipStream_t st = (hipStream_t)stream;
    if (vec == 4) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<4>), grid, dim3(kTX * kTY), 0, st, a);
    else if (vec == 2) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<2>), grid, dim3(kTX * kTY), 0, st, a);
    else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<1>), grid, dim3(kTX * kTY), 0, st, a);
    return hip_status("bn_act_bwd_apply launch");
}

}  // extern "C"
