// Edge-sized (nnz*H) kernels for gfx950: attention logits + leaky-ReLU + per-destination softmax
// (forward and backward) and the segment sum of edge values.
//
// These sweep 4*H bytes per edge instead of 4*H*D, so they are two orders of magnitude lighter
// than the SpMM/SDDMM gathers; the node-side operands el/er ([n,H], 2 MB at ogbn-arxiv) stay
// resident in the 4 MiB XCD L2.  Lanes run ACROSS THE EDGES of a row: a 16-lane group per short
// row (4 rows per wavefront; mean in-degree of the target graphs is ~15), one 256-thread workgroup
// per long row (rows above the row plan's chunk), both through the same per-row routine and in ONE launch.  Softmax is
// the online (running max / running sum) form: two passes over the row, not three; heads are held
// in registers HT at a time.  No atomics; fixed reduction order.
//
// HBM roofline: forward 4*[2*nnz*H + nnz + (n+1) + n*H*k] bytes, backward 4*[3*nnz*H + nnz + (n+1)].
#include "common.h"

namespace bot {

// (round 6 tried a whole wavefront per short row where the mean degree is 64 or more - S-proteins: attention forward 3.70 -> 3.53 ms, backward
// 1.99 -> 1.87, step 287.6 -> 285.5 ms - and withdrew it: the other summation order moved config 4's badly conditioned gradients from
// 53 / 10 / 1 (within 1e-4 of fp64 / only through the 2x clause / farther than the fp32 oracle) to 52 / 12 / 0, over the count the
// full-size test bounds; 0.7 % of one config does not buy a looser bound)
// (round 6 also grouped the row loop's optional index loads into branch-free straight-line code - five dependent memory round trips per edge
// became three - and measured nothing: S-products 3.35 -> 3.42 ms, S-proteins 3.70 -> 3.71 ms.  These launches are bound by the random
// 128-byte lines their el[src] gathers pull through the fabric (S-products: 126 M lines = 16 GB per forward), not by latency; reverted.)
constexpr int kRowLanes = 16;

template <int LANES>
struct GroupCtx {
    static constexpr int kStride = LANES;
    int lane;
    __device__ __forceinline__ float sum(float v) const { return group_sum<LANES>(v); }
    __device__ __forceinline__ float max(float v) const { return group_max<LANES>(v); }
    template <int N>
    __device__ __forceinline__ void sum_n(float (&v)[N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = group_sum<LANES>(v[j]);
    }
    template <int N>
    __device__ __forceinline__ void max_n(float (&v)[N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = group_max<LANES>(v[j]);
    }
};

struct BlockCtx {
    static constexpr int kStride = kBlock;
    int lane;    // thread id in the workgroup
    float* lds;  // (kBlock / 64) * 8 floats
    // all N <= 8 values of a row reduced with ONE pair of barriers (one per value cost 12 barriers a row at H = 6)
    template <int N, bool MAX>
    __device__ __forceinline__ void reduce_n(float (&v)[N]) const {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = MAX ? group_max<64>(v[j]) : group_sum<64>(v[j]);
        __syncthreads();
        if ((lane & 63) == 0) {
#pragma unroll
            for (int j = 0; j < N; ++j) lds[(lane >> 6) * 8 + j] = v[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float s = lds[j];
#pragma unroll
            for (int i = 1; i < kBlock / 64; ++i) s = MAX ? fmaxf(s, lds[i * 8 + j]) : s + lds[i * 8 + j];
            v[j] = s;
        }
    }
    template <int N>
    __device__ __forceinline__ void sum_n(float (&v)[N]) const { reduce_n<N, false>(v); }
    template <int N>
    __device__ __forceinline__ void max_n(float (&v)[N]) const { reduce_n<N, true>(v); }
    __device__ __forceinline__ float sum(float v) const {
        v = group_sum<64>(v);
        __syncthreads();
        if ((lane & 63) == 0) lds[lane >> 6] = v;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kBlock / 64; ++i) s += lds[i];
        return s;
    }
    __device__ __forceinline__ float max(float v) const {
        v = group_max<64>(v);
        __syncthreads();
        if ((lane & 63) == 0) lds[lane >> 6] = v;
        __syncthreads();
        float s = lds[0];
#pragma unroll
        for (int i = 1; i < kBlock / 64; ++i) s = fmaxf(s, lds[i]);
        return s;
    }
};

struct AttnArgs {
    const int32_t* indptr;
    const int32_t* indices;
    int64_t n_rows;
    const int32_t* long_rows;
    int32_t chunk;
    const float* el;
    const float* er;
    const float* ee;
    const int32_t* eperm;
    const uint8_t* keep;
    float slope;
    int32_t H, h0;
    // forward: a (out).  backward: a, da (in), dz, der (out)
    float* a;
    const float* da;
    const int32_t* aperm;
    float* dz;
    const int32_t* zperm;
    float* der;
    bool wide;  // el / ee / a / da / dz records may be moved as 8- / 16-byte vectors (see load_heads)
    // optional (H <= 8): bit j of zsign[row of a] = [z_j > 0], written by the forward; with it the backward needs neither the
    // el[src] gather nor ee nor the edge ids for the leaky-ReLU derivative (1 byte per edge instead of ~4*(2H+1))
    uint8_t* zsign;
    // optional dropout on the attention weights (nn.Dropout(attn_drop) behind edge_softmax, models.py:544): the forward also
    // writes a_drop = a * keep / (1 - p); the backward takes d(a_drop) in `da` and applies the same factor first.  The keep
    // mask is a Philox4x32-10 stream keyed by (seed [+ device word], record index, head / 4): regenerated, never stored.
    float drop_p;
    uint64_t drop_seed;
    const uint64_t* seed_offset;
    float* a_drop;
};

// keep/(1-p) factors of the HT heads [h0, h0+HT) of the edge record `arow`
template <int HT>
__device__ __forceinline__ void attn_drop_factors(const AttnArgs& p, uint64_t seed, int64_t arow, float (&f)[HT]) {
    const float scale = 1.f / (1.f - p.drop_p);
    const int64_t nq = (p.H + 3) / 4;
#pragma unroll
    for (int q = 0; q < (HT + 3) / 4; ++q) {
        uint32_t w[4];
        Philox::gen(seed, (uint64_t)(arow * nq + (p.h0 >> 2) + q), w);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (q * 4 + t < HT) f[q * 4 + t] = ((w[t] >> 8) * (1.0f / 16777216.0f)) >= p.drop_p ? scale : 0.f;
    }
}

// HT consecutive floats at `q`.  `wide`: the record is 4*HT bytes at a multiple of 4*HT bytes from a 16-byte aligned base
// (H == HT, one head tile), so even HT moves as 8- or 16-byte vectors: one memory instruction touches each record's cache
// line once instead of HT times — the edge-sized sweeps are bound by cache-line transactions, not bytes.
template <int HT>
__device__ __forceinline__ void load_heads(const float* q, bool wide, float (&v)[HT]) {
    if constexpr (HT % 4 == 0) {
        if (wide) {
#pragma unroll
            for (int j = 0; j < HT; j += 4) {
                const float4 t = *reinterpret_cast<const float4*>(q + j);
                v[j] = t.x, v[j + 1] = t.y, v[j + 2] = t.z, v[j + 3] = t.w;
            }
            return;
        }
    } else if constexpr (HT % 2 == 0) {
        if (wide) {
#pragma unroll
            for (int j = 0; j < HT; j += 2) {
                const float2 t = *reinterpret_cast<const float2*>(q + j);
                v[j] = t.x, v[j + 1] = t.y;
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < HT; ++j) v[j] = q[j];
}

template <int HT>
__device__ __forceinline__ void store_heads(float* q, bool wide, const float (&v)[HT]) {
    if constexpr (HT % 4 == 0) {
        if (wide) {
#pragma unroll
            for (int j = 0; j < HT; j += 4) *reinterpret_cast<float4*>(q + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
            return;
        }
    } else if constexpr (HT % 2 == 0) {
        if (wide) {
#pragma unroll
            for (int j = 0; j < HT; j += 2) *reinterpret_cast<float2*>(q + j) = make_float2(v[j], v[j + 1]);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < HT; ++j) q[j] = v[j];
}

// z = er[dst] + el[src] + ee[e] for the HT heads of one edge
template <int HT>
__device__ __forceinline__ void edge_z(const AttnArgs& p, int k, int ep, const float (&erv)[HT], float (&z)[HT]) {
#pragma unroll
    for (int j = 0; j < HT; ++j) z[j] = erv[j];
    float t[HT];
    if (p.el) {
        load_heads<HT>(p.el + (int64_t)p.indices[k] * p.H + p.h0, p.wide, t);
#pragma unroll
        for (int j = 0; j < HT; ++j) z[j] += t[j];
    }
    if (p.ee) {
        load_heads<HT>(p.ee + (int64_t)ep * p.H + p.h0, p.wide, t);
#pragma unroll
        for (int j = 0; j < HT; ++j) z[j] += t[j];
    }
}

// Forward, two passes over the row: (1) logits -> `a`, running max / running sum per lane (online softmax);
// (2) rescale `a` in place — a pure stream, nothing is gathered twice.  Dropped edges carry -inf between the passes.
template <int HT, class Ctx>
__device__ __forceinline__ void attn_fwd_row(const AttnArgs& p, const Ctx& ctx, int row, int beg, int end) {
    const float NEG_INF = -__builtin_inff();
    float erv[HT], m[HT], s[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j) {
        erv[j] = p.er ? p.er[(int64_t)row * p.H + p.h0 + j] : 0.f;
        m[j] = NEG_INF;
        s[j] = 0.f;
    }
    for (int k = beg + ctx.lane; k < end; k += Ctx::kStride) {
        const int ep = p.eperm ? p.eperm[k] : k;
        const int64_t arow = p.aperm ? p.aperm[k] : k;
        float* ao = p.a + arow * p.H + p.h0;
        float e[HT];
        if (p.keep && p.keep[ep] == 0) {
#pragma unroll
            for (int j = 0; j < HT; ++j) e[j] = NEG_INF;
            store_heads<HT>(ao, p.wide, e);
            if (p.zsign) p.zsign[arow] = 0;
            continue;
        }
        edge_z<HT>(p, k, ep, erv, e);
        if (p.zsign) {
            unsigned bits = 0;
#pragma unroll
            for (int j = 0; j < HT; ++j) bits |= (e[j] > 0.f ? 1u : 0u) << j;
            p.zsign[arow] = (uint8_t)bits;
        }
#pragma unroll
        for (int j = 0; j < HT; ++j) {
            e[j] = e[j] > 0.f ? e[j] : e[j] * p.slope;
            const float mn = fmaxf(m[j], e[j]);
            s[j] = s[j] * expf(m[j] - mn) + expf(e[j] - mn);  // m = -inf on first use: s = 0 * 0 + 1
            m[j] = mn;
        }
        store_heads<HT>(ao, p.wide, e);
    }
    float M[HT], inv[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j) M[j] = m[j];
    ctx.max_n(M);
#pragma unroll
    for (int j = 0; j < HT; ++j) inv[j] = m[j] > NEG_INF ? s[j] * expf(m[j] - M[j]) : 0.f;
    ctx.sum_n(inv);
#pragma unroll
    for (int j = 0; j < HT; ++j) inv[j] = inv[j] > 0.f ? 1.f / inv[j] : 0.f;
    const uint64_t dseed = p.a_drop ? (p.seed_offset ? p.drop_seed + p.seed_offset[0] * 0x9E3779B97F4A7C15ull : p.drop_seed) : 0;
    for (int k = beg + ctx.lane; k < end; k += Ctx::kStride) {
        const int64_t arow = p.aperm ? p.aperm[k] : k;
        float* ao = p.a + arow * p.H + p.h0;
        float e[HT];
        load_heads<HT>(ao, p.wide, e);
#pragma unroll
        for (int j = 0; j < HT; ++j) e[j] = e[j] > NEG_INF ? expf(e[j] - M[j]) * inv[j] : 0.f;
        store_heads<HT>(ao, p.wide, e);
        if (p.a_drop) {
            float f[HT];
            attn_drop_factors<HT>(p, dseed, arow, f);
#pragma unroll
            for (int j = 0; j < HT; ++j) e[j] *= f[j];
            store_heads<HT>(p.a_drop + arow * p.H + p.h0, p.wide, e);
        }
    }
}

template <int HT, class Ctx>
__device__ __forceinline__ void attn_bwd_row(const AttnArgs& p, const Ctx& ctx, int row, int beg, int end) {
    float erv[HT], t[HT], dacc[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j) {
        erv[j] = p.er ? p.er[(int64_t)row * p.H + p.h0 + j] : 0.f;
        t[j] = 0.f;
        dacc[j] = 0.f;
    }
    const bool dropped = p.drop_p > 0.f;
    const uint64_t dseed = dropped ? (p.seed_offset ? p.drop_seed + p.seed_offset[0] * 0x9E3779B97F4A7C15ull : p.drop_seed) : 0;
    for (int k = beg + ctx.lane; k < end; k += Ctx::kStride) {
        const int64_t arow = p.aperm ? p.aperm[k] : k;
        const int64_t o = arow * p.H + p.h0;
        float av[HT], dv[HT];
        load_heads<HT>(p.a + o, p.wide, av);
        load_heads<HT>(p.da + o, p.wide, dv);
        if (dropped) {
            float f[HT];
            attn_drop_factors<HT>(p, dseed, arow, f);
#pragma unroll
            for (int j = 0; j < HT; ++j) dv[j] *= f[j];
        }
#pragma unroll
        for (int j = 0; j < HT; ++j) t[j] = fmaf(av[j], dv[j], t[j]);
    }
    ctx.sum_n(t);
    const bool need_z = p.slope != 1.f;
    for (int k = beg + ctx.lane; k < end; k += Ctx::kStride) {
        const int64_t arow = p.aperm ? p.aperm[k] : k;
        const int64_t o = arow * p.H + p.h0;
        float av[HT], dv[HT], z[HT], g[HT];
        load_heads<HT>(p.a + o, p.wide, av);
        load_heads<HT>(p.da + o, p.wide, dv);
        if (dropped) {
            float f[HT];
            attn_drop_factors<HT>(p, dseed, arow, f);
#pragma unroll
            for (int j = 0; j < HT; ++j) dv[j] *= f[j];
        }
        if (need_z) {
            if (p.zsign) {
                const unsigned bits = p.zsign[p.aperm ? p.aperm[k] : k];
#pragma unroll
                for (int j = 0; j < HT; ++j) z[j] = (bits >> j) & 1u ? 1.f : 0.f;
            } else {
                edge_z<HT>(p, k, p.eperm ? p.eperm[k] : k, erv, z);
            }
        }
#pragma unroll
        for (int j = 0; j < HT; ++j) {
            g[j] = av[j] * (dv[j] - t[j]);
            if (need_z && !(z[j] > 0.f)) g[j] *= p.slope;
            dacc[j] += g[j];
        }
        store_heads<HT>(p.dz + (int64_t)(p.zperm ? p.zperm[k] : k) * p.H + p.h0, p.wide, g);
    }
    if (p.der) {
        ctx.sum_n(dacc);
        if (ctx.lane == 0) {
#pragma unroll
            for (int j = 0; j < HT; ++j) p.der[(int64_t)row * p.H + p.h0 + j] = dacc[j];
        }
    }
}

// ONE launch for both row classes: the first `n_long` workgroups take one long row each (all 256 threads on the row, they
// start first: they are the longest pieces of work), the others take 16 short rows each (a 16-lane group per row).  As two
// launches the long-row kernel (a few hundred small workgroups, latency-bound by the longest row) ran alone on an empty chip
// behind the short-row kernel: 73 + 47 us per forward at S-arxiv; merged they overlap.
template <int HT, bool BWD>
__global__ __launch_bounds__(kBlock) void attn_kernel(AttnArgs p, int n_long) {
    __shared__ float lds[(kBlock / 64) * 8];
    if ((int)blockIdx.x < n_long) {  // workgroup-uniform
        const int row = p.long_rows[blockIdx.x];
        const int beg = p.indptr[row], end = p.indptr[row + 1];
        BlockCtx ctx{(int)threadIdx.x, lds};
        if constexpr (BWD) attn_bwd_row<HT>(p, ctx, row, beg, end);
        else attn_fwd_row<HT>(p, ctx, row, beg, end);
        return;
    }
    const int64_t row = ((int64_t)(blockIdx.x - n_long) * kBlock + threadIdx.x) / kRowLanes;
    if (row >= p.n_rows) return;
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    if (end - beg > p.chunk) return;  // a long row: owned by one of the first workgroups
    GroupCtx<kRowLanes> ctx{(int)(threadIdx.x % kRowLanes)};
    if constexpr (BWD) attn_bwd_row<HT>(p, ctx, (int)row, beg, end);
    else attn_fwd_row<HT>(p, ctx, (int)row, beg, end);
}

template <bool BWD>
static int launch_attn(AttnArgs p, int64_t n_long, hipStream_t st) {
    const int64_t blocks = (p.n_rows * kRowLanes + kBlock - 1) / kBlock;
    p.wide = p.H <= 8 && aligned(p.el, 16) && aligned(p.ee, 16) && aligned(p.a, 16) && aligned(p.da, 16) && aligned(p.dz, 16) &&
             aligned(p.a_drop, 16);
    for (int h0 = 0; h0 < p.H;) {  // heads in register tiles of up to 8 (one sweep over the edges for H <= 8)
        const int ht = p.H - h0 >= 8 ? 8 : p.H - h0;
        p.h0 = h0;
#define BOT_LAUNCH_ATTN(HT) \
    hipLaunchKernelGGL((attn_kernel<HT, BWD>), dim3((unsigned)(blocks + n_long)), dim3(kBlock), 0, st, p, (int)n_long)
        switch (ht) {
            case 8: BOT_LAUNCH_ATTN(8); break;
            case 7: BOT_LAUNCH_ATTN(7); break;
            case 6: BOT_LAUNCH_ATTN(6); break;
            case 5: BOT_LAUNCH_ATTN(5); break;
            case 4: BOT_LAUNCH_ATTN(4); break;
            case 3: BOT_LAUNCH_ATTN(3); break;
            case 2: BOT_LAUNCH_ATTN(2); break;
            default: BOT_LAUNCH_ATTN(1); break;
        }
#undef BOT_LAUNCH_ATTN
        h0 += ht;
    }
    return hip_status(BWD ? "gat_attn_bwd launch" : "gat_attn_fwd launch");
}

// ---------------------------------------------------------------------------------------------
// segment sum: out[r,:] = sum_k vals[perm[k],:]
// ---------------------------------------------------------------------------------------------
struct SegArgs {
    const int32_t* indptr;
    int64_t n_rows;
    const int32_t* long_rows;
    int32_t chunk;
    const float* vals;
    const int32_t* perm;
    int32_t W, w0;
    float* out;
    bool wide;  // W == WT (one tile) and `vals` is 16-byte aligned: records move as 8- / 16-byte vectors
};

template <int WT, class Ctx>
__device__ __forceinline__ void seg_row(const SegArgs& p, const Ctx& ctx, int row, int beg, int end) {
    float acc[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[j] = 0.f;
    for (int k = beg + ctx.lane; k < end; k += Ctx::kStride) {
        float v[WT];
        load_heads<WT>(p.vals + (int64_t)(p.perm ? p.perm[k] : k) * p.W + p.w0, p.wide, v);
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[j] += v[j];
    }
    ctx.sum_n(acc);
    if (ctx.lane == 0) {
#pragma unroll
        for (int j = 0; j < WT; ++j) p.out[(int64_t)row * p.W + p.w0 + j] = acc[j];
    }
}

template <int WT>
__global__ __launch_bounds__(kBlock) void seg_kernel(SegArgs p, int n_long) {  // long rows first, as in attn_kernel
    __shared__ float lds[(kBlock / 64) * 8];
    if ((int)blockIdx.x < n_long) {
        const int row = p.long_rows[blockIdx.x];
        BlockCtx ctx{(int)threadIdx.x, lds};
        seg_row<WT>(p, ctx, row, p.indptr[row], p.indptr[row + 1]);
        return;
    }
    const int64_t row = ((int64_t)(blockIdx.x - n_long) * kBlock + threadIdx.x) / kRowLanes;
    if (row >= p.n_rows) return;
    const int beg = p.indptr[row], end = p.indptr[row + 1];
    if (end - beg > p.chunk) return;
    GroupCtx<kRowLanes> ctx{(int)(threadIdx.x % kRowLanes)};
    seg_row<WT>(p, ctx, (int)row, beg, end);
}

__global__ __launch_bounds__(kBlock) void degrees_kernel(const int32_t* indptr, int64_t n_rows, int64_t* deg) {
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r < n_rows) deg[r] = (int64_t)indptr[r + 1] - (int64_t)indptr[r];
}

static int attn_check(const char* who, const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                      const int32_t* long_rows, int64_t n_long, int32_t chunk, const float* el, int32_t H) {
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_long >= 0, BOT_E_RANGE, "%s: negative size", who);
    BOT_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, BOT_E_RANGE, "%s: int32 index range exceeded", who);
    BOT_REQUIRE(H >= 1 && chunk >= 1, BOT_E_RANGE, "%s: H=%d chunk=%d", who, H, chunk);
    BOT_REQUIRE(indptr != nullptr, BOT_E_NULL, "%s: indptr is NULL", who);
    BOT_REQUIRE(n_long == 0 || long_rows, BOT_E_NULL, "%s: long_rows is NULL", who);
    BOT_REQUIRE(el == nullptr || indices != nullptr || nnz == 0, BOT_E_NULL, "%s: el given without indices", who);
    return 0;
}

}  // namespace bot

extern "C" {

int bot_gat_attn_fwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                         const int32_t* long_rows, int64_t n_long, int32_t chunk, const float* el, const float* er,
                         const float* ee, const int32_t* eperm, const uint8_t* keep, float slope, int32_t H, float* a,
                         const int32_t* aperm, uint8_t* zsign, float attn_drop, uint64_t drop_seed, const uint64_t* seed_offset,
                         float* a_drop, bot_stream_t stream) {
    using namespace bot;
    if (int rc = attn_check("gat_attn_fwd", indptr, indices, n_rows, nnz, long_rows, n_long, chunk, el, H)) return rc;
    if (n_rows == 0 || nnz == 0) return 0;
    BOT_REQUIRE(a != nullptr, BOT_E_NULL, "gat_attn_fwd: a is NULL");
    BOT_REQUIRE(el || er || ee, BOT_E_NULL, "gat_attn_fwd: no logit source (el, er, ee all NULL)");
    BOT_REQUIRE(zsign == nullptr || H <= 8, BOT_E_RANGE, "gat_attn_fwd: zsign holds 8 heads, H=%d", H);
    BOT_REQUIRE(attn_drop >= 0.f && attn_drop < 1.f, BOT_E_RANGE, "gat_attn_fwd: attn_drop=%f must be in [0,1)", (double)attn_drop);
    BOT_REQUIRE((attn_drop > 0.f) == (a_drop != nullptr), BOT_E_NULL, "gat_attn_fwd: a_drop goes with attn_drop > 0");
    AttnArgs p{indptr, indices, n_rows, long_rows, chunk, el, er, ee, eperm, keep, slope, H, 0, a, nullptr, aperm, nullptr,
               nullptr, nullptr, false, zsign, attn_drop, drop_seed, seed_offset, a_drop};
    return launch_attn<false>(p, n_long, (hipStream_t)stream);
}

int bot_gat_attn_bwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                         const int32_t* long_rows, int64_t n_long, int32_t chunk, const float* el, const float* er,
                         const float* ee, const int32_t* eperm, float slope, int32_t H, const float* a, const float* da,
                         const int32_t* aperm, float* dz, const int32_t* zperm, float* der, const uint8_t* zsign,
                         float attn_drop, uint64_t drop_seed, const uint64_t* seed_offset, bot_stream_t stream) {
    using namespace bot;
    if (int rc = attn_check("gat_attn_bwd", indptr, indices, n_rows, nnz, long_rows, n_long, chunk, el, H)) return rc;
    if (n_rows == 0) return 0;
    BOT_REQUIRE(nnz == 0 || (a && da && dz), BOT_E_NULL, "gat_attn_bwd: a/da/dz is NULL");
    BOT_REQUIRE(slope == 1.f || zsign || el || er || ee, BOT_E_NULL, "gat_attn_bwd: slope != 1 needs zsign or el/er/ee for the sign");
    BOT_REQUIRE(zsign == nullptr || H <= 8, BOT_E_RANGE, "gat_attn_bwd: zsign holds 8 heads, H=%d", H);
    BOT_REQUIRE(attn_drop >= 0.f && attn_drop < 1.f, BOT_E_RANGE, "gat_attn_bwd: attn_drop=%f must be in [0,1)", (double)attn_drop);
    AttnArgs p{indptr, indices, n_rows, long_rows, chunk, el, er, ee, eperm, nullptr, slope, H, 0, const_cast<float*>(a), da,
               aperm, dz, zperm, der, false, const_cast<uint8_t*>(zsign), attn_drop, drop_seed, seed_offset, nullptr};
    return launch_attn<true>(p, n_long, (hipStream_t)stream);
}

int bot_segment_sum_f32(const int32_t* indptr, int64_t n_rows, int64_t nnz, const int32_t* long_rows, int64_t n_long,
                        int32_t chunk, const float* vals, const int32_t* perm, int32_t W, float* out,
                        bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_long >= 0, BOT_E_RANGE, "segment_sum: negative size");
    BOT_REQUIRE(W >= 1 && chunk >= 1, BOT_E_RANGE, "segment_sum: W=%d chunk=%d", W, chunk);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(indptr && out && (nnz == 0 || vals), BOT_E_NULL, "segment_sum: NULL pointer");
    BOT_REQUIRE(n_long == 0 || long_rows, BOT_E_NULL, "segment_sum: long_rows is NULL");
    hipStream_t st = (hipStream_t)stream;
    SegArgs p{indptr, n_rows, long_rows, chunk, vals, perm, W, 0, out, W <= 8 && aligned(vals, 16)};
    const int64_t blocks = (n_rows * kRowLanes + kBlock - 1) / kBlock;
    for (int w0 = 0; w0 < W;) {  // record columns in register tiles of up to 8 (one sweep for W <= 8)
        const int wt = W - w0 >= 8 ? 8 : W - w0;
        p.w0 = w0;
#define BOT_LAUNCH_SEG(WT) hipLaunchKernelGGL((seg_kernel<WT>), dim3((unsigned)(blocks + n_long)), dim3(kBlock), 0, st, p, (int)n_long)
        switch (wt) {
            case 8: BOT_LAUNCH_SEG(8); break;
            case 7: BOT_LAUNCH_SEG(7); break;
            case 6: BOT_LAUNCH_SEG(6); break;
            case 5: BOT_LAUNCH_SEG(5); break;
            case 4: BOT_LAUNCH_SEG(4); break;
            case 3: BOT_LAUNCH_SEG(3); break;
            case 2: BOT_LAUNCH_SEG(2); break;
            default: BOT_LAUNCH_SEG(1); break;
        }
#undef BOT_LAUNCH_SEG
        w0 += wt;
    }
    return hip_status("segment_sum launch");
}

int bot_degrees_i64(const int32_t* indptr, int64_t n_rows, int64_t* deg, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_rows >= 0, BOT_E_RANGE, "degrees: n_rows=%lld", (long long)n_rows);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(indptr && deg, BOT_E_NULL, "degrees: NULL pointer");
    hipLaunchKernelGGL(degrees_kernel, dim3((unsigned)((n_rows + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       indptr, n_rows, deg);
    return hip_status("degrees launch");
}

}  // extern "C"
