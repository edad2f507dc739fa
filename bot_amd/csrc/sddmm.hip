// SDDMM kernels for gfx950.
//
// sddmm_dot: out[k,h] = < x[indices[k],h,:] , y[r,h,:] >  — the gradient of u_mul_e_sum with respect
// to the edge weights.  Same work decomposition and gather shape as spmm.hip: one LANES-wide group
// per (work item, head); the destination slab y[r,h,:] sits in registers for the whole item, the
// neighbour rows stream through with U rows in flight, and the U per-lane partial dot products are
// reduced together by a transposing butterfly (log2(U) exchange steps that halve the number of live
// values, then log2(LANES/U) plain steps) so the cross-lane cost is ~1.3 shuffles per neighbour
// instead of 6.  Each edge is written exactly once; no atomics.
//
// HBM roofline: algorithmic bytes per launch = 4*[2*n*H*D + nnz + nnz*H].
#include "common.h"

namespace bot {

struct DotArgs {
    const int32_t* indices;
    const int4* items;
    int64_t n_items;
    const float* x;
    int64_t ldx, hsx;
    const float* y;
    int64_t ldy, hsy;
    int32_t H, D;
    float* out;
    const int32_t* operm;
    int32_t accumulate;
};

// Reduce U values per lane across a LANES-wide group.  On return lane l holds the group total of
// value number (l % U).
template <int LANES, int U>
__device__ __forceinline__ float transpose_reduce(float (&p)[U], int lane) {
    static_assert(U == 4 || U == 8, "U");
    if constexpr (U == 8) {
        const bool hi = lane & 4;
        float q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float keep = hi ? p[i + 4] : p[i];
            const float send = hi ? p[i] : p[i + 4];
            q[i] = keep + __shfl_xor(send, 4, LANES);
        }
        const bool h2 = lane & 2;
        float r[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = h2 ? q[i + 2] : q[i];
            const float send = h2 ? q[i] : q[i + 2];
            r[i] = keep + __shfl_xor(send, 2, LANES);
        }
        const bool h1 = lane & 1;
        float s = (h1 ? r[1] : r[0]) + __shfl_xor(h1 ? r[0] : r[1], 1, LANES);
#pragma unroll
        for (int m = 8; m < LANES; m <<= 1) s += __shfl_xor(s, m, LANES);
        return s;  // lane l: value index (l&4 ? 4:0) + (l&2 ? 2:0) + (l&1)  == l % 8
    } else {
        const bool h2 = lane & 2;
        float r[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float keep = h2 ? p[i + 2] : p[i];
            const float send = h2 ? p[i] : p[i + 2];
            r[i] = keep + __shfl_xor(send, 2, LANES);
        }
        const bool h1 = lane & 1;
        float s = (h1 ? r[1] : r[0]) + __shfl_xor(h1 ? r[0] : r[1], 1, LANES);
#pragma unroll
        for (int m = 4; m < LANES; m <<= 1) s += __shfl_xor(s, m, LANES);
        return s;  // lane l: value index l % 4
    }
}

template <int VEC, int LANES, int NCHUNK>
__global__ __launch_bounds__(kBlock) void sddmm_dot_kernel(DotArgs a) {
    constexpr int U = 8;
    const int lane = threadIdx.x % LANES;
    const int64_t gid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
    if (gid >= a.n_items * a.H) return;
    int head = (int)(gid / a.n_items);
    const int64_t item = gid - (int64_t)head * a.n_items;
    const int4 it = a.items[item];
    int row = it.x, beg = it.y, end = it.z;
    if constexpr (LANES == 64) {
        row = __builtin_amdgcn_readfirstlane(row);
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
        head = __builtin_amdgcn_readfirstlane(head);
    }
    const float* xb = a.x + (int64_t)head * a.hsx;
    const float* yb = a.y + (int64_t)row * a.ldy + (int64_t)head * a.hsy;

    int off[NCHUNK];
    float yv[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * LANES + lane) * VEC;
        const bool act = e < a.D;
        off[c] = act ? e : 0;
        vload<VEC>(yv[c], yb + off[c]);
        if (!act) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) yv[c][t] = 0.f;  // idle lanes contribute 0 * x[..0]
        }
    }

    for (int k0 = beg; k0 < end; k0 += LANES) {
        const int k = k0 + lane;
        const int idx = k < end ? a.indices[k] : 0;
        const int cnt = min(LANES, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float p[U];
            float v[U][NCHUNK][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // positions past the end of the row re-read a valid neighbour (clamp) and are not stored
                const int s = group_bcast<LANES>(idx, min(i + u, cnt - 1));
                const float* px = xb + (int64_t)s * a.ldx;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + off[c]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                    for (int t = 0; t < VEC; ++t) s = fmaf(v[u][c][t], yv[c][t], s);
                p[u] = s;
            }
            const float tot = transpose_reduce<LANES, U>(p, lane);
            if (lane < U && i + lane < cnt) {
                const int pos = k0 + i + lane;
                const int64_t o = (int64_t)(a.operm ? a.operm[pos] : pos) * a.H + head;
                a.out[o] = a.accumulate ? a.out[o] + tot : tot;
            }
        }
    }
}

template <int VEC, int LANES, int NCHUNK>
static void launch_dot(const DotArgs& a, hipStream_t st) {
    const int64_t groups = a.n_items * a.H;
    const int64_t blocks = (groups * LANES + kBlock - 1) / kBlock;
    if (blocks == 0) return;
    hipLaunchKernelGGL((sddmm_dot_kernel<VEC, LANES, NCHUNK>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

template <int VEC>
static void dispatch_dot(const DotArgs& a, hipStream_t st) {
    const int L = (a.D + VEC - 1) / VEC;
    if (L <= 8) launch_dot<VEC, 8, 1>(a, st);
    else if (L <= 16) launch_dot<VEC, 16, 1>(a, st);
    else if (L <= 32) launch_dot<VEC, 32, 1>(a, st);
    else if (L <= 64) launch_dot<VEC, 64, 1>(a, st);
    else if (L <= 128) launch_dot<VEC, 64, 2>(a, st);
    else if (L <= 192) launch_dot<VEC, 64, 3>(a, st);
    else launch_dot<VEC, 64, 4>(a, st);
}

// sddmm_dot_bcast: out[k,h] = < x[indices[k],:] , y[r,h,:] > — the source row has NO head axis and is gathered ONCE per edge
// for all HB <= 4 heads; the H destination slabs y[r,h,:] sit in registers.  This is the weight gradient of the
// aggregate-before-project layer (spmm_bcast, spmm.hip) when the layer INPUT needs no gradient (the first layer of a stack:
// its input is data): 4*D bytes gathered per edge instead of the 4*H*D of spmm_dot_bcast, and no transposed sweep at all.
// 4 edges x 4 head slots are reduced together by the 16-value transposing butterfly.
template <int VEC, int LANES, int NCHUNK, int HB>
__global__ __launch_bounds__(kBlock) void sddmm_dot_bcast_kernel(DotArgs a) {
    static_assert(LANES >= 16 && HB <= 4, "butterfly layout");
    constexpr int U = 4;
    const int lane = threadIdx.x % LANES;
    const int64_t item = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
    if (item >= a.n_items) return;
    const int4 it = a.items[item];
    int row = it.x, beg = it.y, end = it.z;
    if constexpr (LANES == 64) {
        row = __builtin_amdgcn_readfirstlane(row);
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
    }
    int off[NCHUNK];
    float yv[HB][NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * LANES + lane) * VEC;
        const bool act = e < a.D;
        off[c] = act ? e : 0;
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            vload<VEC>(yv[h][c], a.y + (int64_t)row * a.ldy + (int64_t)h * a.hsy + off[c]);
            if (!act) {
#pragma unroll
                for (int t = 0; t < VEC; ++t) yv[h][c][t] = 0.f;
            }
        }
    }
    for (int k0 = beg; k0 < end; k0 += LANES) {
        const int k = k0 + lane;
        const int idx = k < end ? a.indices[k] : 0;
        const int cnt = min(LANES, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float p[16], v[U][NCHUNK][VEC];
#pragma unroll
            for (int q = 0; q < 16; ++q) p[q] = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = group_bcast<LANES>(idx, min(i + u, cnt - 1));  // past the end: a valid row again, result not stored
                const float* px = a.x + (int64_t)s * a.ldx;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + off[c]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    float d = 0.f;
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                        for (int t = 0; t < VEC; ++t) d = fmaf(v[u][c][t], yv[h][c][t], d);
                    p[u * 4 + h] = d;  // value slot = 4*u + h
                }
            const float tot = transpose_reduce16<LANES>(p, lane);
            const int slot16 = lane & 15, u = slot16 >> 2, h = slot16 & 3;
            if (lane < 16 && h < HB && i + u < cnt) {
                const int pos = k0 + i + u;
                a.out[(int64_t)(a.operm ? a.operm[pos] : pos) * HB + h] = tot;
            }
        }
    }
}

template <int VEC, int HB>
static void dispatch_dot_bcast(const DotArgs& a, hipStream_t st) {
    const int L = (a.D + VEC - 1) / VEC;
#define BOT_DOTB(LN, NC)                                                                                                    \
    do {                                                                                                                    \
        const int64_t blocks = (a.n_items * LN + kBlock - 1) / kBlock;                                                      \
        if (blocks == 0) break;                                                                                             \
        set_kernel("bot::sddmm_dot_bcast_kernel<%d,%d,%d,%d>", VEC, LN, NC, HB);                                           \
        hipLaunchKernelGGL((sddmm_dot_bcast_kernel<VEC, LN, NC, HB>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);      \
    } while (0)
    if (L <= 16) BOT_DOTB(16, 1);
    else if (L <= 32) BOT_DOTB(32, 1);
    else if (L <= 64) BOT_DOTB(64, 1);
    else if (L <= 128) BOT_DOTB(64, 2);
    else BOT_DOTB(64, 4);
#undef BOT_DOTB
}

// e[i,:] = x[src[i],:] (+ y[dst[i],:])
__global__ __launch_bounds__(kBlock) void u_add_v_kernel(const int32_t* src, const int32_t* dst, int64_t n_edges,
                                                        const float* x, const float* y, int32_t W, float* out) {
    const int64_t total = n_edges * W;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < total; g += (int64_t)gridDim.x * kBlock) {
        const int64_t e = g / W;
        const int j = (int)(g - e * W);
        float v = x[(int64_t)src[e] * W + j];
        if (y) v += y[(int64_t)dst[e] * W + j];
        out[g] = v;
    }
}

}  // namespace bot

extern "C" {

int bot_sddmm_dot_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                      int64_t n_items, const float* x, int64_t ldx, int64_t hsx, const float* y, int64_t ldy, int64_t hsy,
                      int32_t H, int32_t D, float* out, const int32_t* operm, int32_t accumulate, bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0, BOT_E_RANGE, "sddmm_dot: negative size");
    BOT_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, BOT_E_RANGE, "sddmm_dot: int32 index range exceeded");
    BOT_REQUIRE(H >= 1 && D >= 1, BOT_E_RANGE, "sddmm_dot: H=%d D=%d must be >= 1", H, D);
    if (nnz == 0 || n_rows == 0) return 0;
    BOT_REQUIRE(indices && items && x && y && out, BOT_E_NULL, "sddmm_dot: NULL pointer");
    BOT_REQUIRE(hsx >= D && hsy >= D && ldx >= (int64_t)(H - 1) * hsx + D && ldy >= (int64_t)(H - 1) * hsy + D, BOT_E_RANGE,
                "sddmm_dot: strides smaller than the slab");
    hipStream_t st = (hipStream_t)stream;
    // one launch covers up to 1024 (VEC 4) / 512 / 256 floats of D; wider slabs are tiled and accumulated
    const int vec = pick_vec(D, {ldx, hsx, ldy, hsy}, {x, y});
    const int cap = vec * 256;
    for (int d0 = 0; d0 < D; d0 += cap) {
        DotArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x + d0, ldx, hsx, y + d0, ldy, hsy, H,
                  (int32_t)(D - d0 < cap ? D - d0 : cap), out, operm, (d0 > 0 || accumulate) ? 1 : 0};
        if (vec == 4) dispatch_dot<4>(a, st);
        else if (vec == 2) dispatch_dot<2>(a, st);
        else dispatch_dot<1>(a, st);
    }
    return hip_status("sddmm_dot launch");
}

int bot_sddmm_dot_bcast_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                            int64_t n_items, const float* x, int64_t ldx, const float* y, int64_t ldy, int64_t hsy, int32_t H,
                            int32_t D, float* out, const int32_t* operm, bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && nnz < INT32_MAX && n_rows < INT32_MAX, BOT_E_RANGE, "sddmm_dot_bcast: bad size");
    BOT_REQUIRE(H >= 1 && H <= 4 && D >= 1 && D <= 1024, BOT_E_RANGE, "sddmm_dot_bcast: H=%d (1..4) D=%d (1..1024)", H, D);
    if (nnz == 0 || n_rows == 0) return 0;
    BOT_REQUIRE(indices && items && x && y && out, BOT_E_NULL, "sddmm_dot_bcast: NULL pointer");
    BOT_REQUIRE(ldx >= D && ldy >= D && (hsy >= D || H == 1), BOT_E_RANGE, "sddmm_dot_bcast: strides smaller than the slab");
    hipStream_t st = (hipStream_t)stream;
    const int vec = pick_vec(D, {ldx, ldy, H > 1 ? hsy : 0}, {x, y});
    BOT_REQUIRE(D <= vec * 256, BOT_E_RANGE, "sddmm_dot_bcast: D=%d exceeds one launch tile", D);
    DotArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, 0, y, ldy, hsy, H, D, out, operm, 0};
#define BOT_DOTB_H(V)                                       \
    switch (H) {                                            \
        case 1: dispatch_dot_bcast<V, 1>(a, st); break;     \
        case 2: dispatch_dot_bcast<V, 2>(a, st); break;     \
        case 3: dispatch_dot_bcast<V, 3>(a, st); break;     \
        default: dispatch_dot_bcast<V, 4>(a, st); break;    \
    }
    if (vec == 4) { BOT_DOTB_H(4) } else if (vec == 2) { BOT_DOTB_H(2) } else { BOT_DOTB_H(1) }
#undef BOT_DOTB_H
    return hip_status("sddmm_dot_bcast launch");
}

int bot_sddmm_u_add_v_f32(const int32_t* src, const int32_t* dst, int64_t n_edges, const float* x, const float* y,
                          int32_t W, float* out, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_edges >= 0 && W >= 1, BOT_E_RANGE, "u_add_v: n_edges=%lld W=%d", (long long)n_edges, W);
    if (n_edges == 0) return 0;
    BOT_REQUIRE(src && x && out && (y == nullptr || dst), BOT_E_NULL, "u_add_v: NULL pointer");
    const int64_t total = n_edges * W;
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(u_add_v_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, src, dst, n_edges, x, y,
                       W, out);
    return hip_status("u_add_v launch");
}

}  // extern "C"
