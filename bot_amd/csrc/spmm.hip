// CSR/CSC SpMM for gfx950: out[r,h,:] = sum_k w[k,h] * x[indices[k],h,:].
//
// Shape of the kernel (MI355X_MICROARCH.md "Indexed rows", cdna_hip_programming.md Appendix B
// "Scatter / gather"): one LANES-wide lane group per (work item, head, feature tile); lanes run
// across the feature dimension with 4/8/16-byte loads so every neighbour row is one coalesced
// segment; the neighbour ids (and weights) of a row are fetched LANES at a time with one coalesced
// load, then broadcast lane by lane (v_readlane -> SGPR base address for full-wave groups), four
// neighbour rows in flight per group; the per-destination sum stays in registers in neighbour
// order and is stored once with plain stores — no atomics, bitwise reproducible.  Long rows are
// split by the row plan into chunk-sized items whose partial sums are added in slot order by
// spmm_combine_kernel.  Work is ordered head-major (see common.h).
//
// HBM roofline: algorithmic bytes per launch = 4*[2*n*H*D + nnz + (n+1) + (w? nnz*H : 0)].
#include <hip/hip_fp16.h>

#include "common.h"

#include <stdlib.h>
#include <algorithm>
#include <string.h>

namespace bot {

struct SpmmArgs {
    const int32_t* indices;
    const int4* items;
    int64_t n_items;
    const float* x;
    int64_t ldx, hsx;
    const float* w;
    const int32_t* wperm;
    int32_t H, D, n_tiles;
    float* out;
    int64_t ldo, hso;
    float* partial;
    int64_t ldp;
    // fused transposed-SpMM + SDDMM-dot (backward of u_mul_e_sum): row-local slab y and the per-edge dot output
    const float* y;
    int64_t ldy, hsy;
    float* dot_out;
    // optional epilogue of the forward SpMM: out[r,h,:] += addend[r,h,:] (the layer's residual branch, models.py:558-560)
    const float* addend;
    int64_t lda, hsa;
    // optional by-product of the fused backward: max|out| into kAbsmaxSlots words (common.h absmax_publish)
    uint32_t* absmax;
    // spmm_bcast only: the result as fp16 halves [h1 | 2^11 h2] of hscale[0] * out (halves.hip), h1 at hout[r, h hsh + e] and the second
    // half h2_off columns behind it, zeros in the columns D .. hpiece - 1 of a head's block; nothing is written to `out`
    __half* hout;
    int64_t ldh, hsh;
    int32_t h2_off, hpiece;
    const float* hscale;
};

// VEC consecutive entries of a halves operand: h1 at p, 2^11 h2 at p + h2_off (see halves.hip halves_split_kernel<2>)
template <int VEC>
__device__ __forceinline__ void store_halves(__half* p, int32_t h2_off, const float (&v)[VEC], float s) {
    __half a[VEC], b[VEC];
#pragma unroll
    for (int t = 0; t < VEC; ++t) {
        const float z = v[t] * s;
        a[t] = __float2half_rn(z);
        b[t] = __float2half_rn((z - __half2float(a[t])) * 2048.0f);
    }
    if constexpr (VEC == 4) {
        *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(a);
        *reinterpret_cast<uint2*>(p + h2_off) = *reinterpret_cast<const uint2*>(b);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<uint32_t*>(p) = *reinterpret_cast<const uint32_t*>(a);
        *reinterpret_cast<uint32_t*>(p + h2_off) = *reinterpret_cast<const uint32_t*>(b);
    } else {
        p[0] = a[0], p[h2_off] = b[0];
    }
}

// Backward of u_mul_e_sum in ONE sweep over the out-edges (CSR direction) — each gathered row dx[v,h,:] is used twice:
//   out[u,h,:]           = sum_k w[wperm[k],h] * x[indices[k],h,:]        (d ft: transposed SpMM)
//   dot_out[wperm[k],h]  = < y[u,h,:] , x[indices[k],h,:] >               (d a : SDDMM dot)
// so the layer's backward needs one E*H*D gather instead of two.  Same decomposition as spmm_kernel; D must fit one
// launch tile (n_tiles == 1), the dot is reduced 4 neighbours at a time with the transposing butterfly.
template <int VEC, int LANES, int NCHUNK>
__global__ __launch_bounds__(kBlock) void spmm_dot_kernel(SpmmArgs a) {
    constexpr int U = 4;
    const int lane = threadIdx.x % LANES;
    const int64_t gid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
    if (gid >= a.n_items * a.H) return;
    int head = (int)(gid / a.n_items);
    const int64_t item = gid - (int64_t)head * a.n_items;
    const int4 it = a.items[item];
    int row = it.x, beg = it.y, end = it.z, slot = it.w;
    if constexpr (LANES == 64) {
        row = __builtin_amdgcn_readfirstlane(row);
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
        slot = __builtin_amdgcn_readfirstlane(slot);
        head = __builtin_amdgcn_readfirstlane(head);
    }
    const float* xb = a.x + (int64_t)head * a.hsx;
    const float* yb = a.y + (int64_t)row * a.ldy + (int64_t)head * a.hsy;
    int off[NCHUNK];
    bool act[NCHUNK];
    float acc[NCHUNK][VEC], yv[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * LANES + lane) * VEC;
        act[c] = e < a.D;
        off[c] = act[c] ? e : 0;
        vload<VEC>(yv[c], yb + off[c]);
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            acc[c][t] = 0.f;
            if (!act[c]) yv[c][t] = 0.f;
        }
    }
    for (int k0 = beg; k0 < end; k0 += LANES) {
        const int k = k0 + lane;
        int idx = 0, wp = 0;
        float wv = 0.f;
        if (k < end) {
            idx = a.indices[k];
            wp = a.wperm ? a.wperm[k] : k;
            wv = a.w[(int64_t)wp * a.H + head];
        }
        const int cnt = min(LANES, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float v[U][NCHUNK][VEC], ww[U], p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);  // past the end: re-read a valid neighbour with weight 0, result not stored
                const int s = group_bcast<LANES>(idx, j);
                ww[u] = i + u < cnt ? group_bcast<LANES>(wv, j) : 0.f;
                const float* px = xb + (int64_t)s * a.ldx;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + off[c]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float d = 0.f;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                    for (int t = 0; t < VEC; ++t) {
                        acc[c][t] = fmaf(ww[u], v[u][c][t], acc[c][t]);
                        d = fmaf(v[u][c][t], yv[c][t], d);
                    }
                p[u] = d;
            }
            const float tot = transpose_reduce4<LANES>(p, lane);
            const int mywp = __shfl(wp, i + (lane & 3), LANES);  // position whose dot this lane now holds
            if (lane < U && i + lane < cnt) a.dot_out[(int64_t)mywp * a.H + head] = tot;
        }
    }
    float* ob = slot < 0 ? a.out + (int64_t)row * a.ldo + (int64_t)head * a.hso
                         : a.partial + (int64_t)slot * a.ldp + (int64_t)head * a.D;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
        if (act[c]) vstore<VEC>(ob + off[c], acc[c]);
}

// The same fused backward with ONE wavefront per work item covering ALL heads of the row.  CPH == 1: each head occupies a
// segment of HL (16/32/64) lanes, 64/HL heads per 64-lane chunk; CPH > 1 (HL == 64): each head spans CPH whole chunks.
// A neighbour row is then ONE contiguous H*D*4-byte read instead of H reads of D*4 bytes issued by different workgroups
// at different times, and the H dot products of an edge are stored back to back into one 4*H-byte record instead of by H
// unrelated workgroups (H separate partial-line writes).  Measured against the head-major kernel above: S-products H=4
// D=120 83.9 -> 51.1 ms, S-proteins H=6 D=80 35.7 -> 21.6 ms, S-arxiv H=3 D=40 0.53 -> 0.31 ms.
template <int VEC, int HL, int NCHUNK, int CPH>
__global__ __launch_bounds__(kBlock) void spmm_dot_rows_kernel(SpmmArgs a) {
    static_assert(CPH == 1 || HL == 64, "multi-chunk heads use whole waves");
    static_assert(NCHUNK % CPH == 0, "whole heads only");
    constexpr int U = 4;
    constexpr int HPC = 64 / HL;        // heads per chunk (CPH == 1)
    constexpr int NSLOT = NCHUNK / CPH;  // dot-product slots per lane segment
    const int lane = threadIdx.x & 63;
    const int64_t item = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (item >= a.n_items) return;
    const int4 it = a.items[item];
    const int row = __builtin_amdgcn_readfirstlane(it.x), beg = __builtin_amdgcn_readfirstlane(it.y);
    const int end = __builtin_amdgcn_readfirstlane(it.z), slot = __builtin_amdgcn_readfirstlane(it.w);
    const int hl = lane & (HL - 1);
    int xoff[NCHUNK], hd[NCHUNK], el[NCHUNK];
    bool act[NCHUNK], live[NCHUNK];  // live: the lane's head exists; act: and its elements lie inside the head slab
    float acc[NCHUNK][VEC], yv[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int head = CPH > 1 ? c / CPH : c * HPC + lane / HL;
        const int e = CPH > 1 ? ((c % CPH) * 64 + lane) * VEC : hl * VEC;
        live[c] = head < a.H;
        act[c] = live[c] && e < a.D;
        hd[c] = live[c] ? head : 0;
        el[c] = e;
        xoff[c] = act[c] ? (int)(head * a.hsx) + e : 0;  // idle lanes re-read element 0: in bounds, never stored
        const float* yb = a.y + (int64_t)row * a.ldy + (act[c] ? head * a.hsy + e : 0);
        vload<VEC>(yv[c], yb);
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            acc[c][t] = 0.f;
            if (!act[c]) yv[c][t] = 0.f;
        }
    }
    for (int k0 = beg; k0 < end; k0 += 64) {
        const int k = k0 + lane;
        int idx = 0, wp = 0;
        if (k < end) {
            idx = a.indices[k];
            wp = a.wperm ? a.wperm[k] : k;
        }
        const int cnt = min(64, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float v[U][NCHUNK][VEC], ww[U][NSLOT], p[NSLOT][U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);  // past the end: re-read a valid neighbour with weight 0, result not stored
                const int s = __builtin_amdgcn_readlane(idx, j);
                const int ps = __builtin_amdgcn_readlane(wp, j);
                const float* px = a.x + (int64_t)s * a.ldx;
                const float* pw = a.w + (int64_t)ps * a.H;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + xoff[c]);
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) ww[u][q] = i + u < cnt ? pw[hd[q * CPH]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    float d = 0.f;
#pragma unroll
                    for (int cc = 0; cc < CPH; ++cc) {
                        const int c = q * CPH + cc;
#pragma unroll
                        for (int t = 0; t < VEC; ++t) {
                            acc[c][t] = fmaf(ww[u][q], v[u][c][t], acc[c][t]);
                            d = fmaf(v[u][c][t], yv[c][t], d);
                        }
                    }
                    p[q][u] = d;
                }
            const int mywp = __shfl(wp, i + (lane & 3));  // position whose dot products this lane holds after the reduction
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
                const float tot = transpose_reduce4<HL>(p[q], hl);
                if (hl < U && i + hl < cnt && live[q * CPH]) a.dot_out[(int64_t)mywp * a.H + hd[q * CPH]] = tot;
            }
        }
    }
    if (a.hout && slot < 0) {            // (bot_spmm_dot_halves_f16) the row's sums straight as a halves operand, no fp32 copy
        const float hs = a.hscale[0];
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
            if (act[c]) store_halves<VEC>(a.hout + (int64_t)row * a.ldh + (int64_t)hd[c] * a.hsh + el[c], a.h2_off, acc[c], hs);
        return;
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
        if (act[c]) {
            float* ob = slot < 0 ? a.out + (int64_t)row * a.ldo + (int64_t)hd[c] * a.hso + el[c]
                                 : a.partial + (int64_t)slot * a.ldp + (int64_t)hd[c] * a.D + el[c];
            vstore<VEC>(ob, acc[c]);
        }
    if (a.absmax) {                      // wave-uniform; the chunks of a long row are finished (and measured) by spmm_combine_kernel
        float m = 0.f;
        if (slot < 0) {
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                for (int t = 0; t < VEC; ++t)
                    if (act[c]) m = fmaxf(m, fabsf(acc[c][t]));
        }
        absmax_publish(wave_absmax(m), a.absmax);
    }
}

// The weighted forward SpMM in the same all-heads layout as spmm_dot_rows_kernel (one wavefront per work item, a head = HL
// lanes or CPH whole chunks): a neighbour row is one contiguous H*D*4-byte read (see dispatch_spmm_rows); the head-major
// kernel below keeps H = 1, unweighted sums and the shapes that do not fit.
template <int VEC, int HL, int NCHUNK, int CPH>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(SpmmArgs a) {
    static_assert(CPH == 1 || HL == 64, "multi-chunk heads use whole waves");
    constexpr int U = 4;
    constexpr int HPC = 64 / HL;
    constexpr int NSLOT = NCHUNK / CPH;
    const int lane = threadIdx.x & 63;
    const int64_t item = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (item >= a.n_items) return;
    const int4 it = a.items[item];
    const int row = __builtin_amdgcn_readfirstlane(it.x), beg = __builtin_amdgcn_readfirstlane(it.y);
    const int end = __builtin_amdgcn_readfirstlane(it.z), slot = __builtin_amdgcn_readfirstlane(it.w);
    const int hl = lane & (HL - 1);
    int xoff[NCHUNK], hd[NCHUNK], el[NCHUNK];
    bool act[NCHUNK];
    float acc[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int head = CPH > 1 ? c / CPH : c * HPC + lane / HL;
        const int e = CPH > 1 ? ((c % CPH) * 64 + lane) * VEC : hl * VEC;
        act[c] = head < a.H && e < a.D;
        hd[c] = head < a.H ? head : 0;
        el[c] = e;
        xoff[c] = act[c] ? (int)(head * a.hsx) + e : 0;
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[c][t] = 0.f;
    }
    for (int k0 = beg; k0 < end; k0 += 64) {
        const int k = k0 + lane;
        int idx = 0, wp = 0;
        if (k < end) {
            idx = a.indices[k];
            wp = a.wperm ? a.wperm[k] : k;
        }
        const int cnt = min(64, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float v[U][NCHUNK][VEC], ww[U][NSLOT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);
                const int s = __builtin_amdgcn_readlane(idx, j);
                const int ps = __builtin_amdgcn_readlane(wp, j);
                const float* px = a.x + (int64_t)s * a.ldx;
                const float* pw = a.w + (int64_t)ps * a.H;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + xoff[c]);
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) ww[u][q] = i + u < cnt ? pw[hd[q * CPH]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                    for (int t = 0; t < VEC; ++t) acc[c][t] = fmaf(ww[u][c / CPH], v[u][c][t], acc[c][t]);
        }
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
        if (act[c]) {
            if (a.addend && slot < 0) {
                float r[VEC];
                vload<VEC>(r, a.addend + (int64_t)row * a.lda + (int64_t)hd[c] * a.hsa + el[c]);
#pragma unroll
                for (int t = 0; t < VEC; ++t) acc[c][t] += r[t];
            }
            float* ob = slot < 0 ? a.out + (int64_t)row * a.ldo + (int64_t)hd[c] * a.hso + el[c]
                                 : a.partial + (int64_t)slot * a.ldp + (int64_t)hd[c] * a.D + el[c];
            vstore<VEC>(ob, acc[c]);
        }
}

// The weighted all-heads forward for rows whose head width is NOT a multiple of 4 floats (ogbn-arxiv: 3 x 250): the head-segment
// layouts above then fall back to 8-byte loads (six per neighbour row and lane), which halves the L2-resident gather rate
// (0.56 vs 0.38 ms at S-arxiv with every source in the L2, profiles/r02_spmm_lane_width.txt).  When the row is H*D CONTIGUOUS
// floats on a 16-byte aligned pitch it is read here as flat float4 lanes instead — three 16-byte loads per neighbour row and lane —
// and the head of every element is resolved per lane: a float4 straddles at most one head boundary (D >= 4), so a lane-chunk holds
// two weights, `wa` for its first `ns` elements and `wb` for the rest.  The per-edge weights are wave-uniform scalar loads.
// The tail float4 of a row may reach past F = H*D (750 -> elements 750, 751): the caller guarantees ldx >= roundup4(F), the extra
// lanes are multiplied into accumulators that are never stored.  Same summation order per element as the head-segment kernels.
template <int NCHUNK, int HMAX>
__global__ __launch_bounds__(kBlock) void spmm_flat_kernel(SpmmArgs a) {
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int64_t item = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (item >= a.n_items) return;
    const int4 it = a.items[item];
    const int row = __builtin_amdgcn_readfirstlane(it.x), beg = __builtin_amdgcn_readfirstlane(it.y);
    const int end = __builtin_amdgcn_readfirstlane(it.z), slot = __builtin_amdgcn_readfirstlane(it.w);
    const int F = a.H * a.D;
    int off[NCHUNK], ha[NCHUNK], hb[NCHUNK], ns[NCHUNK], nv[NCHUNK];
    float acc[NCHUNK][4];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * 64 + lane) * 4;
        nv[c] = min(4, max(0, F - e));                  // elements of this lane-chunk inside the row
        off[c] = nv[c] > 0 ? e : 0;                     // idle lanes re-read the row's first float4: in bounds, never stored
        const int h0 = min(off[c] / a.D, a.H - 1);
        ha[c] = h0, hb[c] = min(h0 + 1, a.H - 1);
        ns[c] = min(4, (h0 + 1) * a.D - off[c]);        // how many of the four elements belong to head h0
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = 0.f;
    }
    for (int k0 = beg; k0 < end; k0 += 64) {
        const int k = k0 + lane;
        int idx = 0, wp = 0;
        if (k < end) {
            idx = a.indices[k];
            wp = a.wperm ? a.wperm[k] : k;
        }
        const int cnt = min(64, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float v[U][NCHUNK][4], wh[U][HMAX];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);      // past the end: re-read a valid neighbour with weight 0
                const int s = __builtin_amdgcn_readlane(idx, j);
                const int ps = __builtin_amdgcn_readlane(wp, j);
                const float* px = a.x + (int64_t)s * a.ldx;
                const float* pw = a.w + (int64_t)ps * a.H;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<4>(v[u][c], px + off[c]);
#pragma unroll
                for (int h = 0; h < HMAX; ++h) wh[u][h] = (i + u < cnt && h < a.H) ? pw[h] : 0.f;   // uniform address: scalar loads
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) {
                    float wa = wh[u][0], wb = wh[u][0];
#pragma unroll
                    for (int h = 1; h < HMAX; ++h) {
                        wa = ha[c] == h ? wh[u][h] : wa;
                        wb = hb[c] == h ? wh[u][h] : wb;
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[c][t] = fmaf(t < ns[c] ? wa : wb, v[u][c][t], acc[c][t]);
                }
        }
    }
    // epilogue (not hot): element-wise stores, two or four at a time where the destination allows — the partial slab and a
    // row-contiguous output have a pitch of F floats, the residual block of a merged GEMM output starts 8-byte aligned only
    float* ob = slot < 0 ? a.out + (int64_t)row * a.ldo : a.partial + (int64_t)slot * a.ldp;
    const float* ab = (a.addend && slot < 0) ? a.addend + (int64_t)row * a.lda : nullptr;
    const bool o16 = ((reinterpret_cast<uintptr_t>(ob) & 15) == 0), o8 = ((reinterpret_cast<uintptr_t>(ob) & 7) == 0);
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        if (nv[c] <= 0) continue;
        if (ab) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < nv[c]) acc[c][t] += ab[off[c] + t];
        }
        if (nv[c] == 4 && o16) vstore<4>(ob + off[c], acc[c]);
        else if (o8 && (nv[c] & 1) == 0) {
            float lo[2] = {acc[c][0], acc[c][1]}, hi[2] = {acc[c][2], acc[c][3]};
            vstore<2>(ob + off[c], lo);
            if (nv[c] == 4) vstore<2>(ob + off[c] + 2, hi);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < nv[c]) ob[off[c] + t] = acc[c][t];
        }
    }
}

template <int VEC, int LANES, int NCHUNK, bool WEIGHTED>
__global__ __launch_bounds__(kBlock) void spmm_kernel(SpmmArgs a) {
    constexpr int TILE = VEC * LANES * NCHUNK;
    constexpr int U = 4;
    const int lane = threadIdx.x % LANES;
    const int64_t gid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
    if (gid >= a.n_items * a.H * a.n_tiles) return;  // whole groups leave together
    const int64_t ht = gid / a.n_items;
    const int64_t item = gid - ht * a.n_items;
    int head = (int)(ht / a.n_tiles);
    int tile = (int)(ht - (int64_t)head * a.n_tiles);
    const int4 it = a.items[item];
    int row = it.x, beg = it.y, end = it.z, slot = it.w;
    if constexpr (LANES == 64) {  // wave-uniform: keep them in SGPRs
        row = __builtin_amdgcn_readfirstlane(row);
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
        slot = __builtin_amdgcn_readfirstlane(slot);
        head = __builtin_amdgcn_readfirstlane(head);
        tile = __builtin_amdgcn_readfirstlane(tile);
    }
    const int doff = tile * TILE;
    const int dcount = min(TILE, a.D - doff);
    const float* xb = a.x + (int64_t)head * a.hsx + doff;

    int off[NCHUNK];
    bool act[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * LANES + lane) * VEC;
        act[c] = e < dcount;
        off[c] = act[c] ? e : 0;  // idle lanes re-read element 0: always in bounds, never stored
    }
    float acc[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[c][t] = 0.f;

    for (int k0 = beg; k0 < end; k0 += LANES) {
        const int k = k0 + lane;
        int idx = 0;
        float wv = 0.f;
        if (k < end) {
            idx = a.indices[k];
            if constexpr (WEIGHTED) {
                const int wp = a.wperm ? a.wperm[k] : k;
                wv = a.w[(int64_t)wp * a.H + head];
            }
        }
        const int cnt = min(LANES, end - k0);
        int i = 0;
        for (; i + U <= cnt; i += U) {
            float v[U][NCHUNK][VEC];
            float ww[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = group_bcast<LANES>(idx, i + u);
                if constexpr (WEIGHTED) ww[u] = group_bcast<LANES>(wv, i + u);
                const float* p = xb + (int64_t)s * a.ldx;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], p + off[c]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                    for (int t = 0; t < VEC; ++t) {
                        if constexpr (WEIGHTED) acc[c][t] = fmaf(ww[u], v[u][c][t], acc[c][t]);
                        else acc[c][t] += v[u][c][t];
                    }
        }
        for (; i < cnt; ++i) {
            const int s = group_bcast<LANES>(idx, i);
            float w1 = 1.f;
            if constexpr (WEIGHTED) w1 = group_bcast<LANES>(wv, i);
            const float* p = xb + (int64_t)s * a.ldx;
            float v[NCHUNK][VEC];
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[c], p + off[c]);
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    if constexpr (WEIGHTED) acc[c][t] = fmaf(w1, v[c][t], acc[c][t]);
                    else acc[c][t] += v[c][t];
                }
        }
    }

    float* ob = slot < 0 ? a.out + (int64_t)row * a.ldo + (int64_t)head * a.hso + doff
                         : a.partial + (int64_t)slot * a.ldp + (int64_t)head * a.D + doff;
    const float* ab = (a.addend && slot < 0) ? a.addend + (int64_t)row * a.lda + (int64_t)head * a.hsa + doff : nullptr;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
        if (act[c]) {
            if (ab) {
                float r[VEC];
                vload<VEC>(r, ab + off[c]);
#pragma unroll
                for (int t = 0; t < VEC; ++t) acc[c][t] += r[t];
            }
            vstore<VEC>(ob + off[c], acc[c]);
        }
}

// out[row,h,d] = partial[first slot] + ... + partial[last slot] (+ addend), in slot order.
__global__ __launch_bounds__(kBlock) void spmm_combine_kernel(const int32_t* long_rows, const int32_t* long_ptr,
                                                             int64_t n_long, int32_t H, int32_t D, const float* partial,
                                                             int64_t ldp, float* out, int64_t ldo, int64_t hso,
                                                             const float* addend = nullptr, int64_t lda = 0, int64_t hsa = 0,
                                                             uint32_t* absmax = nullptr) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t hd = (int64_t)H * D;
    float s = 0.f;
    if (gid < n_long * hd) {
        const int64_t i = gid / hd;
        const int e = (int)(gid - i * hd);
        const int h = e / D, d = e - h * D;
        int p = long_ptr[i];
        const int p1 = long_ptr[i + 1];
        for (; p + 4 <= p1; p += 4) {       // four loads in flight, added in slot order
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = partial[(int64_t)(p + j) * ldp + e];
#pragma unroll
            for (int j = 0; j < 4; ++j) s += v[j];
        }
        for (; p < p1; ++p) s += partial[(int64_t)p * ldp + e];
        if (addend) s += addend[(int64_t)long_rows[i] * lda + (int64_t)h * hsa + d];
        out[(int64_t)long_rows[i] * ldo + (int64_t)h * hso + d] = s;
    }
    if (absmax) absmax_publish(wave_absmax(fabsf(s)), absmax);     // all 64 lanes arrive here (no early return above)
}

// ---------------------------------------------------------------------------------------------
// Aggregate-before-project forms ("aggregate first then mult W" is what GraphConv already does when the input is
// narrower than the output, models.py:377-385; the same reordering for GAT: sum_e a[e,h] (W_h x[u]) = W_h (sum_e a[e,h] x[u])).
// The source row x[u,:] has NO head axis and is gathered ONCE per edge for all H heads:
//   spmm_bcast     out[r,h,:]          = sum_k w[wperm[k],h] * x[indices[k],:]
//   spmm_dot_bcast out[r,:]            = sum_k sum_h w[wperm[k],h] * x[indices[k],h,:]      (backward: d of the source rows)
//                  dot_out[wperm[k],h] = < y[r,:] , x[indices[k],h,:] >                      (backward: d of the weights)
// At BASELINE config 2, layer 0 (168 -> 3 x 250) this gathers 672 B per edge forward and 2 016 B backward instead of
// 3 000 B + 3 000 B, and in partitioned mode shrinks that layer's halo rows by the same 4.5x.
// ---------------------------------------------------------------------------------------------
template <int VEC, int LANES, int NCHUNK, int HB>
__global__ __launch_bounds__(kBlock) void spmm_bcast_kernel(SpmmArgs a) {
    constexpr int U = 4;
    const int lane = threadIdx.x % LANES;
    const int64_t item = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
    if (item >= a.n_items) return;
    const int4 it = a.items[item];
    int row = it.x, beg = it.y, end = it.z, slot = it.w;
    if constexpr (LANES == 64) {
        row = __builtin_amdgcn_readfirstlane(row);
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
        slot = __builtin_amdgcn_readfirstlane(slot);
    }
    int off[NCHUNK];
    bool act[NCHUNK];
    float acc[HB][NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * LANES + lane) * VEC;
        act[c] = e < a.D;
        off[c] = act[c] ? e : 0;
#pragma unroll
        for (int h = 0; h < HB; ++h)
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[h][c][t] = 0.f;
    }
    for (int k0 = beg; k0 < end; k0 += LANES) {
        const int k = k0 + lane;
        int idx = 0;
        float wv[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) wv[h] = 0.f;
        if (k < end) {
            idx = a.indices[k];
            const int wp = a.wperm ? a.wperm[k] : k;
#pragma unroll
            for (int h = 0; h < HB; ++h) wv[h] = a.w[(int64_t)wp * HB + h];
        }
        const int cnt = min(LANES, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float v[U][NCHUNK][VEC], ww[U][HB];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);
                const int s = group_bcast<LANES>(idx, j);
#pragma unroll
                for (int h = 0; h < HB; ++h) ww[u][h] = i + u < cnt ? group_bcast<LANES>(wv[h], j) : 0.f;
                const float* px = a.x + (int64_t)s * a.ldx;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + off[c]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int h = 0; h < HB; ++h)
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                        for (int t = 0; t < VEC; ++t) acc[h][c][t] = fmaf(ww[u][h], v[u][c][t], acc[h][c][t]);
        }
    }
    if (a.hout && slot < 0) {           // the row's result as fp16 halves (zeros behind the D columns of each head's block)
        const float s = a.hscale[0];
#pragma unroll
        for (int h = 0; h < HB; ++h) {
            __half* ob = a.hout + (int64_t)row * a.ldh + (int64_t)h * a.hsh;
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
                const int e = (c * LANES + lane) * VEC;
                if (e < a.hpiece) store_halves<VEC>(ob + e, a.h2_off, acc[h][c], act[c] ? s : 0.f);      // (lanes past D hold a copy of column 0)
            }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < HB; ++h) {
        float* ob = slot < 0 ? a.out + (int64_t)row * a.ldo + (int64_t)h * a.hso : a.partial + (int64_t)slot * a.ldp + (int64_t)h * a.D;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
            if (act[c]) vstore<VEC>(ob + off[c], acc[h][c]);
    }
}

// halves form of spmm_combine_kernel: hout[row, h hsh + e] = halves of hscale[0] * (partial[first slot] + ... + partial[last slot]), e < D;
// zeros for D <= e < hpiece
__global__ __launch_bounds__(kBlock) void spmm_combine_halves_kernel(const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, int32_t H,
                                                                    int32_t D, const float* partial, int64_t ldp, __half* hout, int64_t ldh,
                                                                    int64_t hsh, int32_t h2_off, int32_t hpiece, const float* hscale) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t hp = (int64_t)H * hpiece;
    if (gid >= n_long * hp) return;
    const int64_t i = gid / hp;
    const int e2 = (int)(gid - i * hp);
    const int h = e2 / hpiece, d = e2 - h * hpiece;
    float v[1] = {0.f};
    if (d < D) {
        const int e = h * D + d;
        for (int p = long_ptr[i]; p < long_ptr[i + 1]; ++p) v[0] += partial[(int64_t)p * ldp + e];      // slot order
    }
    store_halves<1>(hout + (int64_t)long_rows[i] * ldh + (int64_t)h * hsh + d, h2_off, v, hscale[0]);
}

template <int VEC, int LANES, int NCHUNK, int HB>
__global__ __launch_bounds__(kBlock) void spmm_dot_bcast_kernel(SpmmArgs a) {
    static_assert(LANES >= 16 && HB <= 4, "butterfly layout");
    constexpr int U = 4;
    const int lane = threadIdx.x % LANES;
    const int64_t item = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
    if (item >= a.n_items) return;
    const int4 it = a.items[item];
    int row = it.x, beg = it.y, end = it.z, slot = it.w;
    if constexpr (LANES == 64) {
        row = __builtin_amdgcn_readfirstlane(row);
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
        slot = __builtin_amdgcn_readfirstlane(slot);
    }
    const float* yb = a.y + (int64_t)row * a.ldy;
    int off[NCHUNK];
    bool act[NCHUNK];
    float acc[NCHUNK][VEC], yv[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * LANES + lane) * VEC;
        act[c] = e < a.D;
        off[c] = act[c] ? e : 0;
        vload<VEC>(yv[c], yb + off[c]);
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            acc[c][t] = 0.f;
            if (!act[c]) yv[c][t] = 0.f;
        }
    }
    for (int k0 = beg; k0 < end; k0 += LANES) {
        const int k = k0 + lane;
        int idx = 0, wp = 0;
        float wv[HB];
#pragma unroll
        for (int h = 0; h < HB; ++h) wv[h] = 0.f;
        if (k < end) {
            idx = a.indices[k];
            wp = a.wperm ? a.wperm[k] : k;
#pragma unroll
            for (int h = 0; h < HB; ++h) wv[h] = a.w[(int64_t)wp * HB + h];
        }
        const int cnt = min(LANES, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float p[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) p[q] = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);
                const int s = group_bcast<LANES>(idx, j);
                const float* px = a.x + (int64_t)s * a.ldx;
                float v[HB][NCHUNK][VEC];
#pragma unroll
                for (int h = 0; h < HB; ++h)
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[h][c], px + (int64_t)h * a.hsx + off[c]);
#pragma unroll
                for (int h = 0; h < HB; ++h) {
                    const float ww = i + u < cnt ? group_bcast<LANES>(wv[h], j) : 0.f;
                    float d = 0.f;
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                        for (int t = 0; t < VEC; ++t) {
                            acc[c][t] = fmaf(ww, v[h][c][t], acc[c][t]);
                            d = fmaf(v[h][c][t], yv[c][t], d);
                        }
                    p[u * 4 + h] = d;  // value slot = 4*u + h
                }
            }
            const float tot = transpose_reduce16<LANES>(p, lane);
            const int slot16 = lane & 15, u = slot16 >> 2, h = slot16 & 3;
            const int mywp = __shfl(wp, min(i + u, LANES - 1), LANES);
            if (lane < 16 && h < HB && i + u < cnt) a.dot_out[(int64_t)mywp * HB + h] = tot;
        }
    }
    float* ob = slot < 0 ? a.out + (int64_t)row * a.ldo : a.partial + (int64_t)slot * a.ldp;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
        if (act[c]) vstore<VEC>(ob + off[c], acc[c]);
}

template <int VEC, int HB, bool DOT>
static void dispatch_bcast(SpmmArgs& a, hipStream_t st) {
    const int L = (std::max(a.D, a.hout ? a.hpiece : 0) + VEC - 1) / VEC;       // (the halves form also writes the zero padding of a head's block)
#define BOT_BCAST(LN, NC)                                                                                                     \
    do {                                                                                                                      \
        const int64_t blocks = (a.n_items * LN + kBlock - 1) / kBlock;                                                        \
        if (blocks == 0) break;                                                                                               \
        set_kernel(DOT ? "bot::spmm_dot_bcast_kernel<%d,%d,%d,%d>" : "bot::spmm_bcast_kernel<%d,%d,%d,%d>", VEC, LN, NC, HB);        \
        if constexpr (DOT) hipLaunchKernelGGL((spmm_dot_bcast_kernel<VEC, LN, NC, HB>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a); \
        else hipLaunchKernelGGL((spmm_bcast_kernel<VEC, LN, NC, HB>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);        \
    } while (0)
    if (L <= 16) BOT_BCAST(16, 1);
    else if (L <= 32) BOT_BCAST(32, 1);
    else if (L <= 64) BOT_BCAST(64, 1);
    else if (L <= 128) BOT_BCAST(64, 2);
    else BOT_BCAST(64, 4);
#undef BOT_BCAST
}

template <bool DOT>
static int run_bcast(SpmmArgs& a, int vec, int H, hipStream_t st) {
#define BOT_BCAST_H(V)                                \
    switch (H) {                                      \
        case 1: dispatch_bcast<V, 1, DOT>(a, st); break; \
        case 2: dispatch_bcast<V, 2, DOT>(a, st); break; \
        case 3: dispatch_bcast<V, 3, DOT>(a, st); break; \
        default: dispatch_bcast<V, 4, DOT>(a, st); break; \
    }
    if (vec == 4) { BOT_BCAST_H(4) } else if (vec == 2) { BOT_BCAST_H(2) } else { BOT_BCAST_H(1) }
#undef BOT_BCAST_H
    return hip_status(DOT ? "spmm_dot_bcast launch" : "spmm_bcast launch");
}

template <int VEC, int LANES, int NCHUNK>
static void launch_spmm(const SpmmArgs& a, hipStream_t st) {
    const int64_t groups = a.n_items * a.H * a.n_tiles;
    const int64_t blocks = (groups * LANES + kBlock - 1) / kBlock;
    if (blocks == 0) return;
    set_kernel("bot::spmm_kernel<%d,%d,%d,%s>", VEC, LANES, NCHUNK, a.w ? "true" : "false");
    if (a.w) hipLaunchKernelGGL((spmm_kernel<VEC, LANES, NCHUNK, true>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
    else hipLaunchKernelGGL((spmm_kernel<VEC, LANES, NCHUNK, false>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

template <int VEC, int LANES, int NCHUNK>
static void launch_spmm_dot(const SpmmArgs& a, hipStream_t st) {
    const int64_t groups = a.n_items * a.H;
    const int64_t blocks = (groups * LANES + kBlock - 1) / kBlock;
    if (blocks == 0) return;
    set_kernel("bot::spmm_dot_kernel<%d,%d,%d>", VEC, LANES, NCHUNK);
    hipLaunchKernelGGL((spmm_dot_kernel<VEC, LANES, NCHUNK>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

template <int VEC>
static void dispatch_spmm_dot(SpmmArgs& a, hipStream_t st) {
    const int L = (a.D + VEC - 1) / VEC;
    if (L <= 8) launch_spmm_dot<VEC, 8, 1>(a, st);
    else if (L <= 16) launch_spmm_dot<VEC, 16, 1>(a, st);
    else if (L <= 32) launch_spmm_dot<VEC, 32, 1>(a, st);
    else if (L <= 64) launch_spmm_dot<VEC, 64, 1>(a, st);
    else if (L <= 128) launch_spmm_dot<VEC, 64, 2>(a, st);
    else if (L <= 192) launch_spmm_dot<VEC, 64, 3>(a, st);
    else launch_spmm_dot<VEC, 64, 4>(a, st);
}

// "rows" layout of spmm_dot (one wave per item, all heads) is taken whenever the shape fits it; BOT_SPMM_DOT_LAYOUT=heads
// forces the head-major kernel (measurements / debugging).
static bool spmm_dot_rows_wanted() {
    static const bool wanted = [] {  // read once per process, not per launch
        const char* e = getenv("BOT_SPMM_DOT_LAYOUT");
        return !(e && !strcmp(e, "heads"));
    }();
    return wanted;
}

// BOT_SPMM_LAYOUT=heads forces the head-major forward kernel (measurements); read once per process.
static bool spmm_rows_wanted() {
    static const bool wanted = [] {
        const char* e = getenv("BOT_SPMM_LAYOUT");
        return !(e && !strcmp(e, "heads"));
    }();
    return wanted;
}

template <int VEC, int HL, int NCHUNK, int CPH>
static void launch_spmm_dot_rows(const SpmmArgs& a, hipStream_t st) {
    const int64_t blocks = (a.n_items * 64 + kBlock - 1) / kBlock;
    if (blocks == 0) return;
    set_kernel("bot::spmm_dot_rows_kernel<%d,%d,%d,%d>", VEC, HL, NCHUNK, CPH);
    hipLaunchKernelGGL((spmm_dot_rows_kernel<VEC, HL, NCHUNK, CPH>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

// Returns false when the shape does not fit the rows layout: H == 1 (the head-major kernel is the same thing), head
// segments under 9 lanes, or more register chunks than a wave can hold.
template <int VEC>
static bool dispatch_spmm_dot_rows(const SpmmArgs& a, hipStream_t st) {
    if (a.H < 2 || !spmm_dot_rows_wanted()) return false;
    const int L = (a.D + VEC - 1) / VEC;
    if (L <= 8) return false;
    if (L > 64) {  // a head spans two chunks: H = 2 or 3 (ogbn-arxiv: 3 x 250 floats = 3 x 125 float2 lanes)
        if (L > 128 || VEC == 4 || a.H > 3) return false;
        if constexpr (VEC != 4) {
            if (a.H == 2) launch_spmm_dot_rows<VEC, 64, 4, 2>(a, st);
            else launch_spmm_dot_rows<VEC, 64, 6, 2>(a, st);
        }
        return true;
    }
    const int HL = L <= 16 ? 16 : (L <= 32 ? 32 : 64);
    const int nchunk = (a.H * HL + 63) / 64;
    if (nchunk > 4) return false;
#define BOT_ROWS(HLV)                                                      \
    do {                                                                   \
        if (nchunk == 1) launch_spmm_dot_rows<VEC, HLV, 1, 1>(a, st);      \
        else if (nchunk == 2) launch_spmm_dot_rows<VEC, HLV, 2, 1>(a, st); \
        else if (nchunk == 3) launch_spmm_dot_rows<VEC, HLV, 3, 1>(a, st); \
        else launch_spmm_dot_rows<VEC, HLV, 4, 1>(a, st);                  \
    } while (0)
    if (HL == 16) BOT_ROWS(16);
    else if (HL == 32) BOT_ROWS(32);
    else BOT_ROWS(64);
#undef BOT_ROWS
    return true;
}

template <int VEC, int HL, int NCHUNK, int CPH>
static void launch_spmm_rows(const SpmmArgs& a, hipStream_t st) {
    const int64_t blocks = (a.n_items * 64 + kBlock - 1) / kBlock;
    if (blocks == 0) return;
    set_kernel("bot::spmm_rows_kernel<%d,%d,%d,%d>", VEC, HL, NCHUNK, CPH);
    hipLaunchKernelGGL((spmm_rows_kernel<VEC, HL, NCHUNK, CPH>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

// All-heads layout for the weighted forward, taken whenever the shape fits (measured against the head-major kernel:
// S-arxiv H=3 D=250 1.24 -> 1.12 ms, H=2 D=250 0.81 -> 0.76, H=4 D=64 and H=3 D=128 equal; S-products H=4 D=120 45.6 -> 42.3 ms;
// S-proteins H=6 D=80 23.2 -> 19.3 ms).  BOT_SPMM_LAYOUT=heads forces the head-major kernel (measurements).
template <int VEC>
static bool dispatch_spmm_rows(const SpmmArgs& a, hipStream_t st) {
    if (a.H < 2 || a.w == nullptr || !spmm_rows_wanted()) return false;
    const int L = (a.D + VEC - 1) / VEC;
    if (L <= 8) return false;
    if (L > 64) {
        if (L > 128 || VEC == 4 || a.H > 3) return false;
        if constexpr (VEC != 4) {
            if (a.H == 2) launch_spmm_rows<VEC, 64, 4, 2>(a, st);
            else launch_spmm_rows<VEC, 64, 6, 2>(a, st);
        }
        return true;
    }
    const int HL = L <= 16 ? 16 : (L <= 32 ? 32 : 64);
    const int nchunk = (a.H * HL + 63) / 64;
    if (nchunk > 4) return false;
#define BOT_ROWS(HLV)                                                  \
    do {                                                               \
        if (nchunk == 1) launch_spmm_rows<VEC, HLV, 1, 1>(a, st);      \
        else if (nchunk == 2) launch_spmm_rows<VEC, HLV, 2, 1>(a, st); \
        else if (nchunk == 3) launch_spmm_rows<VEC, HLV, 3, 1>(a, st); \
        else launch_spmm_rows<VEC, HLV, 4, 1>(a, st);                  \
    } while (0)
    if (HL == 16) BOT_ROWS(16);
    else if (HL == 32) BOT_ROWS(32);
    else BOT_ROWS(64);
#undef BOT_ROWS
    return true;
}

// Which all-heads forward layout the calling thread's next bot_spmm_f32 calls take when both fit (bot_spmm_set_layout):
// 0 = head segments (8-byte loads at D = 250), 1 = flat 16-byte lanes.  Both give bitwise identical results; the flat layout is
// faster when the gathers are served by the L2 (graphs renumbered for locality: 0.68 -> 0.58 ms on S-arxiv-comm, 0.52 -> 0.37 ms
// with every source L2-resident) and ~4 % slower when they are fabric-bound (randomly numbered graphs: 1.08 -> 1.13 ms), so the
// host picks it per graph (bot_amd/_C.py: plans in XCD order = a numbering with locality).  BOT_SPMM_FLAT=0 / 1 overrides.
static thread_local int g_spmm_layout = 0;
static bool spmm_flat_wanted() {
    static const int env = [] {
        const char* e = getenv("BOT_SPMM_FLAT");
        return e ? (!strcmp(e, "0") ? 0 : 1) : -1;
    }();
    return env >= 0 ? env == 1 : g_spmm_layout == 1;
}

// Flat 16-byte lanes (spmm_flat_kernel): weighted, 2..4 heads whose width is not a multiple of 4, rows of H*D contiguous floats
// on a 16-byte aligned pitch that covers the tail float4, slabs of out / addend / partial with contiguous rows.
static bool dispatch_spmm_flat(const SpmmArgs& a, hipStream_t st) {
    const int F = a.H * a.D;
    if (!spmm_flat_wanted() || a.w == nullptr || a.H < 2 || a.H > 4 || a.D % 4 == 0 || a.D < 4 || F > 1024) return false;
    if (a.hsx != a.D || a.hso != a.D || (a.addend && a.hsa != a.D) || !aligned(a.x, 16) || a.ldx % 4 != 0 || a.ldx < (F + 3) / 4 * 4) return false;
    const int nchunk = (F + 255) / 256;
    const int64_t blocks = (a.n_items * 64 + kBlock - 1) / kBlock;
    if (blocks == 0) return true;
    set_kernel("bot::spmm_flat_kernel<%d,%d>", nchunk, a.H <= 3 ? 3 : 4);
#define BOT_FLAT(NC)                                                                                              \
    do {                                                                                                          \
        if (a.H <= 3) hipLaunchKernelGGL((spmm_flat_kernel<NC, 3>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a); \
        else hipLaunchKernelGGL((spmm_flat_kernel<NC, 4>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);         \
    } while (0)
    if (nchunk == 1) BOT_FLAT(1);
    else if (nchunk == 2) BOT_FLAT(2);
    else if (nchunk == 3) BOT_FLAT(3);
    else BOT_FLAT(4);
#undef BOT_FLAT
    return true;
}

template <int VEC>
static void dispatch_spmm(SpmmArgs& a, hipStream_t st) {
    const int L = (a.D + VEC - 1) / VEC;  // lanes needed for one head slab
    a.n_tiles = 1;
    if (L <= 8) launch_spmm<VEC, 8, 1>(a, st);
    else if (L <= 16) launch_spmm<VEC, 16, 1>(a, st);
    else if (L <= 32) launch_spmm<VEC, 32, 1>(a, st);
    else if (L <= 64) launch_spmm<VEC, 64, 1>(a, st);
    else if (L <= 128) launch_spmm<VEC, 64, 2>(a, st);
    else if (L <= 192) launch_spmm<VEC, 64, 3>(a, st);
    else {
        a.n_tiles = (L + 255) / 256;
        launch_spmm<VEC, 64, 4>(a, st);
    }
}

}  // namespace bot

extern "C" {

int64_t bot_spmm_workspace_floats(int64_t n_slots, int32_t H, int32_t D) { return n_slots * (int64_t)H * D; }

int bot_spmm_set_layout(int32_t layout) {
    BOT_REQUIRE(layout == 0 || layout == 1, BOT_E_RANGE, "spmm_set_layout: %d (0 = head segments, 1 = flat 16-byte lanes)", layout);
    bot::g_spmm_layout = layout;
    return 0;
}

int bot_spmm_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                 int64_t n_items, const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, const float* x,
                 int64_t ldx, int64_t hsx, const float* w, const int32_t* wperm, int32_t H, int32_t D, float* out,
                 int64_t ldo, int64_t hso, const float* addend, int64_t lda, int64_t hsa, float* partial,
                 bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && n_long >= 0, BOT_E_RANGE, "spmm: negative size");
    BOT_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, BOT_E_RANGE, "spmm: int32 index range exceeded");
    BOT_REQUIRE(H >= 1 && D >= 1, BOT_E_RANGE, "spmm: H=%d D=%d must be >= 1", H, D);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(items && x && out, BOT_E_NULL, "spmm: items/x/out is NULL");
    BOT_REQUIRE(nnz == 0 || indices, BOT_E_NULL, "spmm: indices is NULL");
    // (a plan may cover a subset of the rows: bot_amd/blocked.py runs the hub rows through this kernel)
    BOT_REQUIRE(n_long == 0 || (long_rows && long_ptr && partial), BOT_E_NULL, "spmm: long rows need long_rows/long_ptr/partial");
    BOT_REQUIRE(ldx >= (int64_t)(H - 1) * hsx + D && ldo >= (int64_t)(H - 1) * hso + D && hsx >= D && hso >= D, BOT_E_RANGE,
                "spmm: strides smaller than the slab (ldx=%lld hsx=%lld ldo=%lld hso=%lld H=%d D=%d)", (long long)ldx,
                (long long)hsx, (long long)ldo, (long long)hso, H, D);
    BOT_REQUIRE(aligned(x, 4) && aligned(out, 4) && aligned(items, 16), BOT_E_ALIGN, "spmm: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    BOT_REQUIRE(addend == nullptr || (hsa >= D && lda >= (int64_t)(H - 1) * hsa + D), BOT_E_RANGE, "spmm: addend strides smaller than the slab");
    SpmmArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, hsx, w, wperm, H, D, 1, out, ldo, hso, partial,
               (int64_t)H * D, nullptr, 0, 0, nullptr, addend, lda, hsa};
    const int vec = addend ? pick_vec(D, {ldx, hsx, ldo, hso, lda, hsa}, {x, out, partial, addend}) : pick_vec(D, {ldx, hsx, ldo, hso}, {x, out, partial});
    const bool rows = dispatch_spmm_flat(a, st) ||
                      (vec == 4 ? dispatch_spmm_rows<4>(a, st) : (vec == 2 ? dispatch_spmm_rows<2>(a, st) : dispatch_spmm_rows<1>(a, st)));
    if (!rows) {
        if (vec == 4) dispatch_spmm<4>(a, st);
        else if (vec == 2) dispatch_spmm<2>(a, st);
        else dispatch_spmm<1>(a, st);
    }
    if (int rc = hip_status("spmm launch")) return rc;
    if (n_long > 0) {
        const int64_t n = n_long * H * D;
        hipLaunchKernelGGL(spmm_combine_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, long_rows,
                           long_ptr, n_long, H, D, partial, (int64_t)H * D, out, ldo, hso, addend, lda, hsa);
        if (int rc = hip_status("spmm combine launch")) return rc;
    }
    return 0;
}

int bot_spmm_dot_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                     int64_t n_items, const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, const float* x,
                     int64_t ldx, int64_t hsx, const float* w, const int32_t* wperm, const float* y, int64_t ldy, int64_t hsy,
                     int32_t H, int32_t D, float* out, int64_t ldo, int64_t hso, float* dot_out, float* partial,
                     uint32_t* absmax_slots, bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && n_long >= 0, BOT_E_RANGE, "spmm_dot: negative size");
    BOT_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, BOT_E_RANGE, "spmm_dot: int32 index range exceeded");
    BOT_REQUIRE(H >= 1 && D >= 1, BOT_E_RANGE, "spmm_dot: H=%d D=%d must be >= 1", H, D);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(items && x && out && y && (nnz == 0 || (indices && w && dot_out)), BOT_E_NULL, "spmm_dot: NULL pointer");
    BOT_REQUIRE(n_long == 0 || (long_rows && long_ptr && partial), BOT_E_NULL, "spmm_dot: long rows need long_rows/long_ptr/partial");
    BOT_REQUIRE(hsx >= D && hso >= D && hsy >= D && ldx >= (int64_t)(H - 1) * hsx + D && ldo >= (int64_t)(H - 1) * hso + D &&
                    ldy >= (int64_t)(H - 1) * hsy + D, BOT_E_RANGE, "spmm_dot: strides smaller than the slab");
    hipStream_t st = (hipStream_t)stream;
    SpmmArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, hsx, w, wperm, H, D, 1, out, ldo, hso, partial,
               (int64_t)H * D, y, ldy, hsy, dot_out, nullptr, 0, 0};
    const int vec = pick_vec(D, {ldx, hsx, ldo, hso, ldy, hsy}, {x, out, partial, y});
    BOT_REQUIRE(D <= vec * 256, BOT_E_RANGE, "spmm_dot: D=%d exceeds the %d floats one launch tile covers (use bot_spmm_f32 + bot_sddmm_dot_f32)",
                D, vec * 256);
    a.absmax = absmax_slots;             // by-product of the all-heads kernel and of the long rows' combine pass
    const bool rows = vec == 4 ? dispatch_spmm_dot_rows<4>(a, st) : (vec == 2 ? dispatch_spmm_dot_rows<2>(a, st) : dispatch_spmm_dot_rows<1>(a, st));
    if (!rows) {
        if (vec == 4) dispatch_spmm_dot<4>(a, st);
        else if (vec == 2) dispatch_spmm_dot<2>(a, st);
        else dispatch_spmm_dot<1>(a, st);
    }
    if (int rc = hip_status("spmm_dot launch")) return rc;
    if (n_long > 0) {
        const int64_t n = n_long * H * D;
        hipLaunchKernelGGL(spmm_combine_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, long_rows,
                           long_ptr, n_long, H, D, partial, (int64_t)H * D, out, ldo, hso, (const float*)nullptr, (int64_t)0, (int64_t)0,
                           rows ? absmax_slots : (uint32_t*)nullptr);
        if (int rc = hip_status("spmm_dot combine launch")) return rc;
    }
    if (absmax_slots && !rows) {         // the head-major kernel has no by-product form: one pass per head slab over the result
        for (int h = 0; h < H; ++h) launch_absmax_slots(out + (int64_t)h * hso, ldo, n_rows, D, absmax_slots, st);
        if (int rc = hip_status("spmm_dot absmax launch")) return rc;
    }
    return 0;
}

int bot_spmm_bcast_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                       int64_t n_items, const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, const float* x,
                       int64_t ldx, const float* w, const int32_t* wperm, int32_t H, int32_t D, float* out, int64_t ldo,
                       int64_t hso, float* partial, bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && n_long >= 0 && nnz < INT32_MAX, BOT_E_RANGE, "spmm_bcast: bad size");
    BOT_REQUIRE(H >= 1 && H <= 4 && D >= 1 && D <= 1024, BOT_E_RANGE, "spmm_bcast: H=%d (1..4) D=%d (1..1024)", H, D);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(items && x && out && (nnz == 0 || (indices && w)), BOT_E_NULL, "spmm_bcast: NULL pointer");
    BOT_REQUIRE(n_long == 0 || (long_rows && long_ptr && partial), BOT_E_NULL, "spmm_bcast: long rows need long_rows/long_ptr/partial");
    BOT_REQUIRE(ldx >= D && (hso >= D || H == 1) && ldo >= D, BOT_E_RANGE, "spmm_bcast: strides smaller than the slab");
    hipStream_t st = (hipStream_t)stream;
    SpmmArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, 0, w, wperm, H, D, 1, out, ldo, hso, partial,
               (int64_t)H * D, nullptr, 0, 0, nullptr, nullptr, 0, 0};
    const int vec = pick_vec(D, {ldx, ldo, hso}, {x, out, partial});
    BOT_REQUIRE(D <= vec * 256, BOT_E_RANGE, "spmm_bcast: D=%d exceeds one launch tile", D);
    if (int rc = run_bcast<false>(a, vec, H, st)) return rc;
    if (n_long > 0) {
        const int64_t n = n_long * H * D;
        hipLaunchKernelGGL(spmm_combine_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, long_rows,
                           long_ptr, n_long, H, D, partial, (int64_t)H * D, out, ldo, hso, (const float*)nullptr, (int64_t)0, (int64_t)0);
        if (int rc = hip_status("spmm_bcast combine launch")) return rc;
    }
    return 0;
}

int bot_spmm_bcast_halves_f16(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items, int64_t n_items,
                              const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, const float* x, int64_t ldx, const float* w,
                              const int32_t* wperm, int32_t H, int32_t D, const float* hscale, uint16_t* hout, int64_t ldh, int64_t hsh, int32_t h2_off,
                              int32_t hpiece, float* partial, bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && n_long >= 0 && nnz < INT32_MAX, BOT_E_RANGE, "spmm_bcast_halves: bad size");
    BOT_REQUIRE(H >= 1 && H <= 4 && D >= 1 && D <= 1024, BOT_E_RANGE, "spmm_bcast_halves: H=%d (1..4) D=%d (1..1024)", H, D);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(items && x && hout && hscale && (nnz == 0 || (indices && w)), BOT_E_NULL, "spmm_bcast_halves: NULL pointer");
    BOT_REQUIRE(n_long == 0 || (long_rows && long_ptr && partial), BOT_E_NULL, "spmm_bcast_halves: long rows need long_rows/long_ptr/partial");
    BOT_REQUIRE(ldx >= D && hpiece >= D && hpiece % 4 == 0 && (hsh >= hpiece || H == 1) && h2_off % 4 == 0 && hsh % 4 == 0 && ldh % 4 == 0 &&
                    h2_off >= (int64_t)(H - 1) * hsh + hpiece && ldh >= h2_off + (int64_t)(H - 1) * hsh + hpiece && aligned(hout, 8),
                BOT_E_RANGE, "spmm_bcast_halves: D=%d hpiece=%d hsh=%lld h2_off=%d ldh=%lld", D, hpiece, (long long)hsh, h2_off, (long long)ldh);
    hipStream_t st = (hipStream_t)stream;
    SpmmArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, 0, w, wperm, H, D, 1, nullptr, 0, 0, partial,
               (int64_t)H * D, nullptr, 0, 0, nullptr, nullptr, 0, 0};
    a.hout = reinterpret_cast<__half*>(hout), a.ldh = ldh, a.hsh = hsh, a.h2_off = h2_off, a.hpiece = hpiece, a.hscale = hscale;
    const int vec = pick_vec(D, {ldx}, {x, partial});
    BOT_REQUIRE(hpiece <= vec * 256, BOT_E_RANGE, "spmm_bcast_halves: hpiece=%d exceeds one launch tile", hpiece);
    // (the lane layout must cover the padded block: dispatch on hpiece, the kernels mask the loads with D)
    a.D = D;
    if (int rc = run_bcast<false>(a, vec, H, st)) return rc;
    if (n_long > 0) {
        const int64_t n = n_long * H * hpiece;
        hipLaunchKernelGGL(spmm_combine_halves_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, long_rows, long_ptr, n_long, H, D,
                           (const float*)partial, (int64_t)H * D, a.hout, ldh, hsh, h2_off, hpiece, hscale);
        if (int rc = hip_status("spmm_bcast_halves combine launch")) return rc;
    }
    return 0;
}

// Would bot_spmm_dot_halves_f16 take these operands (the all-heads layout covers the shape, the operand's columns are aligned to the
// lanes' stores)?  1 / 0; no launch.
int bot_spmm_dot_halves_fits(const float* x, int64_t ldx, int64_t hsx, const float* y, int64_t ldy, int64_t hsy, int32_t H, int32_t D, const uint16_t* hout,
                             int64_t ldh, int64_t hsh, int32_t h2_off) {
    using namespace bot;
    if (H < 2 || D < 1 || !spmm_dot_rows_wanted()) return 0;
    const int vec = pick_vec(D, {ldx, hsx, ldy, hsy}, {x, y});
    if (!(hsh >= D && h2_off >= (int64_t)(H - 1) * hsh + D && ldh >= h2_off + (int64_t)(H - 1) * hsh + D && hsh % vec == 0 && h2_off % vec == 0 && ldh % vec == 0 &&
          aligned(hout, 2 * vec) && D <= vec * 256))
        return 0;
    const int L = (D + vec - 1) / vec;
    if (L <= 8) return 0;
    if (L > 64) return (L <= 128 && vec != 4 && H <= 3) ? 1 : 0;
    const int HL = L <= 16 ? 16 : (L <= 32 ? 32 : 64);
    return (H * HL + 63) / 64 <= 4 ? 1 : 0;
}

// bot_spmm_dot_f32 with `out` written as a LEFT halves operand [h1 | 2^11 h2] of hscale[0] * out (halves.hip) instead of fp32: the
// gradient of the projected features goes straight into the operand of the layer's two backward GEMMs (include/bot_gnn.h).  The all-heads
// ("rows") layout only; BOT_E_RANGE for shapes it does not cover (the caller keeps the fp32 form + a split pass).
int bot_spmm_dot_halves_f16(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                            int64_t n_items, const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, const float* x,
                            int64_t ldx, int64_t hsx, const float* w, const int32_t* wperm, const float* y, int64_t ldy, int64_t hsy,
                            int32_t H, int32_t D, const float* hscale, uint16_t* hout, int64_t ldh, int64_t hsh, int32_t h2_off, float* dot_out,
                            float* partial, bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && n_long >= 0 && nnz < INT32_MAX && n_rows < INT32_MAX, BOT_E_RANGE, "spmm_dot_halves: bad size");
    BOT_REQUIRE(H >= 2 && D >= 1, BOT_E_RANGE, "spmm_dot_halves: H=%d D=%d (the all-heads layout needs H >= 2)", H, D);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(items && x && hout && hscale && y && (nnz == 0 || (indices && w && dot_out)), BOT_E_NULL, "spmm_dot_halves: NULL pointer");
    BOT_REQUIRE(n_long == 0 || (long_rows && long_ptr && partial), BOT_E_NULL, "spmm_dot_halves: long rows need long_rows/long_ptr/partial");
    BOT_REQUIRE(hsx >= D && hsy >= D && ldx >= (int64_t)(H - 1) * hsx + D && ldy >= (int64_t)(H - 1) * hsy + D, BOT_E_RANGE,
                "spmm_dot_halves: strides smaller than the slab");
    hipStream_t st = (hipStream_t)stream;
    SpmmArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, hsx, w, wperm, H, D, 1, nullptr, 0, 0, partial,
               (int64_t)H * D, y, ldy, hsy, dot_out, nullptr, 0, 0};
    const int vec = pick_vec(D, {ldx, hsx, ldy, hsy}, {x, partial, y});
    // a lane stores `vec` consecutive halves of each half with one 2 * vec-byte store: the operand's columns must be aligned to it
    BOT_REQUIRE(hsh >= D && h2_off >= (int64_t)(H - 1) * hsh + D && ldh >= h2_off + (int64_t)(H - 1) * hsh + D && hsh % vec == 0 && h2_off % vec == 0 &&
                    ldh % vec == 0 && aligned(hout, 2 * vec), BOT_E_RANGE, "spmm_dot_halves: D=%d hsh=%lld h2_off=%d ldh=%lld vec=%d", D, (long long)hsh, h2_off,
                (long long)ldh, vec);
    BOT_REQUIRE(D <= vec * 256, BOT_E_RANGE, "spmm_dot_halves: D=%d exceeds one launch tile", D);
    a.hout = reinterpret_cast<__half*>(hout), a.ldh = ldh, a.hsh = hsh, a.h2_off = h2_off, a.hpiece = D, a.hscale = hscale;
    const bool rows = vec == 4 ? dispatch_spmm_dot_rows<4>(a, st) : (vec == 2 ? dispatch_spmm_dot_rows<2>(a, st) : dispatch_spmm_dot_rows<1>(a, st));
    BOT_REQUIRE(rows, BOT_E_RANGE, "spmm_dot_halves: H=%d D=%d does not fit the all-heads layout", H, D);
    if (int rc = hip_status("spmm_dot_halves launch")) return rc;
    if (n_long > 0) {
        const int64_t n = n_long * H * D;
        hipLaunchKernelGGL(spmm_combine_halves_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, long_rows, long_ptr, n_long, H, D,
                           (const float*)partial, (int64_t)H * D, a.hout, ldh, hsh, h2_off, D, hscale);
        if (int rc = hip_status("spmm_dot_halves combine launch")) return rc;
    }
    return 0;
}

int bot_spmm_dot_bcast_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                           int64_t n_items, const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, const float* x,
                           int64_t ldx, int64_t hsx, const float* w, const int32_t* wperm, const float* y, int64_t ldy,
                           int32_t H, int32_t D, float* out, int64_t ldo, float* dot_out, float* partial,
                           bot_stream_t stream) {
    using namespace bot;
    (void)indptr;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && n_long >= 0 && nnz < INT32_MAX, BOT_E_RANGE, "spmm_dot_bcast: bad size");
    BOT_REQUIRE(H >= 1 && H <= 4 && D >= 1 && D <= 1024, BOT_E_RANGE, "spmm_dot_bcast: H=%d (1..4) D=%d (1..1024)", H, D);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(items && x && out && y && (nnz == 0 || (indices && w && dot_out)), BOT_E_NULL, "spmm_dot_bcast: NULL pointer");
    BOT_REQUIRE(n_long == 0 || (long_rows && long_ptr && partial), BOT_E_NULL, "spmm_dot_bcast: long rows need long_rows/long_ptr/partial");
    BOT_REQUIRE(ldx >= D && ldy >= D && ldo >= D && (hsx >= D || H == 1), BOT_E_RANGE, "spmm_dot_bcast: strides smaller than the slab");
    hipStream_t st = (hipStream_t)stream;
    SpmmArgs a{indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, hsx, w, wperm, H, D, 1, out, ldo, D, partial,
               (int64_t)D, y, ldy, 0, dot_out, nullptr, 0, 0};
    const int vec = pick_vec(D, {ldx, hsx, ldo, ldy}, {x, out, partial, y});
    BOT_REQUIRE(D <= vec * 256, BOT_E_RANGE, "spmm_dot_bcast: D=%d exceeds one launch tile", D);
    if (int rc = run_bcast<true>(a, vec, H, st)) return rc;
    if (n_long > 0) {
        const int64_t n = n_long * D;
        hipLaunchKernelGGL(spmm_combine_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, long_rows,
                           long_ptr, n_long, 1, D, partial, (int64_t)D, out, ldo, (int64_t)D, (const float*)nullptr, (int64_t)0, (int64_t)0);
        if (int rc = hip_status("spmm_dot_bcast combine launch")) return rc;
    }
    return 0;
}

}  // extern "C"
