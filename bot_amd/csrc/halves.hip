// fp32 matrices as pairs of fp16 halves, the operand format of bot_gemm_halves_f32 (gemm.cpp).
//
//   x = (h1 + h2) / s,   h1 = fp16(s x),   h2 = fp16(s x - h1)      (round to nearest; s a power of two)
//
// h1 + h2 reproduces s x to within 2^-23 |s x| (two 11-bit significands, the second one signed), the size of fp32's own
// rounding, so a product sum_k a_k b_k evaluated as  sum a1 b1 + a1 b2 + a2 b1  with fp32 accumulation differs from an fp32
// GEMM by the dropped a2 b2 term (2^-22 relative per product) — measured against fp64 it is as close as hipBLASLt's fp32
// GEMM (tools/exp_split_gemm*.py).  The fp16 MFMA rate of gfx950 is 16x the fp32 one, so three fp16 products are ~3x faster.
//
//   halves_scale : s = 2^(14 - ceil(log2 max|x|))  puts the largest entry in (2^13, 2^14] (fp16 max 2^16; entries below
//                  2^-38 of the largest flush to zero) — computed on the device, never read by the host
//   halves_split : out[r] = [h1 | h1 | 2^11 h2] (order 0, left operands) or [h1 | h2 | 2^-11 h1] (order 1, right operands), each
//                  piece zero-padded to `piece` columns, so that ONE GEMM over the concatenated axis forms the three terms
//
// Dynamic range (round 3).  One scale per matrix means an entry 2^-17 of the largest has its SECOND half below fp16's normal
// range (|h2| <= 2^-11 |h1|), and rows of a left operand that small kept only 11-15 bits (measured: rows 2^-24 of the largest
// 1e-4 off relative to their own size, tools/exp_halves_range.py).  The left operand therefore stores 2^11 h2 — as large as h1,
// so it leaves the normal range only where h1 does — and the right operand pairs it with 2^-11 h1 (weights are the right
// operands: entries within 2^-3..2^14 after scaling stay exact, smaller ones contribute below 2^-39 of the row's largest
// product).  Rows of the left operand down to 2^-28 of the matrix maximum keep 22 bits; in ABSOLUTE terms every entry is
// reproduced to 2^-38 of the matrix maximum or better, whatever its size.
#include <hip/hip_fp16.h>

#include "common.h"

namespace bot {

constexpr int kMaxBlocks = 1024;

__global__ __launch_bounds__(kBlock) void absmax_partial_kernel(const float* x, int64_t ldx, int64_t n, int32_t F, float* part) {
    __shared__ float lds[kBlock / kWave];
    float m = 0.f;
    const int64_t total = n * (int64_t)F;
    if (ldx == F && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // contiguous and 16-byte aligned: float4 lanes, four independent loads in flight (round 3: the scalar loop ran at 3 TB/s)
        const int64_t nq = total >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const int64_t stride = (int64_t)gridDim.x * kBlock;
        int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        for (; i + 3 * stride < nq; i += 4 * stride) {
            const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                               fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
        }
        for (; i < nq; i += stride) {
            const float4 a = x4[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
        for (int64_t j = (nq << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += stride) m = fmaxf(m, fabsf(x[j]));
    } else if (ldx == F) {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) m = fmaxf(m, fabsf(x[i]));
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
            const int64_t r = i / F;
            m = fmaxf(m, fabsf(x[r * ldx + (i - r * F)]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / kWave; ++w) m = fmaxf(m, lds[w]);
        part[blockIdx.x] = m;
    }
}

// fmaxf drops NaNs, an infinite entry gives s = 1: non-finite inputs then reach the GEMM as fp16 inf/NaN and poison the
// product exactly as they would in fp32.
__global__ __launch_bounds__(kWave) void halves_scale_kernel(const float* part, int nblk, float* scale) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nblk; i += kWave) m = fmaxf(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) {
        float s = 1.f;
        if (m > 0.f && m < INFINITY) {
            int e;
            const float f = frexpf(m, &e);          // m = f * 2^e, f in [0.5, 1)
            if (f == 0.5f) e -= 1;                  // exact power of two: ceil(log2 m) = e - 1
            s = ldexpf(1.f, min(60, 14 - e));   // clamped for tiny tensors: 1/s and the product of two of them stay normal fp32 numbers
        }
        scale[0] = s;
        scale[1] = 1.f / s;
    }
}

// `width` columns of each piece are written starting at `out` (F from x, zeros behind them); the three pieces are `piece` apart
template <int ORDER>
__global__ __launch_bounds__(128) void halves_split_kernel(const float* x, int64_t ldx, int32_t F, const float* scale, __half* out,
                                                           int64_t ldo, int32_t piece, bool wide, int32_t width) {
    const int64_t r = blockIdx.x;
    const int c = (blockIdx.y * 128 + threadIdx.x) * 2;
    if (c >= width) return;
    const float s = scale ? scale[0] : 1.f;
    float v0 = 0.f, v1 = 0.f;
    const float* xr = x + r * ldx;
    if (c + 1 < F) {
        if (wide) {
            const float2 v = *reinterpret_cast<const float2*>(xr + c);
            v0 = v.x * s, v1 = v.y * s;
        } else {
            v0 = xr[c] * s, v1 = xr[c + 1] * s;
        }
    } else if (c < F) {
        v0 = xr[c] * s;
    }
    const __half a0 = __float2half_rn(v0), a1 = __float2half_rn(v1);
    const float r0 = v0 - __half2float(a0), r1 = v1 - __half2float(a1);        // exact in fp32
    __half* o = out + r * ldo + c;
    const __half2 hi = __halves2half2(a0, a1);
    *reinterpret_cast<__half2*>(o) = hi;
    if (ORDER == 0) {   // [h1 | h1 | 2^11 h2]
        *reinterpret_cast<__half2*>(o + piece) = hi;
        *reinterpret_cast<__half2*>(o + 2 * (int64_t)piece) = __halves2half2(__float2half_rn(r0 * kHalvesShift), __float2half_rn(r1 * kHalvesShift));
    } else if (ORDER == 2) {   // [h1 | 2^11 h2]: a left operand without the duplicate piece (read by csrc/halves3.hip only)
        *reinterpret_cast<__half2*>(o + piece) = __halves2half2(__float2half_rn(r0 * kHalvesShift), __float2half_rn(r1 * kHalvesShift));
    } else {            // [h1 | h2 | 2^-11 h1]
        *reinterpret_cast<__half2*>(o + piece) = __halves2half2(__float2half_rn(r0), __float2half_rn(r1));
        *reinterpret_cast<__half2*>(o + 2 * (int64_t)piece) =
            __halves2half2(__float2half_rn(__half2float(a0) * (1.f / kHalvesShift)), __float2half_rn(__half2float(a1) * (1.f / kHalvesShift)));
    }
}

// A RIGHT operand in FRAGMENT-MAJOR layout (order 3, read by gemm_halves3_nt64_kernel only): the 16 bytes lane l of an MFMA operand fragment
// holds - row 16 t + (l & 15), columns 32 s + 8 (l >> 4) .. + 7 of 16-row tile t and k-step s - stored at halves ((t T + s) 64 + l) 8 .. + 7
// (T = piece / 32 k-steps), so that the 64 lanes of a fragment load read ONE contiguous KB; h1 in the first region, h2 in a second region
// ceil(n / 16) T 512 halves behind it (the third piece, 2^-11 h1, is formed in registers).  Rows beyond n and columns beyond F are zeros.
__global__ __launch_bounds__(128) void halves_split_frag_kernel(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, __half* out,
                                                                int32_t piece, int64_t region) {
    const int64_t r = blockIdx.x;
    const int c = (blockIdx.y * 128 + threadIdx.x) * 2;
    if (c >= piece) return;
    const float s = scale ? scale[0] : 1.f;
    float v0 = 0.f, v1 = 0.f;
    if (r < n) {
        const float* xr = x + r * ldx;
        if (c < F) v0 = xr[c] * s;
        if (c + 1 < F) v1 = xr[c + 1] * s;
    }
    const __half a0 = __float2half_rn(v0), a1 = __float2half_rn(v1);
    const float r0 = v0 - __half2float(a0), r1 = v1 - __half2float(a1);
    const int T = piece >> 5;
    const int64_t o = ((((r >> 4) * T + (c >> 5)) * 64) + ((r & 15) + 16 * ((c & 31) >> 3))) * 8 + (c & 7);
    *reinterpret_cast<__half2*>(out + o) = __halves2half2(a0, a1);
    *reinterpret_cast<__half2*>(out + region + o) = __halves2half2(__float2half_rn(r0), __float2half_rn(r1));
}

// x [n, H * D] -> a LEFT operand without the duplicate piece whose head blocks are DP >= D columns wide (zeros behind a head's D columns):
// out[r, h DP + j] = h1, out[r, h2_off + h DP + j] = 2^11 h2 of scale[0] * x[r, h D + j].  One pass over the H heads (instead of H calls of
// halves_split_cols on column slices): the gradient operand of the aggregate-first GAT layer (fused.py:_GATHiddenAggFirst.backward).
__global__ __launch_bounds__(128) void halves_split_heads_kernel(const float* x, int64_t ldx, int32_t H, int32_t D, const float* scale, __half* out,
                                                                 int64_t ldo, int32_t h2_off, int32_t DP, bool wide) {
    const int64_t r = blockIdx.x;
    const int c = (blockIdx.y * 128 + threadIdx.x) * 2;          // destination column pair
    if (c >= H * DP) return;
    const int h = c / DP, j = c - h * DP;
    const float s = scale ? scale[0] : 1.f;
    float v0 = 0.f, v1 = 0.f;
    const float* xr = x + r * ldx + (int64_t)h * D + j;
    if (j + 1 < D) {
        if (wide) {
            const float2 v = *reinterpret_cast<const float2*>(xr);
            v0 = v.x * s, v1 = v.y * s;
        } else {
            v0 = xr[0] * s, v1 = xr[1] * s;
        }
    } else if (j < D) {
        v0 = xr[0] * s;
    }
    const __half a0 = __float2half_rn(v0), a1 = __float2half_rn(v1);
    const float r0 = v0 - __half2float(a0), r1 = v1 - __half2float(a1);        // exact in fp32
    __half* o = out + r * ldo + c;
    *reinterpret_cast<__half2*>(o) = __halves2half2(a0, a1);
    *reinterpret_cast<__half2*>(o + h2_off) = __halves2half2(__float2half_rn(r0 * kHalvesShift), __float2half_rn(r1 * kHalvesShift));
}

// halves_scale_kernel on `mult` x the maximum (a BOUND assembled on the host side of the maximum: max|dx| x the largest row sum of the edge
// weights), optionally capped at cap[0] x cap_ratio (a second scale of one operand may be finer than the first only by a bounded factor:
// the GEMM multiplies its accumulators by their ratio, csrc/halves3.hip `scale_a2`)
__global__ __launch_bounds__(kWave) void halves_scale2_kernel(const float* part, int nblk, float mult, const float* cap, float cap_ratio, float* scale) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nblk; i += kWave) m = fmaxf(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) {
        m *= mult;
        float s = 1.f;
        if (m > 0.f && m < INFINITY) {
            int e;
            const float f = frexpf(m, &e);
            if (f == 0.5f) e -= 1;
            s = ldexpf(1.f, min(60, 14 - e));
        }
        if (cap) s = fminf(s, cap[0] * cap_ratio);
        scale[0] = s;
        scale[1] = 1.f / s;
    }
}

// Column segments of a LEFT halves operand written from small fp32 sources (or as zeros): the attention columns and the padding columns
// of a layer's gradient operand, whose big column blocks their producers wrote (bot_halves_tail_f16).
constexpr int kTailSegs = 8;
struct TailArgs {
    int col[kTailSegs], width[kTailSegs];
    const float* src[kTailSegs];        // nullptr: zeros
    int64_t ld[kTailSegs];
    int n_seg, total;
};
__global__ __launch_bounds__(kBlock) void halves_tail_kernel(TailArgs t, int64_t n, const float* scale, __half* out, int64_t ldo, int32_t h2_off) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= n * t.total) return;
    const int64_t r = gid / t.total;
    int j = (int)(gid - r * t.total), g = 0;
    while (j >= t.width[g]) j -= t.width[g], ++g;       // (g < n_seg: j < total)
    const float v = t.src[g] ? t.src[g][r * t.ld[g] + j] * scale[0] : 0.f;
    const __half a = __float2half_rn(v);
    __half* o = out + r * ldo + t.col[g] + j;
    o[0] = a;
    o[h2_off] = __float2half_rn((v - __half2float(a)) * kHalvesShift);
}

void launch_halves_scale(const float* part, int n, float* scale, hipStream_t st) {
    hipLaunchKernelGGL(halves_scale_kernel, dim3(1), dim3(kWave), 0, st, part, n, scale);
}

// max|x| of a strided matrix into the by-product slots (common.h): the same reduction as absmax_partial_kernel, published with the
// integer atomic instead of a partials array
__global__ __launch_bounds__(kBlock) void absmax_slots_kernel(const float* x, int64_t ldx, int64_t n, int32_t F, uint32_t* slots) {
    float m = 0.f;
    const int64_t total = n * (int64_t)F;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if (ldx == F && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {      // contiguous, 16-byte aligned: float4 lanes, four loads in flight
        const int64_t nq = total >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        for (; i + 3 * stride < nq; i += 4 * stride) {
            const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                               fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
        }
        for (; i < nq; i += stride) {
            const float4 a = x4[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
        for (int64_t j = (nq << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x; j < total; j += stride) m = fmaxf(m, fabsf(x[j]));
    } else {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
            const int64_t r = i / F;
            m = fmaxf(m, fabsf(x[r * ldx + (i - r * F)]));
        }
    }
    absmax_publish(wave_absmax(m), slots);
}

// The weight gradient x^T d of two LEFT-layout halves operands arrives as chunked partial products (bot_amd/gemm.py:tn):
//   a[s] = x1_s^T [d1_s | 2^11 d2_s]  ([K, 2 PP] per row chunk s),   b[s] = (2^11 x2_s)^T d1_s  ([K, PP]),  optional remainders ra / rb.
// out[k, p] = sum_s a[s][k][p] + (sum_s a[s][k][PP + p] + sum_s b[s][k][p]) * 2^-11 — the chunk sums in chunk order, eight loads in flight.
__global__ __launch_bounds__(kBlock) void tn_combine_kernel(const float* a, const float* b, int32_t S, int32_t K, int32_t PP, int32_t P,
                                                           const float* ra, const float* rb, float* out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= (int64_t)K * P) return;
    const int k = (int)(idx / P), p = (int)(idx - (int64_t)k * P);
    const int64_t sa = (int64_t)K * 2 * PP, sb = (int64_t)K * PP;
    const float* a1 = a + (int64_t)k * 2 * PP + p;
    const float* a2 = a1 + PP;
    const float* b1 = b + (int64_t)k * PP + p;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        float v1[4], v2[4], v3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v1[j] = a1[(s + j) * sa], v2[j] = a2[(s + j) * sa], v3[j] = b1[(s + j) * sb];
#pragma unroll
        for (int j = 0; j < 4; ++j) s1 += v1[j], s2 += v2[j], s3 += v3[j];
    }
    for (; s < S; ++s) s1 += a1[s * sa], s2 += a2[s * sa], s3 += b1[s * sb];
    if (ra) s1 += ra[(int64_t)k * 2 * PP + p], s2 += ra[(int64_t)k * 2 * PP + PP + p], s3 += rb[(int64_t)k * PP + p];
    out[(int64_t)k * ldo + p] = s1 + (s2 + s3) * (1.f / kHalvesShift);
}

void launch_absmax_slots(const float* x, int64_t ldx, int64_t n, int32_t F, uint32_t* slots, hipStream_t st) {
    const int64_t total = n * (int64_t)F;
    if (total <= 0) return;
    const int blocks = (int)max((int64_t)1, min((int64_t)kMaxBlocks, (total + kBlock * 8 - 1) / (kBlock * 8)));
    hipLaunchKernelGGL(absmax_slots_kernel, dim3(blocks), dim3(kBlock), 0, st, x, ldx, n, F, slots);
}

}  // namespace bot

extern "C" {

int64_t bot_halves_workspace_floats(void) { return bot::kMaxBlocks; }

int bot_halves_scale_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float* scale, float* workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && F >= 1 && ldx >= F, BOT_E_RANGE, "halves_scale: n=%lld F=%d ldx=%lld", (long long)n, F, (long long)ldx);
    BOT_REQUIRE(scale && workspace && (x || n == 0), BOT_E_NULL, "halves_scale: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n * (int64_t)F;
    const int blocks = (int)max((int64_t)1, min((int64_t)kMaxBlocks, (total + kBlock * 8 - 1) / (kBlock * 8)));
    hipLaunchKernelGGL(absmax_partial_kernel, dim3(blocks), dim3(kBlock), 0, st, x, ldx, n, F, workspace);
    launch_halves_scale(workspace, blocks, scale, st);
    return hip_status("halves_scale launch");
}

int32_t bot_absmax_slots(void) { return bot::kAbsmaxSlots; }

int bot_absmax_slots_f32(const float* x, int64_t ldx, int64_t n, int32_t F, uint32_t* slots, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && F >= 1 && ldx >= F, BOT_E_RANGE, "absmax_slots: n=%lld F=%d ldx=%lld", (long long)n, F, (long long)ldx);
    BOT_REQUIRE(slots && (x || n == 0), BOT_E_NULL, "absmax_slots: NULL pointer");
    launch_absmax_slots(x, ldx, n, F, slots, (hipStream_t)stream);
    return hip_status("absmax_slots launch");
}

int bot_halves_scale_from_slots_f32(const uint32_t* slots, float* scale, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(slots && scale, BOT_E_NULL, "halves_scale_from_slots: NULL pointer");
    // a non-negative float and its bit pattern are the same word: the slots ARE an array of partial maxima
    launch_halves_scale(reinterpret_cast<const float*>(slots), kAbsmaxSlots, scale, (hipStream_t)stream);
    return hip_status("halves_scale_from_slots launch");
}

int bot_halves_scale_from_slots2_f32(const uint32_t* slots, float mult, const float* cap_scale, float cap_ratio, float* scale, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(slots && scale, BOT_E_NULL, "halves_scale_from_slots2: NULL pointer");
    BOT_REQUIRE(mult > 0.f && mult < INFINITY && (cap_scale == nullptr || cap_ratio > 0.f), BOT_E_RANGE, "halves_scale_from_slots2: mult=%g cap_ratio=%g", mult, cap_ratio);
    hipLaunchKernelGGL(halves_scale2_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, reinterpret_cast<const float*>(slots), kAbsmaxSlots, mult, cap_scale,
                       cap_ratio, scale);
    return hip_status("halves_scale_from_slots2 launch");
}

int bot_halves_tail_f16(int64_t n, int32_t n_seg, const int64_t* seg_cols, const float* const* srcs, const int64_t* src_ld, const float* scale, uint16_t* out,
                        int64_t ldo, int32_t h2_off, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && n_seg >= 1 && n_seg <= kTailSegs && seg_cols && srcs && src_ld && scale && out, BOT_E_NULL, "halves_tail: 1 .. %d segments, no NULL pointer",
                kTailSegs);
    TailArgs t{};
    t.n_seg = n_seg, t.total = 0;
    for (int g = 0; g < n_seg; ++g) {
        const int64_t col = seg_cols[2 * g], w = seg_cols[2 * g + 1];
        BOT_REQUIRE(col >= 0 && w >= 1 && col + w <= h2_off && h2_off + col + w <= ldo && (srcs[g] == nullptr || src_ld[g] >= w), BOT_E_RANGE,
                    "halves_tail: segment %d: col=%lld width=%lld (h2_off=%d ldo=%lld)", g, (long long)col, (long long)w, h2_off, (long long)ldo);
        t.col[g] = (int)col, t.width[g] = (int)w, t.src[g] = srcs[g], t.ld[g] = src_ld[g];
        t.total += (int)w;
    }
    if (n == 0) return 0;
    const int64_t total = n * t.total;
    hipLaunchKernelGGL(halves_tail_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, t, n, scale, (__half*)out, ldo,
                       h2_off);
    return hip_status("halves_tail launch");
}

int bot_halves_tn_combine_f32(const float* a, const float* b, int32_t chunks, int32_t K, int32_t PP, int32_t P, const float* rem_a,
                              const float* rem_b, float* out, int64_t ldo, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(chunks >= 1 && K >= 1 && P >= 1 && PP >= P && ldo >= P, BOT_E_RANGE, "halves_tn_combine: chunks=%d K=%d PP=%d P=%d ldo=%lld", chunks, K, PP,
                P, (long long)ldo);
    BOT_REQUIRE(a && b && out && ((rem_a == nullptr) == (rem_b == nullptr)), BOT_E_NULL, "halves_tn_combine: NULL pointer (the remainders go together)");
    const int64_t total = (int64_t)K * P;
    hipLaunchKernelGGL(tn_combine_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, a, b, chunks, K, PP, P,
                       rem_a, rem_b, out, ldo);
    return hip_status("halves_tn_combine launch");
}

static int halves_split_impl(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, int32_t order, uint16_t* out,
                             int64_t ldo, int32_t piece, int32_t width, bot_stream_t stream);

int bot_halves_split_heads_f16(const float* x, int64_t ldx, int64_t n, int32_t H, int32_t D, const float* scale, uint16_t* out, int64_t ldo,
                               int32_t h2_off, int32_t DP, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && H >= 1 && D >= 1 && DP >= D && DP % 2 == 0 && h2_off >= H * DP && h2_off % 2 == 0 && ldo >= h2_off + (int64_t)H * DP && ldo % 2 == 0 &&
                    ldx >= (int64_t)H * D, BOT_E_RANGE, "halves_split_heads: n=%lld H=%d D=%d DP=%d h2_off=%d ldo=%lld ldx=%lld", (long long)n, H, D, DP, h2_off,
                (long long)ldo, (long long)ldx);
    BOT_REQUIRE(aligned(x, 4) && aligned(out, 4), BOT_E_ALIGN, "halves_split_heads: x and out must be 4-byte aligned");
    BOT_REQUIRE((x && out) || n == 0, BOT_E_NULL, "halves_split_heads: NULL pointer");
    if (n == 0) return 0;
    const bool wide = ldx % 2 == 0 && D % 2 == 0 && aligned(x, 8);
    hipLaunchKernelGGL(halves_split_heads_kernel, dim3((unsigned)n, (unsigned)((H * DP / 2 + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x, ldx, H, D, scale,
                       (__half*)out, ldo, h2_off, DP, wide);
    return hip_status("halves_split_heads launch");
}

int bot_halves_split_frag_f16(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, uint16_t* out, int32_t piece, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && F >= 1 && ldx >= F && piece >= F && piece % 64 == 0, BOT_E_RANGE, "halves_split_frag: n=%lld F=%d ldx=%lld piece=%d (a multiple of 64)",
                (long long)n, F, (long long)ldx, piece);
    BOT_REQUIRE(x && out && aligned(out, 16), BOT_E_NULL, "halves_split_frag: NULL or misaligned pointer");
    const int64_t tiles = (n + 15) / 16;
    hipLaunchKernelGGL(halves_split_frag_kernel, dim3((unsigned)(tiles * 16), (unsigned)((piece / 2 + 127) / 128)), dim3(128), 0, (hipStream_t)stream, x, ldx, n, F,
                       scale, (__half*)out, piece, tiles * (piece / 32) * 512);
    return hip_status("halves_split_frag launch");
}

int bot_halves_split_f16(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, int32_t order, uint16_t* out,
                         int64_t ldo, int32_t piece, bot_stream_t stream) {
    return halves_split_impl(x, ldx, n, F, scale, order, out, ldo, piece, piece, stream);
}

int bot_halves_split_cols_f16(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, int32_t order, uint16_t* out,
                              int64_t ldo, int32_t piece, int32_t width, bot_stream_t stream) {
    return halves_split_impl(x, ldx, n, F, scale, order, out, ldo, piece, width, stream);
}

static int halves_split_impl(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, int32_t order, uint16_t* out,
                             int64_t ldo, int32_t piece, int32_t width, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(width >= F && width <= piece && width % 2 == 0, BOT_E_RANGE, "halves_split: width=%d (F=%d piece=%d)", width, F, piece);
    BOT_REQUIRE(order == 0 || order == 1 || order == 2, BOT_E_RANGE, "halves_split: order=%d", order);
    BOT_REQUIRE(n >= 0 && F >= 1 && ldx >= F && piece >= F && ldo >= (order == 2 ? 2 : 3) * (int64_t)piece, BOT_E_RANGE,
                "halves_split: n=%lld F=%d ldx=%lld piece=%d ldo=%lld", (long long)n, F, (long long)ldx, piece, (long long)ldo);
    BOT_REQUIRE(piece % 2 == 0 && ldo % 2 == 0 && aligned(x, 4) && aligned(out, 4), BOT_E_ALIGN,
                "halves_split: piece and ldo must be even, x and out 4-byte aligned");
    const bool wide = ldx % 2 == 0 && aligned(x, 8);
    BOT_REQUIRE((x && out) || n == 0, BOT_E_NULL, "halves_split: NULL pointer");
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)n, (unsigned)((width / 2 + 127) / 128));
    if (order == 0) hipLaunchKernelGGL(halves_split_kernel<0>, grid, dim3(128), 0, st, x, ldx, F, scale, (__half*)out, ldo, piece, wide, width);
    else if (order == 2) hipLaunchKernelGGL(halves_split_kernel<2>, grid, dim3(128), 0, st, x, ldx, F, scale, (__half*)out, ldo, piece, wide, width);
    else hipLaunchKernelGGL(halves_split_kernel<1>, grid, dim3(128), 0, st, x, ldx, F, scale, (__half*)out, ldo, piece, wide, width);
    return hip_status("halves_split launch");
}

}  // extern "C"
