// bot_skinny_gemm_f32:  C[m, n] (+)= A[m, k] op(B)  for k <= 256 and huge m — the projections of the aggregate-before-project GAT
// layer (config-2 layer 0: three heads, Fin = 168, D = 250; src/no-sampling/models.py:490-492, 553-557 evaluated as
// W_h . sum a x).  Library fp32 GEMMs run these shapes at ~60 TFLOP/s (K = 168 is one or two tile steps: all prologue and
// epilogue), and the fp16-halves GEMMs of gemm.cpp lose their gain to the scale + split passes over operands this narrow.
//
// Here the fp32 operands are split IN REGISTERS into three bf16 terms each (a = a1 + a2 + a3 exactly: 3 x 8 significand bits;
// bf16 has fp32's exponent range, so no scale is needed) and a product is evaluated as
//     a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1            (dropped: 2^-24 relative and below)
// with v_mfma_f32_32x32x16_bf16 and fp32 accumulation: fp32-GEMM accuracy at 16/6 of the fp32 MFMA rate, no extra pass.
//
// One workgroup = 4 waves = 128 rows x NT*32 columns; the column block of B lives in LDS as three bf16 images [col][k]
// (the MFMA B fragment of a lane is 8 consecutive k of one column: one ds_read_b128); workgroups are persistent over row tiles
// so that the B images are built once; each wave streams its 32 rows of A from global memory (8 consecutive k per lane and
// k-step, the next step's loads issued before this step's MFMAs).
#include "common.h"

namespace bot {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSkMaxK = 256;
constexpr int kSkRows = 128;   // rows per workgroup tile (32 per wave)

struct SkinnyArgs {
    const float* A;
    int64_t lda, sa;
    const float* B;
    int64_t ldb, sb;
    float* C;
    int64_t ldc, sc;
    int64_t m;
    int32_t n, k, kp;      // kp = k rounded up to x16
    int32_t b_is_kn;       // B stored [k, n] (1) or [n, k] (0)
    int32_t accumulate;
    int32_t a_wide;        // rows of A are 8-byte aligned: float2 loads
};

__device__ __forceinline__ unsigned short bf16_rn(float x) {
    unsigned int u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);      // round to nearest even (NaN payloads aside)
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

__device__ __forceinline__ void split3(float x, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
    p1 = bf16_rn(x);
    const float r1 = x - bf16_f32(p1);
    p2 = bf16_rn(r1);
    p3 = bf16_rn(r1 - bf16_f32(p2));
}

union Frag {
    bf16x8 v;
    unsigned short s[8];
    uint4 q;
};

template <int NT>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs a) {
    extern __shared__ unsigned short lds[];             // [3][NT*32][ks]
    constexpr int NB = NT * 32;
    const int ks = a.kp + 8;                            // row pitch of the LDS images in halfs
    const int col0 = blockIdx.y * NB;
    const int64_t z = blockIdx.z;
    const float* A = a.A + z * a.sa;
    const float* B = a.B + z * a.sb;
    float* C = a.C + z * a.sc;
    // ---- the column block of B as three bf16 images [col][k]
    if (a.b_is_kn) {
        for (int idx = threadIdx.x; idx < NB * a.kp; idx += 256) {
            const int c = idx % NB, k = idx / NB;       // consecutive threads: consecutive columns of one k row
            const float v = (col0 + c < a.n && k < a.k) ? B[(int64_t)k * a.ldb + col0 + c] : 0.f;
            unsigned short p1, p2, p3;
            split3(v, p1, p2, p3);
            lds[(0 * NB + c) * ks + k] = p1, lds[(1 * NB + c) * ks + k] = p2, lds[(2 * NB + c) * ks + k] = p3;
        }
    } else {
        for (int idx = threadIdx.x; idx < NB * a.kp; idx += 256) {
            const int c = idx / a.kp, k = idx % a.kp;   // consecutive threads: consecutive k of one stored row
            const float v = (col0 + c < a.n && k < a.k) ? B[(int64_t)(col0 + c) * a.ldb + k] : 0.f;
            unsigned short p1, p2, p3;
            split3(v, p1, p2, p3);
            lds[(0 * NB + c) * ks + k] = p1, lds[(1 * NB + c) * ks + k] = p2, lds[(2 * NB + c) * ks + k] = p3;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int nsteps = a.kp / 16;
    const int64_t ntiles = (a.m + kSkRows - 1) / kSkRows;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = tile * kSkRows + wave * 32 + r;
        const bool rok = row < a.m;
        const float* ar = A + (rok ? row : 0) * a.lda;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        float cur[8], nxt[8];
        auto load8 = [&](float (&v)[8], int step) {
            const int k0 = step * 16 + 8 * h;
            if (rok && k0 + 8 <= a.k && a.a_wide) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 t2 = *reinterpret_cast<const float2*>(ar + k0 + 2 * j);
                    v[2 * j] = t2.x, v[2 * j + 1] = t2.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (rok && k0 + j < a.k) ? ar[k0 + j] : 0.f;
            }
        };
        load8(cur, 0);
        for (int step = 0; step < nsteps; ++step) {
            if (step + 1 < nsteps) load8(nxt, step + 1);
            Frag a1, a2, a3;
#pragma unroll
            for (int j = 0; j < 8; ++j) split3(cur[j], a1.s[j], a2.s[j], a3.s[j]);
            const int kk = step * 16 + 8 * h;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                Frag b1, b2, b3;
                const int c = t * 32 + r;
                b1.q = *reinterpret_cast<const uint4*>(&lds[(0 * NB + c) * ks + kk]);
                b2.q = *reinterpret_cast<const uint4*>(&lds[(1 * NB + c) * ks + kk]);
                b3.q = *reinterpret_cast<const uint4*>(&lds[(2 * NB + c) * ks + kk]);
                // smallest terms first: they are added into the accumulator before the leading product
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.v, b1.v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b3.v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b2.v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b1.v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b2.v, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
        }
        // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        const int64_t rbase = tile * kSkRows + wave * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = col0 + t * 32 + r;
            if (c >= a.n) continue;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int64_t rr = rbase + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (rr < a.m) {
                    float* p = C + rr * a.ldc + c;
                    *p = a.accumulate ? *p + acc[t][j] : acc[t][j];
                }
            }
        }
    }
}

}  // namespace bot

extern "C" {

int bot_skinny_gemm_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int32_t b_is_kn, float* C, int64_t ldc, int64_t m,
                        int32_t n, int32_t k, int32_t accumulate, int32_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                        bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(m >= 0 && n >= 1 && k >= 1 && k <= kSkMaxK && batch >= 1, BOT_E_RANGE, "skinny_gemm: m=%lld n=%d k=%d (k <= %d) batch=%d",
                (long long)m, n, k, kSkMaxK, batch);
    BOT_REQUIRE(lda >= k && ldb >= (b_is_kn ? n : k) && ldc >= n, BOT_E_RANGE, "skinny_gemm: lda=%lld ldb=%lld ldc=%lld", (long long)lda,
                (long long)ldb, (long long)ldc);
    if (m == 0) return 0;
    BOT_REQUIRE(A && B && C, BOT_E_NULL, "skinny_gemm: NULL pointer");
    BOT_REQUIRE(aligned(A, 4) && aligned(B, 4) && aligned(C, 4), BOT_E_ALIGN, "skinny_gemm: pointers must be 4-byte aligned");
    SkinnyArgs a{A, lda, stride_a, B, ldb, stride_b, C, ldc, stride_c, m, n, k, (k + 15) / 16 * 16, b_is_kn, accumulate, 0};
    a.a_wide = aligned(A, 8) && lda % 2 == 0 && stride_a % 2 == 0;
    // column blocks of 128 (4 MFMA tiles) while the three LDS images fit, else 96
    const int nt = (size_t)3 * 128 * (a.kp + 8) * 2 <= 160 * 1024 - 1024 ? 4 : 3;
    const int nb = nt * 32;
    const int ny = (n + nb - 1) / nb;
    const int64_t ntiles = (m + kSkRows - 1) / kSkRows;
    // persistent over row tiles: about two workgroups' worth of work queued per CU in total
    int64_t gx = (int64_t)512 / ((int64_t)ny * batch);
    gx = gx < 1 ? 1 : (gx > ntiles ? ntiles : gx);
    const size_t lds = (size_t)3 * nb * (a.kp + 8) * 2;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)gx, (unsigned)ny, (unsigned)batch);
    if (nt == 4) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_gemm_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((skinny_gemm_kernel<4>), grid, dim3(256), lds, st, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_gemm_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((skinny_gemm_kernel<3>), grid, dim3(256), lds, st, a);
    }
    set_kernel("bot::skinny_gemm_kernel<%d> k=%d n=%d batch=%d", nt, k, n, batch);
    return hip_status("skinny_gemm launch");
}

}  // extern "C"
