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
// One workgroup = 8 waves = 256 rows x NT*32 columns; the column block of B lives in LDS as three bf16 images [col][k]
// (the MFMA B fragment of a lane is 8 consecutive k of one column: one ds_read_b128); workgroups are persistent over row tiles
// so that the B images are built once; each wave streams its 32 rows of A from global memory (8 consecutive k per lane and
// k-step, the next step's loads issued before this step's MFMAs).
#include "common.h"

namespace bot {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSkMaxK = 256;
#ifndef SK_WAVES
#define SK_WAVES 12
#endif
constexpr int kSkWaves = SK_WAVES;    // waves per workgroup: two per SIMD, so that one wave's global-load / store latency hides behind the other's MFMAs
constexpr int kSkRows = 32 * kSkWaves;   // rows per workgroup tile (32 per wave)
constexpr int kSkThreads = 64 * kSkWaves;

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
    int32_t a_wide;        // rows of A are 16-byte (2) / 8-byte (1) aligned: float4 / float2 loads
    int32_t ny, gx, batch; // column blocks, row-tile sequences, batch entries (grid decode)
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

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pair (element 0 in the low half), round to nearest even: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int pack_bf16(float x0, float x1) {
    const f32x2 v = {x0, x1};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
// the three bf16 terms of two floats, packed pairwise
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned int& p1, unsigned int& p2, unsigned int& p3) {
    p1 = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16(r0, r1);
    p3 = pack_bf16(r0 - __uint_as_float(p2 << 16), r1 - __uint_as_float(p2 & 0xffff0000u));
}

union Frag {
    bf16x8 v;
    unsigned short s[8];
    unsigned int u[4];
    uint4 q;
};

template <int NT, int KP, int AW>
__global__ __launch_bounds__(kSkThreads) void skinny_gemm_kernel(SkinnyArgs a) {
    extern __shared__ unsigned short lds[];             // [3][NT*32][ks]
    constexpr int NB = NT * 32;
    constexpr int ks = KP + 8;                          // row pitch of the LDS images in halfs (compile time: every LDS offset is an immediate)
    // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs, so id % 8 is the XCD and id / 8 the slot on it.  The ny
    // column blocks of one (row-tile sequence, batch entry) take ADJACENT slots of ONE XCD: they stream the same rows of A at the
    // same time, and all but the first find them in that XCD's L2.
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int seq = (slot / a.ny) * 8 + xcd;            // which (row-tile sequence, batch entry)
    if (seq >= a.gx * a.batch) return;
    const int col0 = (slot % a.ny) * NB;
    const int64_t z = seq % a.batch;
    const int gxi = seq / a.batch;
    const float* A = a.A + z * a.sa;
    const float* B = a.B + z * a.sb;
    float* C = a.C + z * a.sc;
    // ---- the column block of B as three bf16 images [col][k]: k pairs per thread (one v_cvt_pk per term), four pairs' loads in
    // flight before any of them is used (B sits in the L2; a dependent load per element would cost a latency each)
    {
        constexpr int kh = KP / 2;                      // k pairs per column
        const int total = NB * kh;
        unsigned int* lds32 = reinterpret_cast<unsigned int*>(lds);
        constexpr int ks2 = ks / 2;
        constexpr int U = 4;
        for (int base = threadIdx.x; base < total; base += kSkThreads * U) {
            float v0[U], v1[U];
            int cc[U], kk2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * kSkThreads;
                int c, k2;
                if (a.b_is_kn) c = idx % NB, k2 = idx / NB;      // consecutive threads: consecutive columns of one k row
                else c = idx / kh, k2 = idx % kh;                 // consecutive threads: consecutive k of one stored row
                cc[u] = c, kk2[u] = k2;
                v0[u] = v1[u] = 0.f;
                if (idx < total && col0 + c < a.n) {
                    const int k = 2 * k2;
                    if (a.b_is_kn) {
                        if (k < a.k) v0[u] = B[(int64_t)k * a.ldb + col0 + c];
                        if (k + 1 < a.k) v1[u] = B[(int64_t)(k + 1) * a.ldb + col0 + c];
                    } else {
                        const float* bp = B + (int64_t)(col0 + c) * a.ldb + k;
                        if (k < a.k) v0[u] = bp[0];
                        if (k + 1 < a.k) v1[u] = bp[1];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * kSkThreads >= total) break;
                unsigned int p1, p2, p3;
                split3_pair(v0[u], v1[u], p1, p2, p3);
                lds32[(0 * NB + cc[u]) * ks2 + kk2[u]] = p1, lds32[(1 * NB + cc[u]) * ks2 + kk2[u]] = p2, lds32[(2 * NB + cc[u]) * ks2 + kk2[u]] = p3;
            }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    constexpr int nsteps = KP / 16;
    const int64_t ntiles = (a.m + kSkRows - 1) / kSkRows;
    // 8 consecutive k (k0 = 16 step + 8 h) of this lane's row of a tile, branch-free on the main path: rows beyond m are clamped to
    // the last row (their accumulator rows are never stored) and k beyond K is multiplied by the zero padding of the B images, so
    // only the ONE partial k-step of a row (K not a multiple of 8 per lane half) takes guarded element loads — it must not read
    // past the end of the matrix, and whatever it reads has to be finite.
    auto load8 = [&](float (&v)[8], int64_t tile, int step) {
#ifdef SK_ABLATE_LOAD    /* measurement builds only: no A traffic */
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)(lane + j + step) * 0.001f + (float)tile;
        return;
#endif
        int64_t row = tile * kSkRows + wave * 32 + r;
        row = row < a.m ? row : a.m - 1;
        const float* ar = A + row * a.lda;
        const int k0 = step * 16 + 8 * h;
        if (k0 + 8 <= a.k) {
            if constexpr (AW == 2) {
                const float4 t0 = *reinterpret_cast<const float4*>(ar + k0), t1 = *reinterpret_cast<const float4*>(ar + k0 + 4);
                v[0] = t0.x, v[1] = t0.y, v[2] = t0.z, v[3] = t0.w, v[4] = t1.x, v[5] = t1.y, v[6] = t1.z, v[7] = t1.w;
            } else if constexpr (AW == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 t2 = *reinterpret_cast<const float2*>(ar + k0 + 2 * j);
                    v[2 * j] = t2.x, v[2 * j + 1] = t2.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ar[k0 + j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = k0 + j < a.k ? ar[k0 + j] : 0.f;
        }
    };
    // A is fetched TWO k-steps at a time: the two lane halves of a row then consume one whole 128-byte line (k = 32 s' .. 32 s' + 31)
    // in one go — fetched one step at a time the second half of every line was a second L1 miss one iteration later.
    float cur[2][8], nxt[2][8];
    constexpr int npairs = (nsteps + 1) / 2;
    auto load16 = [&](float (&v)[2][8], int64_t tile, int pair) {
        load8(v[0], tile, 2 * pair);
        if (2 * pair + 1 < nsteps) load8(v[1], tile, 2 * pair + 1);
    };
    if (gxi < ntiles) load16(cur, gxi, 0);
    for (int64_t tile = gxi; tile < ntiles; tile += a.gx) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        Frag a1, a2, a3;
#pragma unroll 1
        for (int pair = 0; pair < npairs; ++pair) {
            {                        // the next pair of k-steps of this tile, or — at the end — the first pair of the next tile (in flight over the stores)
                if (pair + 1 < npairs) load16(nxt, tile, pair + 1);
                else if (tile + a.gx < ntiles) load16(nxt, tile + a.gx, 0);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int step = 2 * pair + half;
                if (step >= nsteps) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) split3_pair(cur[half][2 * j], cur[half][2 * j + 1], a1.u[j], a2.u[j], a3.u[j]);
                const unsigned short* bl = lds + r * ks + 8 * h + step * 16;      // + (p * NB + t * 32) * ks: immediates
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    Frag b1, b2, b3;
                    b1.q = *reinterpret_cast<const uint4*>(bl + (0 * NB + t * 32) * ks);
                    b2.q = *reinterpret_cast<const uint4*>(bl + (1 * NB + t * 32) * ks);
                    b3.q = *reinterpret_cast<const uint4*>(bl + (2 * NB + t * 32) * ks);
                    // smallest terms first: they are added into the accumulator before the leading product
#ifdef SK_ABLATE_MFMA   /* measurement builds only (tools/exp_skinny_ablate.sh): one product instead of six */
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, acc[t], 0, 0, 0);
#else
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3.v, b1.v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b3.v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b2.v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2.v, b1.v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b2.v, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.v, b1.v, acc[t], 0, 0, 0);
#endif
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[0][j] = nxt[0][j], cur[1][j] = nxt[1][j];
        }
        // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  The address of an element is
        // split into a wave-uniform part (tile, wave, register row: scalar registers) and a 32-bit per-lane offset (lane half, column).
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        const int64_t rwave = tile * kSkRows + wave_s * 32;
        float* Cw = C + rwave * a.ldc + col0;
#ifdef SK_ABLATE_STORE   /* measurement builds only: no C traffic unless a (never occurring) value shows up */
        bool any = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) any |= acc[t][0] == 1234.5f;
        if (!any) continue;
#endif
        if (rwave + 32 <= a.m) {            // the whole 32-row group exists: no per-row guards
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (col0 + t * 32 + r >= a.n) continue;
                const int voff = 4 * h * (int)a.ldc + t * 32 + r;
                if (a.accumulate) {         // all sixteen loads in flight, then the stores
                    float old[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) old[j] = (Cw + (int64_t)((j & 3) + 8 * (j >> 2)) * a.ldc)[voff];
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[t][j] += old[j];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) (Cw + (int64_t)((j & 3) + 8 * (j >> 2)) * a.ldc)[voff] = acc[t][j];
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (col0 + t * 32 + r >= a.n) continue;
                const int voff = 4 * h * (int)a.ldc + t * 32 + r;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int dr = (j & 3) + 8 * (j >> 2);
                    if (rwave + 4 * h + dr < a.m) {
                        float* p = Cw + (int64_t)dr * a.ldc + voff;
                        *p = a.accumulate ? *p + acc[t][j] : acc[t][j];
                    }
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
    SkinnyArgs a{A, lda, stride_a, B, ldb, stride_b, C, ldc, stride_c, m, n, k, (k + 15) / 16 * 16, b_is_kn, accumulate, 0, 0, 0, 0};
    a.a_wide = aligned(A, 16) && lda % 4 == 0 && stride_a % 4 == 0 ? 2 : (aligned(A, 8) && lda % 2 == 0 && stride_a % 2 == 0 ? 1 : 0);
    // the reduction axis is padded to one of four compiled lengths (every LDS offset of the inner loop is then an immediate);
    // column blocks of 128 (4 MFMA tiles) while the three LDS images fit, else 96
    const int kp = k <= 64 ? 64 : (k <= 128 ? 128 : (k <= 176 ? 176 : 256));
    a.kp = kp;
    const int nt = kp <= 176 ? 4 : 3;
    const int nb = nt * 32;
    const int ny = (n + nb - 1) / nb;
    const int64_t ntiles = (m + kSkRows - 1) / kSkRows;
    // persistent over row tiles: ONE workgroup per CU (the LDS images allow no second one) and no more workgroups than CUs —
    // a few left over for a second round would run alone for a whole round
    // (32 CUs per XCD: floor(32 / ny) row-tile sequences per XCD, each with its ny column blocks)
    int64_t gx = (int64_t)(32 / ny > 0 ? 32 / ny : 1) * 8 / batch;
    gx = gx < 1 ? 1 : (gx > ntiles ? ntiles : gx);
    const size_t lds = (size_t)3 * nb * (kp + 8) * 2;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    (void)hipGetDevice(&dev);
    a.ny = ny, a.gx = (int)gx, a.batch = batch;
    const dim3 grid((unsigned)(8 * ((gx * batch + 7) / 8) * ny));
#define BOT_SK_LAUNCH1(NT_, KP_, AW_)                                                                                                 \
    do {                                                                                                                              \
        static bool attr_set[64] = {};  /* the LDS size of an instantiation is fixed: raise its limit once per device */                \
        if (!attr_set[dev & 63]) {                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_gemm_kernel<NT_, KP_, AW_>),                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                          \
            attr_set[dev & 63] = true;                                                                                                \
        }                                                                                                                             \
        hipLaunchKernelGGL((skinny_gemm_kernel<NT_, KP_, AW_>), grid, dim3(kSkThreads), lds, st, a);                                    \
    } while (0)
#define BOT_SK_LAUNCH(NT_, KP_)                        \
    do {                                               \
        if (a.a_wide == 2) BOT_SK_LAUNCH1(NT_, KP_, 2); \
        else if (a.a_wide == 1) BOT_SK_LAUNCH1(NT_, KP_, 1); \
        else BOT_SK_LAUNCH1(NT_, KP_, 0);               \
    } while (0)
    if (kp == 64) BOT_SK_LAUNCH(4, 64);
    else if (kp == 128) BOT_SK_LAUNCH(4, 128);
    else if (kp == 176) BOT_SK_LAUNCH(4, 176);
    else BOT_SK_LAUNCH(3, 256);
#undef BOT_SK_LAUNCH1
#undef BOT_SK_LAUNCH
    set_kernel("bot::skinny_gemm_kernel<%d,%d> k=%d n=%d batch=%d", nt, kp, k, n, batch);
    return hip_status("skinny_gemm launch");
}

}  // extern "C"
