// bot_gemm_halves3_nt_f32 — the fp32 projection  C[m, n] = alpha[n] * sum_k x[m, k] w[n, k]  on the fp16 matrix cores from the two-term
// operand halves of halves.hip, hand-written for gfx950 (round 4).
//
// hipBLASLt (gemm.cpp) evaluates  a1 b1 + a1 b2 + a2 b1  as ONE fp16 GEMM over a three-fold reduction axis, A' = [a1 | a1 | 2^11 a2],
// B' = [b1 | b2 | 2^-11 b1]: every k-step of the library kernel stages a1 and b1 twice.  This kernel stages each operand's two halves
// ONCE per k-step (4 pieces instead of 6 through global -> LDS -> registers) and issues the three MFMAs per fragment pair from them:
//
//   acc += a1 . b1;   acc += a1 . b2;   acc += (2^11 a2) . (2^-11 b1)         (the last factor is a v_pk_mul_f16 of the b1 fragment:
//                                                                               exact, bit for bit the third piece of a right operand)
//
// Geometry: 256 x 256 output tile per 512-thread workgroup (8 waves as 2 (M) x 4 (N), 128 x 64 per wave), BK = 32 halves per piece per
// k-step, v_mfma_f32_16x16x32_f16 (96 per wave per k-step), two LDS stages of 4 pieces x 16 KB = 128 KB filled by global_load_lds_dwordx4
// (LDS-DMA: no staging registers) one k-step ahead; ONE raw s_barrier per k-step with the DMA of the next stage in flight across the
// whole MFMA phase.  LDS image per piece: [256 rows][64 B], the 16-byte chunk index XOR-swizzled with f(row quad) = (-(row >> 2)) & 3 so
// that each 16-lane group of a ds_read_b128 (MI355X_MICROARCH.md "LDS") covers all 64 banks; the DMA writes lane-linear, so the swizzle is
// applied to its per-lane SOURCE address (cdna_hip_programming.md rule 21).  The MFMA operands are swapped (D = W-fragment x X-fragment)
// so that a lane holds 4 CONSECUTIVE output columns of one row: float4 stores.  Workgroup -> tile order is XCD-aware: the column tiles of
// one row block run on one XCD (blocks b, b + 8, ... share an L2), so the X tile leaves HBM once.
//
// Reference call sites: the projections of src/no-sampling/models.py:490-492 and :558-560 (fc / res_fc, merged) and their input gradients.
#include <hip/hip_fp16.h>

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "common.h"

namespace bot {
namespace {

constexpr int BK = 32;                          // halves per piece per k-step: one v_mfma_f32_16x16x32_f16 deep

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct H3Args {
    const _Float16* A;      // [M, lda]: a1 at column 0, 2^11 a2 at column a2_off
    const _Float16* B;      // [N, ldb]: b1 at column 0, b2 at column b2_off
    const float* scale_a;   // (s, 1/s) pairs of the two operands (halves_scale): alpha = scale_a[1] * scale_b[1]
    const float* scale_b;
    float* C;               // [M, ldc]
    int64_t lda, ldb, ldc;
    int M, N, K;            // K: columns per piece, a multiple of BK
    int a2_off, b2_off;
    int tiles_m, tiles_n;
    // optional epilogue (grouped launches): C = act(C * col_scale[j] + col_shift[j]) per output column j = c_off + column (an eval-mode
    // BatchNorm / bias behind the layer, models.py:726-734), ReLU, and max|C| into by-product slots (common.h absmax_publish)
    const float* col_scale;
    const float* col_shift;
    int relu;
    uint32_t* absmax;
    // optional (bot_gemm_halves3_nt2_f32): A's columns from k-step k2 on are stored under a SECOND scale (scale_a2: the handful of attention
    // columns of a layer's gradient buffer, whose magnitude is known only after the big column blocks have been written under a bound):
    // the accumulators are multiplied by scale_a2[0] / scale_a[0] (a power of two: exact) in front of k-step k2, alpha = scale_a2[1] scale_b[1]
    const float* scale_a2;
    int k2;                 // -1: one scale
    // optional (grouped nt64 launches): column statistics of the stored values as a by-product - per 256-row tile t and output column c
    // stats_pivot[t F + c] = the tile's FIRST stored value of the column (written here: a pivot inside the data, so the squares do not
    // cancel for columns whose mean is far from zero - ABI 19), stats_part[(2 t + 0) F + c] = sum of (v - pivot),
    // [(2 t + 1) F + c] = sum of squares, stats_minmax likewise min / max (one row block per tile: bot_bn_stats_halves_partials_f32 combines
    // the tiles' (count, pivot, sum, sum of squares) exactly, in double)
    float* stats_part;
    float* stats_minmax;
    float* stats_pivot;
    int stats_F;
    // optional (bot_gemm_halves3_nt3_f32, gemm_halves3_nt64_*_bnb_kernel): C is the gradient arriving at a fused BatchNorm / ReLU / dropout
    // epilogue (dense.hip bn_act_bwd_*) whose input was bnb_x [M, N]: the column sums of that backward's reduce pass (masked gradient g,
    // g * xhat) and the column maxima of |g|, |xhat| leave with the tile - per 256-row tile t: bnb_part[(2 t + 0) N + c] = sum g,
    // [(2 t + 1) N + c] = sum g xhat, bnb_pmax likewise the maxima (the workspace layout of bn_act_bwd_reduce_kernel, one row block per tile)
    const float* bnb_x;
    int64_t bnb_ldx;
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_w;
    const float* bnb_b;
    int bnb_relu;
    float bnb_p;
    uint64_t bnb_seed;
    const uint64_t* bnb_seed_offset;
    float* bnb_part;
    float* bnb_pmax;
    int b_frag;             // B is a fragment-major RIGHT operand (halves.hip order 3; gemm_halves3_nt64_kernel only): rows of 16-row tiles = N rounded up
    int mode;               // 0 = the product.  Measurement switches (tools/exp_halves3.py): bit 0 no output stores; bit 2 / 3 B / A never
                            // advance along k; bit 5 the plain loop (barrier at the end of a k-step) and, in it, bit 6 no barrier / wait,
                            // bit 7 no DMA inside the loop, bit 8 one A fragment pair per k-step
};

// Grouped form (bot_gemm_halves3_nt_grouped_f32): the column tiles of a launch are GROUPS, each with its own B rows, output block and
// A columns - the per-head products of the aggregate-first GAT layer (fused.py:_GATHiddenAggFirst) as ONE launch:
//   C[r, c_off + j] = alpha * sum_{t < k_steps} sum_{i < 32} A[r, (t < k_seg ? a_col0 : a_col1) + 32 t + i] . B[b_row0 + j, 32 t + i],  j < n_valid <= 256
struct H3Group {
    int b_row0, n_valid;    // first B row of the tile; output columns written (the tile computes 256)
    int a_off0, a_off1;     // byte offsets added to A's k offset before / from k-step k_seg
    int k_steps, pad;
    int64_t c_off;          // offset (floats) of the group's first output column from C
};
constexpr int kMaxGroups = 12;
struct H3Groups {
    H3Group g[kMaxGroups];
    int k_seg;
};

// LDS-DMA, 16 bytes per lane: buffer_load_dwordx4 ... lds with the tile's resource descriptor in SGPRs, a per-lane 32-bit byte offset and a
// wave-uniform byte offset (SGPR).  (The buffer builtins exist in the device pass only: the host pass needs just the kernel's stub.)
__device__ __forceinline__ void dma16(const void* tile_base, unsigned char* lptr, uint32_t voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tile_base), 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lptr, 16, voff, soff, 0, 0);
#endif
}

// One BM x BN output tile per workgroup of WM x WN waves (wave tile MT x NT MFMA tiles of 16 x 16).  LDS: two stages of
// [a1 | a2 | b1 | b2], each piece [rows][64 B] with the 16-byte chunk index XOR-swizzled by f(row quad) = (-(row >> 2)) & 3.
// ABLATE: the measurement build (tools/exp_halves3.py --ablate): only there is `p.mode` read inside the kernel; the production
// instantiations (ABLATE = false) carry no switch, no branch and no live register for it.
template <int BM, int BN, int WM, int WN, bool PIPE, bool GROUPED, bool ABLATE>
__device__ __forceinline__ void gemm_halves3_nt_body(const H3Args& p, const H3Groups* groups) {
    constexpr int kWaves = WM * WN;
    constexpr int MT = BM / WM / 16, NT = BN / WN / 16;
    constexpr int kABytes = BM * BK * 2, kBBytes = BN * BK * 2;             // one piece of each operand
    constexpr int kStageBytes = 2 * kABytes + 2 * kBBytes;
    constexpr int GA = BM / 16 / kWaves, GB = BN / 16 / kWaves;              // 16-row groups (1 KB = one wave DMA instruction) per wave and piece
    static_assert(GA >= 1 && GB >= 1 && GA * kWaves * 16 == BM && GB * kWaves * 16 == BN, "tile / wave shape");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // XCD-aware tile order: blocks b and b + 8 share an XCD (and its L2); XCD x takes the row blocks 8 q + x and walks their column tiles
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int tm = (j / p.tiles_n) * 8 + xcd, tn = j % p.tiles_n;
    if (tm >= p.tiles_m) return;
    const int m0 = tm * BM;
    // the column tile: B rows n0 .., output columns c0 .. c0 + n_valid - 1 (offsets from C), T k-steps, A's k offset shifted by a_off0 /
    // a_off1 bytes before / from k-step k_seg (all zero / the plain tiling unless GROUPED)
    int n0 = tn * BN, n_valid = p.N - n0, T = p.K / BK, a_off0 = 0, a_off1 = 0, k_seg = 0;
    int64_t c0 = n0;
    if constexpr (GROUPED) {
        const H3Group& G = groups->g[tn];
        n0 = G.b_row0, n_valid = G.n_valid, T = G.k_steps, a_off0 = G.a_off0, a_off1 = G.a_off1, c0 = G.c_off, k_seg = groups->k_seg;
    }
    const float alpha = (p.scale_a2 ? p.scale_a2[1] : p.scale_a[1]) * p.scale_b[1];

    // LDS-DMA plan: lane i of a wave instruction lands at byte 16 i of its 1 KB row group: row i >> 2, stored chunk i & 3, which holds
    // the row's chunk (i & 3) ^ f(row quad) - the swizzle is applied to the per-lane SOURCE address (the DMA writes lane-linear)
    const int lr = lane >> 2, cq = (lane & 3) ^ ((-(lane >> 4)) & 3);
    // Addresses as a wave-uniform base (SGPR pair, advanced by a scalar add per k-step) + a per-lane 32-bit byte offset that never
    // changes: the DMA instruction then carries 4 bytes of address per lane instead of 8 and the loop has no address arithmetic on the
    // vector unit (an LDS-DMA's issue occupies the SIMD's vector issue port, which its partner wave's MFMAs need).  Offsets are
    // relative to the tile's first row, so they stay below 256 rows x the row pitch whatever the operand's size.
    // (buffer_load_dwordx4 ... lds: resource descriptor of the tile's first row in SGPRs + constant VGPR offset + SGPR k offset)
    const _Float16* tileA = p.A + (int64_t)m0 * p.lda;
    const _Float16* tileB = p.B + (int64_t)n0 * p.ldb;
    uint32_t offA[2 * GA], offB[2 * GB];
#pragma unroll
    for (int h = 0; h < GA; ++h) {
        const uint32_t r = (uint32_t)(min(m0 + (w + kWaves * h) * 16 + lr, p.M - 1) - m0) * (uint32_t)p.lda + cq * 8;
        offA[h] = r * 2, offA[GA + h] = (r + p.a2_off) * 2;
    }
#pragma unroll
    for (int h = 0; h < GB; ++h) {
        const uint32_t r = (uint32_t)(min(n0 + (w + kWaves * h) * 16 + lr, p.N - 1) - n0) * (uint32_t)p.ldb + cq * 8;
        offB[h] = r * 2, offB[GB + h] = (r + p.b2_off) * 2;
    }
    const int stepA = (ABLATE && (p.mode & 8)) ? 0 : BK * 2, stepB = (ABLATE && (p.mode & 4)) ? 0 : BK * 2;     // ablations: an operand re-read from its first k-step
    // DMA instruction i of a k-step (0 .. kDma - 1): a1 groups, a2 groups, b1 groups, b2 groups; `kt` = the k-step being fetched
    constexpr int kDma = 2 * GA + 2 * GB;
    auto issue_one = [&](int i, int stage, int kt) {
        unsigned char* base = lds + stage * kStageBytes + w * 1024;
        if (i < 2 * GA) {
            const int q = i / GA, h = i % GA;
            dma16(tileA, base + q * kABytes + h * kWaves * 1024, offA[i], GROUPED ? kt * stepA + (kt < k_seg ? a_off0 : a_off1) : kt * stepA);
        } else {
            const int k = i - 2 * GA, q = k / GB, h = k % GB;
            dma16(tileB, base + 2 * kABytes + q * kBBytes + h * kWaves * 1024, offB[k], kt * stepB);
        }
    };
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < kDma; ++i) issue_one(i, stage, kt);
    };

    // fragment addresses: lane l reads row (l & 15) of a 16-row tile, chunk l >> 4, stored at chunk ^ f(row quad)
    const int wr = w / WN, wc = w % WN;
    const int frow = lane & 15, fch = (lane >> 4) ^ ((-(frow >> 2)) & 3);
    const int a_off = (wr * MT * 16 + frow) * 64 + fch * 16;                      // + mt * 1024
    const int b_off = 2 * kABytes + (wc * NT * 16 + frow) * 64 + fch * 16;        // + nt * 1024

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto rescale = [&]() {       // in front of k-step k2: what was accumulated under scale_a continues under scale_a2
        const float r = p.scale_a2[0] * p.scale_a[1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * r;
    };
    auto mfma12 = [&](int mt, const half8& a1, const half8& a2, const half8 (&b1)[NT], const half8 (&b2)[NT], const half8 (&b1s)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // operands swapped: D[i = output column within the tile][j = output row] -> a lane holds 4 consecutive columns of one row
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[nt], a1, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2[nt], a1, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1s[nt], a2, acc[mt][nt], 0, 0, 0);
        }
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (!PIPE) {
        auto kstep = [&](int t, auto more_tag) {
            constexpr bool more = decltype(more_tag)::value;     // the last k-step is peeled: no branch around the DMA inside the MFMA phase
            const int nstage = (t + 1) & 1;
            const unsigned char* st = lds + (t & 1) * kStageBytes;
            if (t == p.k2) rescale();
            half8 b1[NT], b2[NT], b1s[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b1[nt] = *reinterpret_cast<const half8*>(st + b_off + nt * 1024);
                b2[nt] = *reinterpret_cast<const half8*>(st + b_off + kBBytes + nt * 1024);
                b1s[nt] = b1[nt] * (_Float16)(1.0f / kHalvesShift);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int am = (ABLATE && (p.mode & 256)) ? 0 : mt;     // ablation: one A fragment pair per k-step instead of MT
                const half8 a1 = *reinterpret_cast<const half8*>(st + a_off + am * 1024);
                const half8 a2 = *reinterpret_cast<const half8*>(st + a_off + kABytes + am * 1024);
                // the DMA of the next stage is spread over the MFMA phase (one or two instructions per row of MFMA tiles)
                if (more && !(ABLATE && (p.mode & 128))) {              // ablation: no LDS-DMA in the loop
#pragma unroll
                    for (int i = mt * kDma / MT; i < (mt + 1) * kDma / MT; ++i) issue_one(i, nstage, t + 1);
                }
                mfma12(mt, a1, a2, b1, b2, b1s);
            }
            if (!(ABLATE && (p.mode & 64))) {                           // ablation: no wait, no barrier
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        };
        for (int t = 0; t + 1 < T; ++t) kstep(t, std::true_type{});
        kstep(T - 1, std::false_type{});
    } else {
        // Software-pipelined across the barrier.  The ONE barrier of a k-step sits in front of the step's LAST row of MFMA tiles: by then
        // every read of the step's stage has returned (lgkmcnt(0)) and this wave's share of the next stage has landed (vmcnt(0)), so
        // behind it (a) the next step's B fragments and first A fragments are requested, (b) the DMA of the step after next starts into
        // the stage just freed - and the 12 MFMAs of the last tile row, whose operands are in registers, cover the LDS latency that a
        // barrier at the END of the step exposes on both waves of a SIMD at once.  DMA instructions of step s: two behind the barrier
        // of step s - 2, two each beside tile rows 0 .. 2 of step s - 1 (landed long before the next barrier).
        static_assert(kDma == 8 && MT >= 4, "DMA schedule of the pipelined loop");
        half8 b1[NT], b2[NT], b1s[NT], a1, a2;
        auto read_b = [&](const unsigned char* st, half8 (&x1)[NT], half8 (&x2)[NT]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                x1[nt] = *reinterpret_cast<const half8*>(st + b_off + nt * 1024);
                x2[nt] = *reinterpret_cast<const half8*>(st + b_off + kBBytes + nt * 1024);
            }
        };
        if (T > 1) {
            issue_one(0, 1, 1);
            issue_one(1, 1, 1);
        }
        read_b(lds, b1, b2);
        a1 = *reinterpret_cast<const half8*>(lds + a_off);
        a2 = *reinterpret_cast<const half8*>(lds + a_off + kABytes);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b1s[nt] = b1[nt] * (_Float16)(1.0f / kHalvesShift);
        for (int t = 0; t < T; ++t) {
            const bool more = t + 1 < T, more2 = t + 2 < T;
            const unsigned char* st = lds + (t & 1) * kStageBytes;
            const unsigned char* sn = lds + ((t + 1) & 1) * kStageBytes;
            if (t == p.k2) rescale();        // (wave-uniform, once per tile; k2 = -1 without a second scale)
#pragma unroll
            for (int mt = 0; mt < MT - 1; ++mt) {
                const half8 an1 = *reinterpret_cast<const half8*>(st + a_off + (mt + 1) * 1024);
                const half8 an2 = *reinterpret_cast<const half8*>(st + a_off + kABytes + (mt + 1) * 1024);
                if (more && mt < 3) {
                    issue_one(2 + 2 * mt, (t + 1) & 1, t + 1);
                    issue_one(3 + 2 * mt, (t + 1) & 1, t + 1);
                }
                mfma12(mt, a1, a2, b1, b2, b1s);
                a1 = an1, a2 = an2;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            half8 b1n[NT], b2n[NT], an1 = a1, an2 = a2;
            if (more2) {
                issue_one(0, t & 1, t + 2);
                issue_one(1, t & 1, t + 2);
            }
            if (more) {
                read_b(sn, b1n, b2n);
                an1 = *reinterpret_cast<const half8*>(sn + a_off);
                an2 = *reinterpret_cast<const half8*>(sn + a_off + kABytes);
            }
            mfma12(MT - 1, a1, a2, b1, b2, b1s);
            if (more) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    b1[nt] = b1n[nt], b2[nt] = b2n[nt];
                    b1s[nt] = b1n[nt] * (_Float16)(1.0f / kHalvesShift);
                }
                a1 = an1, a2 = an2;
            }
        }
        __builtin_amdgcn_s_barrier();       // the epilogue reuses the stages: every wave is past its last fragment read
    }

    // epilogue: acc[mt][nt][r] = C[m0 + wr MT 16 + mt 16 + (lane & 15)][n0 + wc NT 16 + nt 16 + (lane >> 4) 4 + r]
    if constexpr (ABLATE) {
        if (p.mode & 1) {
            if (acc[0][0][0] == 1.2345e-33f) p.C[0] = acc[MT - 1][NT - 1][3] + acc[MT / 2][1][2];      // keeps the accumulators live
            return;
        }
    }
    float* const Cb = p.C + c0;                     // the tile's first output column
    const bool vec_ok = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0);
    const bool vec2_ok = (p.ldc % 2 == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 7) == 0);
    // Through LDS, so that a store instruction writes 4 rows x 256 contiguous bytes (full 128-byte lines) instead of 16 rows x 64 bytes:
    // the accumulator layout gives a lane 4 consecutive columns of ONE row per tile, 16 rows per instruction (measured with the
    // full-line address pattern as an ablation: -0.06 ... -0.14 ms of 1.1).  All stages are free after the last barrier; every wave
    // stages its own 32-row slabs (2 rows of MFMA tiles) in a private region, pitch NT 16 + 4 floats: conflict-free ds_write_b128.
    static_assert(MT % 2 == 0 && NT == 4, "epilogue staging assumes 64-column wave tiles, an even number of tile rows");
    constexpr int P = NT * 16 + 4;
    static_assert(32 * P * 4 <= 2 * kStageBytes / kWaves, "staging slab does not fit the wave's share of the LDS");
    float* stg = reinterpret_cast<float*>(lds + w * (2 * kStageBytes / kWaves));
    const int q4 = (lane >> 4) * 4, l15 = lane & 15;
    const int col = wc * NT * 16 + l15 * 4;         // within the tile
    // optional per-column affine + ReLU + max|C| (this lane's four columns are the same in every pass)
    float cs[4] = {1.f, 1.f, 1.f, 1.f}, ch[4] = {0.f, 0.f, 0.f, 0.f};
    const bool affine = p.col_scale != nullptr || p.col_shift != nullptr;
    if (affine) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < n_valid) {
                if (p.col_scale) cs[e] = p.col_scale[c0 + col + e];
                if (p.col_shift) ch[e] = p.col_shift[c0 + col + e];
            }
    }
    float amax = 0.f;
#pragma unroll
    for (int pass = 0; pass < MT / 2; ++pass) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 v = acc[pass * 2 + mm][nt] * alpha;
                *reinterpret_cast<f32x4*>(stg + (mm * 16 + l15) * P + nt * 16 + q4) = v;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 4 + (lane >> 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * P + l15 * 4);
            const int row = m0 + wr * MT * 16 + pass * 32 + r;
            if (affine || p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(v[e], cs[e], ch[e]);
                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                }
            }
            if (p.absmax && row < p.M) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < n_valid) amax = fmaxf(amax, fabsf(v[e]));
            }
            if (row < p.M) {
                float* c = Cb + (int64_t)row * p.ldc + col;
                if (vec_ok && col + 3 < n_valid) {
                    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (vec2_ok && col + 3 < n_valid) {      // rows on an 8-byte pitch (the [N, 750] input gradient)
                    *reinterpret_cast<float2*>(c) = make_float2(v[0], v[1]);
                    *reinterpret_cast<float2*>(c + 2) = make_float2(v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < n_valid) c[e] = v[e];
                }
            }
        }
    }
    if (p.absmax) absmax_publish(wave_absmax(amax), p.absmax);      // (every lane of every wave arrives here)
}

template <int BM, int BN, int WM, int WN, bool PIPE, bool ABLATE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_halves3_nt_kernel(H3Args p) {
    gemm_halves3_nt_body<BM, BN, WM, WN, PIPE, false, ABLATE>(p, nullptr);
}

template <int BM, int BN, int WM, int WN, bool PIPE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_halves3_nt_grouped_kernel(H3Args p, H3Groups groups) {
    gemm_halves3_nt_body<BM, BN, WM, WN, PIPE, true, false>(p, &groups);
}

// ----------------------------------------------------------------------------------------------------------------------------
// The 128-byte-line form (round 5).  Ablations of the kernel above and of a B-direct variant (profiles/r05_nt_bd.txt)
// showed that what the k-loop pays besides the MFMAs is the operand traffic a CU pulls from its L2 - 64 KB per k-step - and that its price
// is set by the ADDRESS PATTERN, not the mechanism: a wave instruction that fetches 16 rows x 64 bytes (half cache lines, BK = 32 halves)
// costs about twice one that fetches 8 rows x 128 bytes (timing-only pattern switches on the B-direct kernel: 1.115 -> 1.021 ms forward,
// 1.065 -> 0.973 ms input gradient).  This kernel stages TWO k-steps per iteration so that every fetch is a whole line:
//   * 8 waves as 1 (M) x 8 (N): a wave owns all 256 rows x 32 columns of the tile (MT = 16, NT = 2; 96 MFMAs per k-step as before);
//   * LEFT operand through LDS: per iteration and piece [256 rows][128 B] = 32 KB, two pieces, two stages = 128 KB; a DMA instruction
//     writes 8 rows x 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7, which makes every 16-lane group of a
//     ds_read_b128 (16 rows, two adjacent chunks) hit 64 different banks; the swizzle rides on the DMA's per-lane SOURCE address;
//   * the wave's own 32 weight rows straight from global memory / L2 into registers (nobody else needs them): the two k-steps' fragments
//     of a row are the two halves of ONE line, loaded back to back (the second hits the line the first brought in), one iteration ahead;
//   * ONE barrier per iteration (two k-steps), placed in front of the last row of MFMA tiles as above.
// Same operands, same three products in the same order per accumulator as the kernel above: bit for bit its result.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ half8 load16(const void* tile_base, uint32_t voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tile_base), 0, 0x7fffffff, 0x00020000);
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
#else
    return half8{};
#endif
}

template <bool GROUPED, bool BFRAG, bool BNB = false>
__device__ __forceinline__ void gemm_halves3_nt64_body(const H3Args& p, const H3Groups* groups) {
    static_assert(!(GROUPED && BNB), "the BatchNorm-backward by-product indexes the output columns of a plain launch");
    constexpr int BM = 256, BN = 256, kWaves = 8, MT = BM / 16, NT = BN / kWaves / 16;
    constexpr int kRow = 4 * BK;                             // bytes of a row per piece and iteration: two k-steps = one 128-byte line
    constexpr int kABytes = BM * kRow;                       // one piece of A per iteration: 32 KB
    constexpr int kStageBytes = 2 * kABytes, kStages = 2;
    constexpr int GA = BM / 8 / kWaves;                      // 8-row groups (one 1 KB DMA instruction) per wave and piece: 4
    static_assert(NT == 2 && GA == 4 && kRow == 128, "wave tile 256 x 32, 128-byte rows");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kStages * kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int tm = (j / p.tiles_n) * 8 + xcd, tn = j % p.tiles_n;
    if (tm >= p.tiles_m) return;
    const int m0 = tm * BM;
    int n0 = tn * BN, n_valid = p.N - n0, T = p.K / BK, a_off0 = 0, a_off1 = 0, k_seg = 0;
    int64_t c0 = n0;
    if constexpr (GROUPED) {
        const H3Group& G = groups->g[tn];
        n0 = G.b_row0, n_valid = G.n_valid, T = G.k_steps, a_off0 = G.a_off0, a_off1 = G.a_off1, c0 = G.c_off, k_seg = groups->k_seg;
    }
    const int T2 = T >> 1;                                   // iterations (the launcher takes this form for an even number of k-steps only)
    const float alpha = (p.scale_a2 ? p.scale_a2[1] : p.scale_a[1]) * p.scale_b[1];

    // A: lane i of DMA instruction g (8-row group g = w + 8 h) lands at byte 16 i of the group's 1 KB: row 8 g + (i >> 3), stored chunk
    // i & 7, which holds the row's chunk (i & 7) ^ ((row >> 1) & 7)
    const _Float16* tileA = p.A + (int64_t)m0 * p.lda;
    const _Float16* tileB = p.B + (int64_t)n0 * p.ldb;
    // (per-lane offsets only where they differ per lane: the second half's column offset and the k offset ride in the scalar offset)
    uint32_t offA[GA];
#pragma unroll
    for (int h = 0; h < GA; ++h) {
        const int row = (w + kWaves * h) * 8 + (lane >> 3);
        offA[h] = ((uint32_t)(min(m0 + row, p.M - 1) - m0) * (uint32_t)p.lda + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2;
    }
    const int a2b = p.a2_off * 2;
    auto issue_a = [&](int i, int stage, int it) {           // i = 0 .. 7: a1 groups w, w + 8, w + 16, w + 24, then a2 likewise
        const int q = i / GA, h = i % GA;
        dma16(tileA, lds + stage * kStageBytes + q * kABytes + (w + kWaves * h) * 1024, offA[h],
              (GROUPED ? it * kRow + (2 * it < k_seg ? a_off0 : a_off1) : it * kRow) + q * a2b);
    };
    // B: lane l holds row l & 15 of a 16-row tile, bytes 16 (l >> 4) .. + 15 of a k-step's 64: k-step 2 it at byte 0, 2 it + 1 at byte 64 of the line
    // BFRAG: the operand is stored fragment-major (halves.hip order 3): the fragment of 16-row tile t and k-step s is the KB at ((t T + s) KB),
    // lane l at byte 16 l - one contiguous KB per load instruction; h2 one region behind
    uint32_t offB[NT];
    int tileoff[NT];
    const int Tk = p.K / BK;                                 // (BFRAG: the plain launch only - the operand's k-steps per tile)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if constexpr (BFRAG) {
            offB[nt] = lane * 16;
            tileoff[nt] = min((n0 + w * (NT * 16) + nt * 16) >> 4, ((p.N + 15) >> 4) - 1) * Tk * 1024;
        } else {
            offB[nt] = ((uint32_t)(min(n0 + w * (NT * 16) + nt * 16 + (lane & 15), p.N - 1) - n0) * (uint32_t)p.ldb) * 2 + (lane >> 4) * 16;
            tileoff[nt] = 0;
        }
    }
    const int b2 = BFRAG ? ((p.N + 15) >> 4) * Tk * 1024 : p.b2_off * 2;
    const _Float16* baseB = BFRAG ? p.B : tileB;
    auto load_b = [&](half8 (&x1)[NT], half8 (&x2)[NT], int it, int sub) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int so = BFRAG ? tileoff[nt] + (2 * it + sub) * 1024 : it * kRow + sub * 64;
            x1[nt] = load16(baseB, offB[BFRAG ? 0 : nt], so);
            x2[nt] = load16(baseB, offB[BFRAG ? 0 : nt], so + b2);
        }
    };
    // A fragments: lane l reads row l & 15 of a 16-row tile, chunk 4 sub + (l >> 4), stored at chunk ^ ((row >> 1) & 7)
    const int frow = lane & 15, fkey = (frow >> 1) & 7;
    const int a_off_s0 = frow * kRow + (((lane >> 4) ^ fkey) * 16);              // + mt * 2048 (+ kABytes for the second half)
    const int a_off_s1 = frow * kRow + (((4 + (lane >> 4)) ^ fkey) * 16);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto rescale = [&]() {
        const float r = p.scale_a2[0] * p.scale_a[1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * r;
    };
    auto mfma6 = [&](int mt, const half8& a1, const half8& a2, const half8 (&b1)[NT], const half8 (&b2h)[NT], const half8 (&b1s)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[nt], a1, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2h[nt], a1, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1s[nt], a2, acc[mt][nt], 0, 0, 0);
        }
    };
    const _Float16 sh = (_Float16)(1.0f / kHalvesShift);

    // prologue: A of iteration 0 and B of both its k-steps
    half8 x1[NT], x2[NT], y1[NT], y2[NT], z1[NT], z2[NT], bs[NT], a1, a2;       // x: k-step 2 it, y: k-step 2 it + 1, z: the next iteration's second
#pragma unroll
    for (int i = 0; i < 2 * GA; ++i) issue_a(i, 0, 0);
    load_b(x1, x2, 0, 0);
    load_b(y1, y2, 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    a1 = *reinterpret_cast<const half8*>(lds + a_off_s0);
    a2 = *reinterpret_cast<const half8*>(lds + a_off_s0 + kABytes);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        // (uses the compiler can see: its wait-count pass then knows these registers have landed HERE; left pending into the loop it
        // re-waits at the head of every iteration with vmcnt(0), behind the loads just issued for the next one)
        asm volatile("" : "+v"(x1[nt]), "+v"(x2[nt]), "+v"(y1[nt]), "+v"(y2[nt]));
    }
    // One iteration = two k-steps.  MORE (iteration it + 1 exists) is compile-time: the steady-state loop has no branch around its loads.
    auto iteration = [&](int it, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        const int stage = it & 1;
        const unsigned char* st = lds + stage * kStageBytes;
        const unsigned char* sn = lds + (stage ^ 1) * kStageBytes;
        // ---- k-step 2 it: fragments x, the DMA of iteration it + 1 spread over the first tile rows (its stage is free since the
        // barrier of iteration it - 1)
        if (2 * it == p.k2) rescale();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bs[nt] = x1[nt] * sh;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int nx = mt + 1 < MT ? a_off_s0 + (mt + 1) * 2048 : a_off_s1;      // the last row prefetches the first fragments of k-step 2 it + 1
            const half8 an1 = *reinterpret_cast<const half8*>(st + nx);
            const half8 an2 = *reinterpret_cast<const half8*>(st + nx + kABytes);
            if constexpr (more) {
                if (mt < 2 * GA) issue_a(mt, stage ^ 1, it + 1);
            }
            mfma6(mt, a1, a2, x1, x2, bs);
            a1 = an1, a2 = an2;
        }
        // ---- the next iteration's weight fragments: both k-steps of a row are the two halves of one line, loaded back to back (x is dead)
        if constexpr (more) {
            __builtin_amdgcn_sched_barrier(0);
            load_b(x1, x2, it + 1, 0);
            load_b(z1, z2, it + 1, 1);
            __builtin_amdgcn_sched_barrier(0);       // (left alone, the scheduler sinks these loads to the end of the iteration)
        }
        // ---- k-step 2 it + 1: fragments y
        if (2 * it + 1 == p.k2) rescale();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bs[nt] = y1[nt] * sh;
#pragma unroll
        for (int mt = 0; mt < MT - 1; ++mt) {
            const half8 an1 = *reinterpret_cast<const half8*>(st + a_off_s1 + (mt + 1) * 2048);
            const half8 an2 = *reinterpret_cast<const half8*>(st + a_off_s1 + kABytes + (mt + 1) * 2048);
            mfma6(mt, a1, a2, y1, y2, bs);
            a1 = an1, a2 = an2;
        }
        // the next iteration's operands have landed (they were issued a k-step or more ago) and every read of this stage has returned
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        half8 an1 = a1, an2 = a2;
        if constexpr (more) {
            an1 = *reinterpret_cast<const half8*>(sn + a_off_s0);
            an2 = *reinterpret_cast<const half8*>(sn + a_off_s0 + kABytes);
        }
        mfma6(MT - 1, a1, a2, y1, y2, bs);
        if constexpr (more) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) y1[nt] = z1[nt], y2[nt] = z2[nt];
            a1 = an1, a2 = an2;
        }
    };
    int it = 0;
    for (; it + 1 < T2; ++it) iteration(it, std::true_type{});
    iteration(it, std::false_type{});
    __builtin_amdgcn_s_barrier();       // the epilogue reuses the stages: every wave is past its last fragment read

    // epilogue: acc[mt][nt][r] = C[m0 + mt 16 + (lane & 15)][c0 + w 32 + nt 16 + (lane >> 4) 4 + r], staged per wave in slabs of 32 rows x 32
    // columns (pitch 36 floats) so that a store instruction writes 8 rows x 128 contiguous bytes
    float* const Cb = p.C + c0;
    const bool vec_ok = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0);
    const bool vec2_ok = (p.ldc % 2 == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 7) == 0);
    constexpr int P = NT * 16 + 4;
    static_assert(32 * P * 4 <= kStages * kStageBytes / kWaves, "staging slab does not fit the wave's share of the LDS");
    float* stg = reinterpret_cast<float*>(lds + w * (kStages * kStageBytes / kWaves));
    const int q4 = (lane >> 4) * 4, l15 = lane & 15, l7 = lane & 7;
    const int col = w * (NT * 16) + l7 * 4;         // within the tile: this lane's four columns, the same in every pass
    float cs[4] = {1.f, 1.f, 1.f, 1.f}, ch[4] = {0.f, 0.f, 0.f, 0.f};
    const bool affine = p.col_scale != nullptr || p.col_shift != nullptr;
    if (affine) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < n_valid) {
                if (p.col_scale) cs[e] = p.col_scale[c0 + col + e];
                if (p.col_shift) ch[e] = p.col_shift[c0 + col + e];
            }
    }
    float amax = 0.f;
    // column statistics of what is stored (the wave owns ALL rows of its 32 columns: no cross-wave step)
    const bool stats = p.stats_part != nullptr;
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f}, st_mn[4], st_mx[4], piv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) st_mn[e] = INFINITY, st_mx[e] = -INFINITY;
    // BatchNorm-backward by-product (BNB): this lane's quad of columns is a Philox quad of the epilogue's dropout mask (c0 + col is a multiple
    // of 4), the x rows are read 8 rows x 128 bytes per instruction like the stores
    float b_s[4] = {0.f, 0.f, 0.f, 0.f}, b_q[4] = {0.f, 0.f, 0.f, 0.f}, b_gm[4] = {0.f, 0.f, 0.f, 0.f}, b_xm[4] = {0.f, 0.f, 0.f, 0.f};
    float b_mu[4] = {0.f, 0.f, 0.f, 0.f}, b_is[4] = {0.f, 0.f, 0.f, 0.f}, b_sc[4] = {1.f, 1.f, 1.f, 1.f}, b_sh[4] = {0.f, 0.f, 0.f, 0.f};
    int b_nv = 0;
    bool b_wide = false;
    uint64_t b_seed = 0;
    int64_t b_nquad = 0;
    float b_keep = 1.f;
    if constexpr (BNB) {
        b_nv = max(0, min(4, n_valid - col));
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < b_nv) {
                b_mu[e] = p.bnb_mean[c0 + col + e], b_is[e] = p.bnb_invstd[c0 + col + e];
                if (p.bnb_w) b_sc[e] = p.bnb_w[c0 + col + e];
                if (p.bnb_b) b_sh[e] = p.bnb_b[c0 + col + e];
            }
        b_wide = (p.bnb_ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.bnb_x) & 15) == 0);
        b_seed = p.bnb_seed_offset ? p.bnb_seed + p.bnb_seed_offset[0] * 0x9E3779B97F4A7C15ull : p.bnb_seed;      // dense.hip eff_seed
        b_nquad = (p.N + 3) / 4;
        b_keep = p.bnb_p > 0.f ? 1.f / (1.f - p.bnb_p) : 1.f;
    }
    // the x quads of a pass are requested two passes ahead (a ring of three): a pass is short, an HBM round trip is not
    f32x4 b_ring[3][4];
    auto load_x = [&](auto pass_c) __attribute__((always_inline)) {
        constexpr int pass = decltype(pass_c)::value;
        if constexpr (BNB && pass < MT / 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + pass * 32 + i * 8 + (lane >> 3);
                f32x4& q = b_ring[pass % 3][i];
                q = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row < p.M && b_nv > 0) {
                    const float* xp = p.bnb_x + (int64_t)row * p.bnb_ldx + c0 + col;
                    if (b_wide && b_nv == 4) {
                        q = *reinterpret_cast<const f32x4*>(xp);
                    } else {
                        const float2 lo = *reinterpret_cast<const float2*>(xp);
                        q[0] = lo.x, q[1] = lo.y;
                        if (b_nv == 4) {
                            const float2 hi = *reinterpret_cast<const float2*>(xp + 2);
                            q[2] = hi.x, q[3] = hi.y;
                        }
                    }
                }
            }
        }
    };
    load_x(std::integral_constant<int, 0>{});
    load_x(std::integral_constant<int, 1>{});
    // (a compile-time pass index: the accumulator tiles must be addressed by constants whatever the unroller decides about a body this size)
    auto do_pass = [&](auto pass_c) __attribute__((always_inline)) {
        constexpr int pass = decltype(pass_c)::value;
        load_x(std::integral_constant<int, pass + 2>{});
        const f32x4 (&b_x)[4] = b_ring[pass % 3];
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 v = acc[pass * 2 + mm][nt] * alpha;
                *reinterpret_cast<f32x4*>(stg + (mm * 16 + l15) * P + nt * 16 + q4) = v;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 8 + (lane >> 3);
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * P + l7 * 4);
            const int row = m0 + pass * 32 + r;
            if (affine || p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(v[e], cs[e], ch[e]);
                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                }
            }
            if constexpr (pass == 0) {
                if (stats && i == 0) {      // the tile's first row (always a row of the matrix) sits in lanes 0 .. 7 of this read: the pivot
#pragma unroll
                    for (int e = 0; e < 4; ++e) piv[e] = __shfl(v[e], l7);
                }
            }
            if (stats && row < p.M) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[e] - piv[e];
                    st_s[e] += d;
                    st_q[e] = fmaf(d, d, st_q[e]);
                    st_mn[e] = fminf(st_mn[e], v[e]), st_mx[e] = fmaxf(st_mx[e], v[e]);
                }
            }
            if constexpr (BNB) {
                if (row < p.M && b_nv > 0) {
                    float f[4] = {1.f, 1.f, 1.f, 1.f};
                    if (p.bnb_p > 0.f) {        // dense.hip drop_factors: element (r, c) = word c % 4 of the block with counter r * ceil(F / 4) + c / 4
                        uint32_t wd[4];
                        Philox::gen(b_seed, (uint64_t)((int64_t)row * b_nquad + ((c0 + col) >> 2)), wd);
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = ((wd[e] >> 8) * (1.0f / 16777216.0f)) >= p.bnb_p ? b_keep : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (b_x[i][e] - b_mu[e]) * b_is[e];
                        float gg = v[e] * f[e];
                        if (p.bnb_relu && !(fmaf(xh, b_sc[e], b_sh[e]) > 0.f)) gg = 0.f;
                        if (e < b_nv) {
                            b_s[e] += gg;
                            b_q[e] = fmaf(gg, xh, b_q[e]);
                            b_gm[e] = fmaxf(b_gm[e], fabsf(gg)), b_xm[e] = fmaxf(b_xm[e], fabsf(xh));
                        }
                    }
                }
            }
            if (p.absmax && row < p.M) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < n_valid) amax = fmaxf(amax, fabsf(v[e]));
            }
            if (row < p.M) {
                float* c = Cb + (int64_t)row * p.ldc + col;
                if (vec_ok && col + 3 < n_valid) {
                    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (vec2_ok && col + 3 < n_valid) {
                    *reinterpret_cast<float2*>(c) = make_float2(v[0], v[1]);
                    *reinterpret_cast<float2*>(c + 2) = make_float2(v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < n_valid) c[e] = v[e];
                }
            }
        }
    };
    static_for<MT / 2>(do_pass);
    if (p.absmax) absmax_publish(wave_absmax(amax), p.absmax);
    if (stats) {        // fold the 8 row groups of the wave (lanes l, l + 8, ..., l + 56 hold the same four columns): fixed order, deterministic
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                st_s[e] += __shfl_xor(st_s[e], o);
                st_q[e] += __shfl_xor(st_q[e], o);
                st_mn[e] = fminf(st_mn[e], __shfl_xor(st_mn[e], o));
                st_mx[e] = fmaxf(st_mx[e], __shfl_xor(st_mx[e], o));
            }
        }
        if (lane < 8) {
            const int64_t F = p.stats_F;
            float* ps = p.stats_part + ((int64_t)tm * 2) * F + c0 + col;
            float* pm = p.stats_minmax + ((int64_t)tm * 2) * F + c0 + col;
            float* pp = p.stats_pivot + (int64_t)tm * F + c0 + col;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < n_valid) ps[e] = st_s[e], ps[F + e] = st_q[e], pm[e] = st_mn[e], pm[F + e] = st_mx[e], pp[e] = piv[e];
        }
    }
    if constexpr (BNB) {       // the same fold: 8 row groups per wave, fixed order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                b_s[e] += __shfl_xor(b_s[e], o);
                b_q[e] += __shfl_xor(b_q[e], o);
                b_gm[e] = fmaxf(b_gm[e], __shfl_xor(b_gm[e], o));
                b_xm[e] = fmaxf(b_xm[e], __shfl_xor(b_xm[e], o));
            }
        }
        if (lane < 8) {
            const int64_t F = p.N;
            float* ps = p.bnb_part + ((int64_t)tm * 2) * F + c0 + col;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < b_nv) ps[e] = b_s[e], ps[F + e] = b_q[e];
            if (p.bnb_pmax) {
                float* pm = p.bnb_pmax + ((int64_t)tm * 2) * F + c0 + col;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < b_nv) pm[e] = b_gm[e], pm[F + e] = b_xm[e];
            }
        }
    }
}

__global__ __launch_bounds__(512) void gemm_halves3_nt64_kernel(H3Args p) { gemm_halves3_nt64_body<false, false>(p, nullptr); }
__global__ __launch_bounds__(512) void gemm_halves3_nt64_frag_kernel(H3Args p) { gemm_halves3_nt64_body<false, true>(p, nullptr); }
__global__ __launch_bounds__(512) void gemm_halves3_nt64_grouped_kernel(H3Args p, H3Groups groups) { gemm_halves3_nt64_body<true, false>(p, &groups); }
__global__ __launch_bounds__(512) void gemm_halves3_nt64_bnb_kernel(H3Args p) { gemm_halves3_nt64_body<false, false, true>(p, nullptr); }
__global__ __launch_bounds__(512) void gemm_halves3_nt64_frag_bnb_kernel(H3Args p) { gemm_halves3_nt64_body<false, true, true>(p, nullptr); }

// BOT_NT_KERNEL=256x32 (default) / 128x64: which of the two forms the NT launches take when both cover the shape; read once per process
static bool nt64_wanted() {
    static const bool wanted = [] {
        const char* e = getenv("BOT_NT_KERNEL");
        return !(e && !strcmp(e, "128x64"));
    }();
    return wanted;
}

// ============================================================================================================================
// TN: the weight gradient  dW[k, p] = sum_n x[n, k] d[n, p]  of two LEFT-layout operands (x: [N, 3 KP], d: [N, 3 PP]; h1 at piece 0,
// 2^11 h2 at piece 2), a reduction over the N ~ 1.7e5 node rows with a small result:
//
//   dW = x1^T d1 + x1^T d2 + x2^T d1 = sum_n  d1 x1 + (2^11 d2)(2^-11 x1) + (2^-11 d1)(2^11 x2)
//
// Same staging idea as the NT kernel (each operand's two halves cross global -> LDS once per step of 32 rows and serve three MFMAs per
// fragment pair; the 2^-11 factors are v_pk_mul_f16 of fragments), but the reduction index runs along the ROWS of both operands, so an
// MFMA fragment (8 consecutive n of one column) is a column segment of a row-major tile: it is read with gfx950's transposing LDS read
// ds_read_b64_tr_b16 (lane t of a 16-lane group passes the address of row t >> 2, columns 4 (t & 3) .. + 3 of a 4 x 16 block and
// receives column t's four rows; probed with integer data, tools/probe_ds_read_tr.hip) — two per fragment.
//
// Decomposition: 192 (k) x 192 (p) output tiles and split-K over row ranges, so that ONE split of the config-2 shape (768 x 1536) is
// 4 x 8 = 32 tiles = the 32 CUs of one XCD: all workgroups that share an x or d tile share an L2, and every operand row leaves HBM once
// (256 x 256 tiles give 18 per split: 42 % L2 hits and 1.28 ms, measured).  Three LDS stages of 48 KB with the DMA two steps ahead and a
// COUNTED vmcnt wait: most loads miss the L2 by design (each row block is new) and need more than one MFMA phase to land (with the DMA
// one step ahead and issued late in the step the 256 x 256 form took 1.83 ms).  LDS image per piece and stage: columns 0 .. 127 as
// [32 rows][256 B] and columns 128 .. 191 as [32 rows][128 B] (a DMA instruction writes 1 KB linearly: 4 or 8 whole rows), the 32-byte
// chunk index XOR-swizzled with key8(n) = (n & 3) | ((n >> 1) & 4) resp. key4(n) = ((n >> 1) & 1) | ((n >> 2) & 2) so that the 8 rows a
// 32-lane half reads at once land on 8 different 32-byte bank groups (zero conflicts, PMC); the swizzle rides on the DMA's per-lane
// SOURCE column.  Every workgroup writes its fp32 partial tile; tn_reduce_h3_kernel adds the splits in split order and applies 1 / (s_x s_d).
constexpr int TBK = 32;                          // rows of n per step
constexpr int TT = 192;                          // tile edge (k and p)
constexpr int kTnSubA = TBK * 256, kTnSubB = TBK * 128;      // bytes: columns 0..127 / 128..191 of one piece
constexpr int kTnPiece = kTnSubA + kTnSubB;      // 12 KB
constexpr int kTnStage = 4 * kTnPiece;           // x1 | x2 | d1 | d2 : 48 KB
constexpr int kTnStages = 3;

struct TnArgs3 {
    const _Float16* X;      // [N, ldx]: x1 at column 0, 2^11 x2 at column x2_off
    const _Float16* D;      // [N, ldd]: d1 at column 0, 2^11 d2 at column d2_off
    float* part;            // [splits][KP][PP] partial sums
    int64_t ldx, ldd;
    int N, K, P, KP, PP;    // K, P: valid columns; KP, PP: piece widths (multiples of 64)
    int x2_off, d2_off;
    int tiles_k, tiles_p, splits, rows_per_split;       // rows_per_split: a multiple of TBK
    int mode;               // 0 = the product; measurement switches (tools/exp_halves3.py): bit 0 no DMA in the loop, bit 1 no barrier / wait
};


// Grouped form (bot_gemm_halves3_tn_grouped_f32): the tiles of a split are a LIST, each with its own x / d columns and output block - the
// per-head weight gradients of the aggregate-first GAT layer and the gradient of its merged projection as ONE launch:
//   out[out_off + k ldo + p] = alpha * sum_n X[n, x_col0 + k] . D[n, d_col0 + p],   k < k_valid <= 192, p < p_valid <= 192
// (transposed: out[out_off + p ldo + k])
struct TnTile {
    int x_col0, d_col0, k_valid, p_valid;
    int64_t out_off, ldo;
    int transposed, pad;
};
constexpr int kMaxTnTiles = 16;
struct TnTiles {
    TnTile t[kMaxTnTiles];
};

__device__ __forceinline__ int tn_key8(int n) { return (n & 3) | ((n >> 1) & 4); }
__device__ __forceinline__ int tn_key4(int n) { return ((n >> 1) & 1) | ((n >> 2) & 2); }

typedef short v4s __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct TrLane {             // per-lane constants of the two transposing reads of a fragment (rows rA = 8 g + (t >> 2) and rA + 4)
    int a0, a1, b0, b1;     // row byte offsets in the 256-byte-pitch / 128-byte-pitch sub-image (+ 8 (t & 3))
    int k80, k81, k40, k41; // swizzle keys of the two rows
};

// byte offset inside a piece of transposing read r (0 / 1) of the fragment of 16-column chunk `chunk` (0 .. 11) for this lane
__device__ __forceinline__ int tr_off(const TrLane& L, int chunk, int r) {
    if (chunk < 8) return (r ? L.a1 : L.a0) + ((chunk ^ (r ? L.k81 : L.k80)) << 5);
    return kTnSubA + (r ? L.b1 : L.b0) + (((chunk - 8) ^ (r ? L.k41 : L.k40)) << 5);
}

// 8 consecutive n (rows 8 g .. 8 g + 7 of the stage) of one column: two transposing reads at per-lane LDS byte addresses a0 / a1 (+ the
// compile-time piece offset).  Inline asm, NOT the __builtin_amdgcn_ds_read_tr16_b64 intrinsic: behind an LDS-DMA the compiler puts an
// s_waitcnt vmcnt(0) in front of every use of that intrinsic (it cannot tell the read from the DMA's destination), which drains the whole
// prefetch pipeline once per tile row (measured: 1.46 -> 2.04 ms when the DMA moved into the MFMA phase).  The caller waits for the reads
// with tr_wait() before the first use.
typedef int v2i __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void tr_issue(v2i& lo, v2i& hi, unsigned a0, unsigned a1) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "n"(OFF));
#endif
}
__device__ __forceinline__ void tr_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);        // no MFMA may be hoisted above the wait (cdna_hip_programming.md rule 18)
#endif
}
__device__ __forceinline__ half8 tr_pack(const v2i& lo, const v2i& hi) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i q = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(half8, q);
}

// PT: columns of d per tile: 192, or 128 (grouped launches whose d blocks are a multiple of 128 wide: 4 instead of 6 tile rows per wave, the
// 128-byte sub-image of d neither fetched nor read)
template <bool GROUPED, int PT, bool ABLATE = false>     // ABLATE: the measurement build, the only one that reads `p.mode` in the kernel
__device__ __forceinline__ void gemm_halves3_tn_body(const TnArgs3& p, const TnTiles* tiles) {
    static_assert(PT == 192 || PT == 128, "tile widths of d");
    constexpr int MTN = PT / 32;            // 16-column tile rows per wave (2 wave rows)
    constexpr int kDma = PT == 192 ? 6 : 5; // DMA instructions per wave and step
    __shared__ __attribute__((aligned(1024))) unsigned char lds[kTnStages * kTnStage];
    // (readfirstlane: the compiler must KNOW the wave index is uniform, or every DMA instruction - whose LDS address goes through M0 - is
    // wrapped in a waterfall loop and every LDS read behind it waits for vmcnt(0))
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks b, b + 8, ... share an XCD: XCD x takes the splits x, x + 8, ... and walks their tiles (all tiles of a split on one L2)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tps = p.tiles_k * p.tiles_p;
    const int split = (j / tps) * 8 + xcd, tile = j % tps;
    if (split >= p.splits) return;
    // the tile's first x / d column and where its 192 x 192 partial goes (plain: the [KP, PP] slab of the split; grouped: tile after tile)
    int xc = (tile / p.tiles_p) * TT, dc = (tile % p.tiles_p) * TT;
    if constexpr (GROUPED) xc = tiles->t[tile].x_col0, dc = tiles->t[tile].d_col0;
    const int row0 = split * p.rows_per_split;
    const int rows = min(p.rows_per_split, p.N - row0);          // > 0 by construction of the splits
    const int T = (rows + TBK - 1) / TBK;

    // LDS-DMA plan: 48 instructions of 1 KB per step (12 per piece: 8 x [4 rows x 256 B], 4 x [8 rows x 128 B]), six per wave.  Rows past the operand's end (the last split's last step) are outside the descriptor and arrive as zeros.
    const _Float16* tileX = p.X + (int64_t)row0 * p.ldx + xc;
    const _Float16* tileD = p.D + (int64_t)row0 * p.ldd + dc;
    // (plain integer arithmetic: HIP's min<int64_t> goes through double, which makes the descriptor a VGPR value -> waterfall loops)
    const int64_t bytesX = ((int64_t)rows * p.ldx - xc) * 2, bytesD = ((int64_t)rows * p.ldd - dc) * 2;
    const uint32_t limX = bytesX > 0x7fffffff ? 0x7fffffffu : (uint32_t)bytesX;
    const uint32_t limD = bytesD > 0x7fffffff ? 0x7fffffffu : (uint32_t)bytesD;
    // Instruction i of wave w:  i = 0 .. 3: rows 4 w .. 4 w + 3 of the 256-byte sub-image of piece i (x1, x2, d1, d2);  i = 4 / 5: rows
    // 8 (w & 3) .. + 7 of the 128-byte sub-image of x piece (w >> 2) / d piece (w >> 2).  WHICH operand an instruction reads is a
    // compile-time property of i (its resource descriptor must be known uniform: a descriptor picked by a run-time select is a VGPR value
    // and the compiler wraps the load in a waterfall loop).
    uint32_t voff[6];       // per-lane source byte offset
    int loff[6];            // LDS byte offset inside a stage (wave-uniform)
#pragma unroll
    for (int i = 0; i < kDma; ++i) {
        const bool isx = i < 2 || i == 4;
        int q, row, col, lo;
        if (i < 4) {
            q = i;
            row = 4 * w + (lane >> 4);
            const int sl = lane & 15;
            col = (((sl >> 1) ^ tn_key8(row)) << 4) + (sl & 1) * 8;
            lo = q * kTnPiece + w * 1024;
        } else {
            q = (i == 4 ? 0 : 2) + (w >> 2);
            row = 8 * (w & 3) + (lane >> 3);
            const int sl = lane & 7;
            col = 128 + (((sl >> 1) ^ tn_key4(row)) << 4) + (sl & 1) * 8;
            lo = q * kTnPiece + kTnSubA + (w & 3) * 1024;
        }
        const int64_t ld = isx ? p.ldx : p.ldd;
        const int po = (q & 1) ? (isx ? p.x2_off : p.d2_off) : 0;
        voff[i] = (uint32_t)((int64_t)row * ld + po + col) * 2;
        loff[i] = lo;
    }
    const int stepX = TBK * (int)p.ldx * 2, stepD = TBK * (int)p.ldd * 2;
    auto issue_one = [&](int i, int stage, int kt) {
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned char* base = lds + stage * kTnStage;
        if (i < 2 || i == 4) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(tileX), 0, (int)limX, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(base + loff[i]), 16, voff[i], kt * stepX, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(tileD), 0, (int)limD, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(base + loff[i]), 16, voff[i], kt * stepD, 0, 0);
        }
#endif
    };
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < kDma; ++i) issue_one(i, stage, kt);
    };

    // fragment addressing: lane l = (t = l & 15, g = l >> 4) wants rows 8 g .. 8 g + 7 of column t of a 16-column chunk
    const int wr = w >> 2, wc = w & 3;                    // wave tile: 96 of the p columns (wr: 6 chunks) x 48 of the k columns (wc: 3 chunks)
    const int t = lane & 15, g = lane >> 4;
    const int rA = 8 * g + (t >> 2), rB = rA + 4, sub = 8 * (t & 3);
    TrLane L;
    L.a0 = rA * 256 + sub, L.a1 = rB * 256 + sub, L.b0 = rA * 128 + sub, L.b1 = rB * 128 + sub;
    L.k80 = tn_key8(rA), L.k81 = tn_key8(rB), L.k40 = tn_key4(rA), L.k41 = tn_key4(rB);
    // the swizzle makes a fragment's address lane-dependent in a way no immediate can carry: 18 per-lane offsets, computed once
    int xo[3][2], dofs[MTN][2];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) xo[nt][0] = tr_off(L, wc * 3 + nt, 0), xo[nt][1] = tr_off(L, wc * 3 + nt, 1);
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt) dofs[mt][0] = tr_off(L, wr * MTN + mt, 0), dofs[mt][1] = tr_off(L, wr * MTN + mt, 1);

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    f32x4 acc[MTN][3];                                      // [mt: 16 p columns each][nt: 16 k columns each]
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const _Float16 sh = (_Float16)(1.0f / kHalvesShift);

    issue(0, 0);
    if (T > 1) issue(1, 1);
    if (T > 1) {                                                      // step 0 landed, step 1 may still be in flight
        if constexpr (kDma == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kt = 0; kt < T; ++kt) {
        const int s2 = stage >= 1 ? stage - 1 : 2;        // (stage + 2) % 3: free since the barrier that ended step kt - 1
        const bool ahead = kt + 2 < T && !(ABLATE && (p.mode & 1));      // this step issues the DMA of step kt + 2, one instruction per tile row: six
        // at once right behind the barrier cost every wave ~900 cycles of issue before its first MFMA (0.55 of 1.46 ms, ablation)
        // the k-column fragments (x: the MFMA's B operand after the swap) of this wave's 48 columns: x1, 2^11 x2, 2^-11 x1
        const unsigned sb = lds_base + stage * kTnStage;
        v2i xl[3][2], xh[3][2], dl[2][2], dh[2][2];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            tr_issue<0>(xl[nt][0], xh[nt][0], sb + xo[nt][0], sb + xo[nt][1]);
            tr_issue<kTnPiece>(xl[nt][1], xh[nt][1], sb + xo[nt][0], sb + xo[nt][1]);
        }
        tr_issue<2 * kTnPiece>(dl[0][0], dh[0][0], sb + dofs[0][0], sb + dofs[0][1]);
        tr_issue<3 * kTnPiece>(dl[0][1], dh[0][1], sb + dofs[0][0], sb + dofs[0][1]);
        tr_wait();
        half8 x1[3], x2[3], x1s[3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            x1[nt] = tr_pack(xl[nt][0], xh[nt][0]);
            x2[nt] = tr_pack(xl[nt][1], xh[nt][1]);
            x1s[nt] = x1[nt] * sh;
        }
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
            const half8 d1 = tr_pack(dl[mt & 1][0], dh[mt & 1][0]);
            const half8 d2 = tr_pack(dl[mt & 1][1], dh[mt & 1][1]);
            const half8 d1s = d1 * sh;
            if (mt + 1 < MTN) {               // the next tile row's fragments, requested before this row's MFMAs
                tr_issue<2 * kTnPiece>(dl[(mt + 1) & 1][0], dh[(mt + 1) & 1][0], sb + dofs[mt + 1][0], sb + dofs[mt + 1][1]);
                tr_issue<3 * kTnPiece>(dl[(mt + 1) & 1][1], dh[(mt + 1) & 1][1], sb + dofs[mt + 1][0], sb + dofs[mt + 1][1]);
            }
            if (ahead) {
                issue_one(mt, s2, kt + 2);
                if (kDma > MTN && mt == 0) issue_one(MTN, s2, kt + 2);      // (five instructions over four tile rows)
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                // D[i = p column][j = k column]: a lane holds 4 consecutive p of one k (float4 stores along p)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1, x1[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d2, x1s[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d1s, x2[nt], acc[mt][nt], 0, 0, 0);
            }
            if (mt + 1 < MTN) tr_wait();
        }
        // step kt + 1 must have landed before anyone reads it; step kt + 2 (this wave's 6 youngest instructions) may stay in flight
        if (!(ABLATE && (p.mode & 2))) {
            if (ahead) {
                if constexpr (kDma == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    // acc[mt][nt][r] = partial of dW[k = xc + wc 48 + nt 16 + (lane & 15)][p = dc + wr 96 + mt 16 + (lane >> 4) 4 + r]
    if constexpr (GROUPED) {
        float* out = p.part + ((int64_t)split * tps + tile) * (TT * PT);       // the tile's own [192][PT] block
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int k = wc * 48 + nt * 16 + (lane & 15);
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) {
                const int pc = wr * (PT / 2) + mt * 16 + (lane >> 4) * 4;
                *reinterpret_cast<float4*>(out + k * PT + pc) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
            }
        }
    } else {
        float* out = p.part + (int64_t)split * p.KP * p.PP;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int k = xc + wc * 48 + nt * 16 + (lane & 15);
            if (k >= p.KP) continue;
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) {
                const int pc = dc + wr * 96 + mt * 16 + (lane >> 4) * 4;
                if (pc < p.PP) *reinterpret_cast<float4*>(out + (int64_t)k * p.PP + pc) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
            }
        }
    }
}

__global__ __launch_bounds__(512) void gemm_halves3_tn_kernel(TnArgs3 p) { gemm_halves3_tn_body<false, 192>(p, nullptr); }
__global__ __launch_bounds__(512) void gemm_halves3_tn_ablate_kernel(TnArgs3 p) { gemm_halves3_tn_body<false, 192, true>(p, nullptr); }
template <int PT>
__global__ __launch_bounds__(512) void gemm_halves3_tn_grouped_kernel(TnArgs3 p, TnTiles tiles) { gemm_halves3_tn_body<true, PT>(p, &tiles); }

// grouped: out[out_off + k ldo + p] = scale_x[1] scale_d[1] * sum_s part[s][tile][k][p]   (split order), k < k_valid, p < p_valid
__global__ __launch_bounds__(256) void tn_reduce_h3_grouped_kernel(const float* part, int splits, int n_tiles, TnTiles tiles, const float* scale_x,
                                                                   const float* scale_d, float* out, int PT) {
    const int P4 = PT / 4;
    const int tile = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= TT * P4) return;
    const TnTile& t = tiles.t[tile];
    const int k = i / P4, pc = (i - k * P4) * 4;
    if (k >= t.k_valid || pc >= t.p_valid) return;
    const float alpha = scale_x[1] * scale_d[1];
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = part + (int64_t)tile * (TT * PT) + k * PT + pc;
    const int64_t sstride = (int64_t)n_tiles * (TT * PT);
    constexpr int U = 4;                                    // loads in flight; added in split order
    for (int s0 = 0; s0 < splits; s0 += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s0 + u < splits ? *reinterpret_cast<const float4*>(src + (s0 + u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; ++u) s4.x += v[u].x, s4.y += v[u].y, s4.z += v[u].z, s4.w += v[u].w;
    }
    const float r[4] = {s4.x * alpha, s4.y * alpha, s4.z * alpha, s4.w * alpha};
    if (t.transposed) {
        float* o = out + t.out_off + (int64_t)pc * t.ldo + k;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (pc + e < t.p_valid) o[e * t.ldo] = r[e];
    } else {
        float* o = out + t.out_off + (int64_t)k * t.ldo + pc;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (pc + e < t.p_valid) o[e] = r[e];
    }
}

// out[k, p] = scale_x[1] scale_d[1] * sum_s part[s][k][p]   (split order), k < K, p < P
// (scale_d2: d's columns from p2 on - a multiple of 4 - were stored under a second scale, bot_gemm_halves3_tn2_f32)
__global__ __launch_bounds__(256) void tn_reduce_h3_kernel(const float* part, int splits, int K, int P, int KP, int PP, const float* scale_x,
                                                           const float* scale_d, float* out, int64_t ldo, const float* scale_d2, int p2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int p4 = (P + 3) >> 2;
    if (i >= (int64_t)K * p4) return;
    const int k = (int)(i / p4), pc = (int)(i - (int64_t)k * p4) * 4;
    const float alpha = scale_x[1] * ((scale_d2 && pc >= p2) ? scale_d2[1] : scale_d[1]);
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = part + (int64_t)k * PP + pc;
    constexpr int U = 4;                                    // loads in flight; added in split order
    for (int s0 = 0; s0 < splits; s0 += U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s0 + u < splits ? *reinterpret_cast<const float4*>(src + (int64_t)(s0 + u) * KP * PP) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; ++u) s4.x += v[u].x, s4.y += v[u].y, s4.z += v[u].z, s4.w += v[u].w;
    }
    float* o = out + (int64_t)k * ldo + pc;
    const float r[4] = {s4.x * alpha, s4.y * alpha, s4.z * alpha, s4.w * alpha};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (pc + e < P) o[e] = r[e];
}

template <int BM, int BN, int WM, int WN, bool PIPE, bool ABLATE>
void launch_h3(H3Args p, int64_t m, int64_t n, hipStream_t st) {
    p.tiles_m = (int)((m + BM - 1) / BM), p.tiles_n = (int)((n + BN - 1) / BN);
    const int groups = (p.tiles_m + 7) / 8;
    hipLaunchKernelGGL((gemm_halves3_nt_kernel<BM, BN, WM, WN, PIPE, ABLATE>), dim3(groups * 8 * p.tiles_n), dim3(WM * WN * 64), 0, st, p);
}

}  // namespace
}  // namespace bot

extern "C" int bot_gemm_halves3_nt_f32(int64_t m, int64_t n, int64_t k, const float* scale_a, const float* scale_b, const uint16_t* A, int64_t lda,
                                       int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, float* C, int64_t ldc, int32_t mode,
                                       bot_stream_t stream) {
    return bot_gemm_halves3_nt2_f32(m, n, k, scale_a, nullptr, 0, scale_b, A, lda, a2_off, B, ldb, b2_off, 0, C, ldc, mode, stream);
}

extern "C" int bot_gemm_halves3_nt2_f32(int64_t m, int64_t n, int64_t k, const float* scale_a, const float* scale_a2, int64_t k_split, const float* scale_b,
                                        const uint16_t* A, int64_t lda, int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, int32_t b_layout,
                                        float* C, int64_t ldc, int32_t mode, bot_stream_t stream) {
    return bot_gemm_halves3_nt3_f32(m, n, k, scale_a, scale_a2, k_split, scale_b, A, lda, a2_off, B, ldb, b2_off, b_layout, C, ldc, nullptr, mode, stream);
}

extern "C" int32_t bot_gemm_halves3_nt_bn_rows(int64_t k) {
    using namespace bot;
    return (k > 0 && k % (2 * BK) == 0 && nt64_wanted()) ? 256 : 0;
}

extern "C" int bot_gemm_halves3_nt3_f32(int64_t m, int64_t n, int64_t k, const float* scale_a, const float* scale_a2, int64_t k_split, const float* scale_b,
                                        const uint16_t* A, int64_t lda, int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, int32_t b_layout,
                                        float* C, int64_t ldc, const bot_bn_bwd_stats_t* bn, int32_t mode, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(bn == nullptr || (mode == 0 && bot_gemm_halves3_nt_bn_rows(k) == 256), -1,
                "gemm_halves3_nt3: the BatchNorm-backward by-product rides on the 256 x 32 form (mode 0, k a multiple of 64, BOT_NT_KERNEL not 128x64)");
    BOT_REQUIRE(bn == nullptr || (bn->x && bn->mean && bn->invstd && bn->part), -1, "gemm_halves3_nt3: bn->x, mean, invstd and part must be set");
    BOT_REQUIRE(bn == nullptr || (n % 2 == 0 && bn->ldx >= n && bn->ldx % 2 == 0 && aligned(bn->x, 8) && bn->p >= 0.f && bn->p < 1.f), -1,
                "gemm_halves3_nt3: the by-product needs an even n (= the BatchNorm width), 8-byte aligned x rows of pitch >= n and p in [0, 1)");
    BOT_REQUIRE(b_layout == 0 || (b_layout == 1 && mode == 0 && k % 64 == 0), -1,
                "gemm_halves3_nt2: b_layout 1 (fragment-major B, bot_halves_split_frag_f16 with piece = k) needs mode 0 and k a multiple of 64");
    BOT_REQUIRE(scale_a2 == nullptr || (k_split > 0 && k_split < k && k_split % BK == 0), -1,
                "gemm_halves3_nt2: the second scale starts at a column that is a positive multiple of %d below k (got %lld of %lld)", BK, (long long)k_split,
                (long long)k);
    BOT_REQUIRE(m > 0 && n > 0 && k > 0 && k % BK == 0, -1, "gemm_halves3_nt: m, n > 0 and k a positive multiple of %d (got %lld %lld %lld)", BK,
                (long long)m, (long long)n, (long long)k);
    BOT_REQUIRE(scale_a && scale_b && A && B && C, -1, "gemm_halves3_nt: null pointer");
    BOT_REQUIRE(aligned(A, 16) && aligned(B, 16) && lda % 8 == 0 && ldb % 8 == 0 && a2_off % 8 == 0 && b2_off % 8 == 0, -1,
                "gemm_halves3_nt: operands, row pitches and piece offsets must be 16-byte aligned");
    BOT_REQUIRE(a2_off + k <= lda && (b_layout == 1 || b2_off + k <= ldb) && ldc >= n && m < (1ll << 31) - 256 && n < (1ll << 31) - 256, -1, "gemm_halves3_nt: bad pitches");
    BOT_REQUIRE(b_layout == 0 || ((n + 15) / 16) * (k / 32) * 2048 < (1ll << 31), -1, "gemm_halves3_nt2: fragment-major B exceeds the 2 GiB a buffer descriptor spans");
    H3Args p;
    p.A = reinterpret_cast<const _Float16*>(A), p.B = reinterpret_cast<const _Float16*>(B), p.scale_a = scale_a, p.scale_b = scale_b, p.C = C;
    p.lda = lda, p.ldb = ldb, p.ldc = ldc, p.M = (int)m, p.N = (int)n, p.K = (int)k, p.a2_off = (int)a2_off, p.b2_off = (int)b2_off;
    p.tiles_m = p.tiles_n = 0;
    p.col_scale = p.col_shift = nullptr, p.relu = 0, p.absmax = nullptr;
    p.scale_a2 = scale_a2, p.k2 = scale_a2 ? (int)(k_split / BK) : -1;
    p.b_frag = b_layout;
    p.stats_part = p.stats_minmax = nullptr, p.stats_pivot = nullptr, p.stats_F = 0;
    p.bnb_x = nullptr, p.bnb_ldx = 0, p.bnb_mean = p.bnb_invstd = p.bnb_w = p.bnb_b = nullptr, p.bnb_relu = 0, p.bnb_p = 0.f, p.bnb_seed = 0;
    p.bnb_seed_offset = nullptr, p.bnb_part = p.bnb_pmax = nullptr;
    p.mode = mode;
    const bool force_128x64 = (mode & 1024) != 0;            // (tools: both forms in one process)
    mode &= ~1024;
    p.mode = mode;
    if (bn) {
        p.bnb_x = bn->x, p.bnb_ldx = bn->ldx, p.bnb_mean = bn->mean, p.bnb_invstd = bn->invstd, p.bnb_w = bn->weight, p.bnb_b = bn->bias;
        p.bnb_relu = bn->relu, p.bnb_p = bn->p, p.bnb_seed = bn->seed, p.bnb_seed_offset = bn->seed_offset, p.bnb_part = bn->part, p.bnb_pmax = bn->pmax;
        p.tiles_m = (int)((m + 255) / 256), p.tiles_n = (int)((n + 255) / 256);
        set_kernel(b_layout ? "bot::gemm_halves3_nt64_frag_bnb_kernel" : "bot::gemm_halves3_nt64_bnb_kernel");
        if (b_layout) hipLaunchKernelGGL(gemm_halves3_nt64_frag_bnb_kernel, dim3(((p.tiles_m + 7) / 8) * 8 * p.tiles_n), dim3(512), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(gemm_halves3_nt64_bnb_kernel, dim3(((p.tiles_m + 7) / 8) * 8 * p.tiles_n), dim3(512), 0, (hipStream_t)stream, p);
    } else if (mode == 0 && !force_128x64 && nt64_wanted() && (k / BK) % 2 == 0) {      // the 128-byte-line form: an even number of k-steps
        p.tiles_m = (int)((m + 255) / 256), p.tiles_n = (int)((n + 255) / 256);
        set_kernel(b_layout ? "bot::gemm_halves3_nt64_frag_kernel" : "bot::gemm_halves3_nt64_kernel");
        if (b_layout) hipLaunchKernelGGL(gemm_halves3_nt64_frag_kernel, dim3(((p.tiles_m + 7) / 8) * 8 * p.tiles_n), dim3(512), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(gemm_halves3_nt64_kernel, dim3(((p.tiles_m + 7) / 8) * 8 * p.tiles_n), dim3(512), 0, (hipStream_t)stream, p);
    } else if (b_layout) {
        BOT_REQUIRE(false, -1, "gemm_halves3_nt2: a fragment-major B is read by the 256 x 32 form only (BOT_NT_KERNEL=128x64 / mode bit 1024 exclude it)");
    } else if (mode & 32) {        // measurement builds only: one barrier at the END of a k-step, with the ablation switches
        set_kernel("bot::gemm_halves3_nt_kernel<256,256,2,4,plain>");
        launch_h3<256, 256, 2, 4, false, true>(p, m, n, (hipStream_t)stream);
    } else if (mode) {      // the pipelined loop with `mode` read in the kernel (bit 0: no output stores; bits 2 / 3: an operand never advances)
        set_kernel("bot::gemm_halves3_nt_kernel<256,256,2,4,pipelined,ablate>");
        launch_h3<256, 256, 2, 4, true, true>(p, m, n, (hipStream_t)stream);
    } else {                // the product: no switch inside the kernel
        set_kernel("bot::gemm_halves3_nt_kernel<256,256,2,4,pipelined>");
        launch_h3<256, 256, 2, 4, true, false>(p, m, n, (hipStream_t)stream);
    }
    return hip_status("gemm_halves3_nt");
}

extern "C" int bot_gemm_halves3_nt_grouped_f32(int64_t m, int64_t b_rows, const float* scale_a, const float* scale_b, const uint16_t* A, int64_t lda,
                                               int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, float* C, int64_t ldc, int32_t n_groups,
                                               const int64_t* groups, int32_t k_seg, const float* col_scale, const float* col_shift, int32_t relu,
                                               uint32_t* absmax_slots, int32_t mode, bot_stream_t stream) {
    return bot_gemm_halves3_nt_grouped2_f32(m, b_rows, scale_a, scale_b, A, lda, a2_off, B, ldb, b2_off, C, ldc, n_groups, groups, k_seg, col_scale, col_shift,
                                            relu, absmax_slots, nullptr, nullptr, nullptr, 0, mode, stream);
}

extern "C" int bot_gemm_halves3_nt_grouped2_f32(int64_t m, int64_t b_rows, const float* scale_a, const float* scale_b, const uint16_t* A, int64_t lda,
                                                int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, float* C, int64_t ldc, int32_t n_groups,
                                                const int64_t* groups, int32_t k_seg, const float* col_scale, const float* col_shift, int32_t relu,
                                                uint32_t* absmax_slots, float* stats_part, float* stats_minmax, float* stats_pivot, int32_t stats_F,
                                                int32_t mode, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE((stats_part == nullptr) == (stats_minmax == nullptr) && (stats_part == nullptr) == (stats_pivot == nullptr) && (stats_part == nullptr || stats_F >= 1), -1,
                "gemm_halves3_nt_grouped2: stats_part, stats_minmax and stats_pivot go together (stats_F = their row width)");
    BOT_REQUIRE(m > 0 && b_rows > 0 && n_groups >= 1 && n_groups <= kMaxGroups && k_seg >= 0, -1, "gemm_halves3_nt_grouped: m, b_rows > 0, 1 .. %d groups (got %lld %lld %d)",
                kMaxGroups, (long long)m, (long long)b_rows, n_groups);
    BOT_REQUIRE(scale_a && scale_b && A && B && C && groups, -1, "gemm_halves3_nt_grouped: null pointer");
    BOT_REQUIRE(aligned(A, 16) && aligned(B, 16) && lda % 8 == 0 && ldb % 8 == 0 && a2_off % 8 == 0 && b2_off % 8 == 0, -1,
                "gemm_halves3_nt_grouped: operands, row pitches and piece offsets must be 16-byte aligned");
    BOT_REQUIRE(m < (1ll << 31) - 256 && b_rows < (1ll << 31) - 256 && 256 * lda * 2 < (1ll << 31) && 256 * ldb * 2 < (1ll << 31), -1, "gemm_halves3_nt_grouped: bad sizes");
    H3Args p;
    p.A = reinterpret_cast<const _Float16*>(A), p.B = reinterpret_cast<const _Float16*>(B), p.scale_a = scale_a, p.scale_b = scale_b, p.C = C;
    p.lda = lda, p.ldb = ldb, p.ldc = ldc, p.M = (int)m, p.N = (int)b_rows, p.K = 0, p.a2_off = (int)a2_off, p.b2_off = (int)b2_off;
    p.mode = 0;             // (the grouped kernel is a production instantiation: no measurement switch inside)
    p.scale_a2 = nullptr, p.k2 = -1, p.b_frag = 0;
    p.stats_part = stats_part, p.stats_minmax = stats_minmax, p.stats_pivot = stats_pivot, p.stats_F = stats_F;
    p.col_scale = col_scale, p.col_shift = col_shift, p.relu = relu, p.absmax = absmax_slots;
    H3Groups g;
    g.k_seg = k_seg;
    for (int i = 0; i < n_groups; ++i) {
        const int64_t* d = groups + 6 * i;       // b_row0, n_valid, a_col0, a_col1, k_steps, c_off
        const int64_t k_steps = d[4], a_hi = std::max(k_seg > 0 ? d[2] + 32 * std::min<int64_t>(k_seg, k_steps) : 0, k_steps > k_seg ? d[3] + 32 * k_steps : 0);
        BOT_REQUIRE(d[0] >= 0 && d[0] < b_rows && d[1] >= 1 && d[1] <= 256 && k_steps >= 1 && d[2] >= 0 && d[3] >= 0 && d[2] % 8 == 0 && d[3] % 8 == 0 && d[5] >= 0, -1,
                    "gemm_halves3_nt_grouped: group %d: b_row0=%lld n_valid=%lld a_col0=%lld a_col1=%lld k_steps=%lld c_off=%lld", i, (long long)d[0],
                    (long long)d[1], (long long)d[2], (long long)d[3], (long long)d[4], (long long)d[5]);
        BOT_REQUIRE(a_hi + a2_off <= lda && 32 * k_steps + b2_off <= ldb && 32 * k_steps <= b2_off, -1,
                    "gemm_halves3_nt_grouped: group %d reads past a row (A columns to %lld + %lld of %lld, B columns to %lld + %lld of %lld)", i, (long long)a_hi,
                    (long long)a2_off, (long long)lda, (long long)(32 * k_steps), (long long)b2_off, (long long)ldb);
        g.g[i] = H3Group{(int)d[0], (int)d[1], (int)d[2] * 2, (int)d[3] * 2, (int)k_steps, 0, d[5]};
    }
    p.tiles_m = (int)((m + 255) / 256), p.tiles_n = n_groups;
    bool even = k_seg % 2 == 0;
    for (int i = 0; i < n_groups; ++i) even = even && g.g[i].k_steps % 2 == 0;
    BOT_REQUIRE(stats_part == nullptr || (!(mode & 1024) && even && nt64_wanted()), -1,
                "gemm_halves3_nt_grouped2: the statistics by-product exists in the 256 x 32 form only (even k-step counts, BOT_NT_KERNEL != 128x64)");
    if (!(mode & 1024) && even && nt64_wanted()) {
        set_kernel("bot::gemm_halves3_nt64_grouped_kernel");
        hipLaunchKernelGGL(gemm_halves3_nt64_grouped_kernel, dim3(((p.tiles_m + 7) / 8) * 8 * n_groups), dim3(512), 0, (hipStream_t)stream, p, g);
        return hip_status("gemm_halves3_nt_grouped");
    }
    set_kernel("bot::gemm_halves3_nt_grouped_kernel<256,256,2,4,pipelined>");
    hipLaunchKernelGGL((gemm_halves3_nt_grouped_kernel<256, 256, 2, 4, true>), dim3(((p.tiles_m + 7) / 8) * 8 * n_groups), dim3(512), 0, (hipStream_t)stream, p, g);
    return hip_status("gemm_halves3_nt_grouped");
}

namespace bot {
namespace {
// splits: one per XCD (8) while a split keeps >= 4096 rows, fewer for short operands - and, round 5, q splits per XCD where the tile count
// of a split does not fill an XCD's 32 CUs (config 2: 4 x 8 = 32 tiles, q = 1; S-products' [480, N] x [N, 968]: 3 x 6 = 18 tiles used
// 144 of 256 CUs: q = 7 -> 126 workgroups per XCD in four rounds, 98 % of the slots; 11.4 -> 8.4 ms): the q <= 8 with the best
// (q tiles) / (32 ceil(q tiles / 32)), the smallest on ties, each split still >= 4096 rows
int tn_splits(int64_t n_rows, int tiles) {
    const int64_t by_rows = n_rows / 4096;
    if (by_rows < 8) return (int)std::max<int64_t>(1, by_rows);
    int best_q = 1;
    double best = 0.0;
    for (int q = 1; q <= 8 && 8 * q <= by_rows; ++q) {
        const int wg = q * tiles, rounds = (wg + 31) / 32;
        const double eff = (double)wg / (32.0 * rounds);
        if (eff > best + 1e-9) best = eff, best_q = q;
    }
    return 8 * best_q;
}
}  // namespace
}  // namespace bot

namespace bot {
namespace {
// grouped: as many whole rounds of 8 splits (one per XCD) as keep an XCD's workgroups (splits / 8 x tiles) within its 32 CUs, each
// split >= 1024 rows
int tn_grouped_splits(int64_t n_rows, int n_tiles) {
    // (round 6: config 2's 12-tile launch runs 16 splits = 192 workgroups, 24 of an XCD's 32 CUs; 4 / 5 / 8 splits per XCD - 96 % / 94 % /
    // 100 % of the slots in two or three rounds - gave 10.57 / 10.55 / 10.55 ms per step against 10.49: more partial tiles, shorter splits)
    const int per_xcd = std::max(1, 32 / n_tiles);
    return (int)std::max<int64_t>(1, std::min<int64_t>(8 * per_xcd, n_rows / 1024));
}
}  // namespace
}  // namespace bot

extern "C" int64_t bot_gemm_halves3_tn_grouped_workspace_floats(int64_t n_rows, int32_t n_tiles) {
    return (int64_t)bot::tn_grouped_splits(n_rows, n_tiles) * n_tiles * bot::TT * bot::TT;
}

extern "C" int bot_gemm_halves3_tn_grouped_f32(int64_t n_rows, const float* scale_x, const float* scale_d, const uint16_t* X, int64_t ldx, int64_t x2_off,
                                               const uint16_t* D, int64_t ldd, int64_t d2_off, float* out, int32_t n_tiles, const int64_t* tiles,
                                               float* workspace, int32_t mode, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_rows > 0 && n_tiles >= 1 && n_tiles <= kMaxTnTiles, -1, "gemm_halves3_tn_grouped: need n > 0 and 1 .. %d tiles (got %lld, %d)", kMaxTnTiles,
                (long long)n_rows, n_tiles);
    BOT_REQUIRE(scale_x && scale_d && X && D && out && workspace && tiles, -1, "gemm_halves3_tn_grouped: null pointer");
    BOT_REQUIRE(aligned(X, 16) && aligned(D, 16) && ldx % 8 == 0 && ldd % 8 == 0 && x2_off % 8 == 0 && d2_off % 8 == 0 && x2_off < ldx && d2_off < ldd &&
                    n_rows < (1ll << 31) - 4096, -1, "gemm_halves3_tn_grouped: bad alignment or pitches");
    TnArgs3 a;
    TnTiles tl;
    for (int i = 0; i < n_tiles; ++i) {
        const int64_t* d = tiles + 7 * i;        // x_col0, k_valid, d_col0, p_valid, out_off, ldo, transposed
        BOT_REQUIRE(d[0] >= 0 && d[0] % 8 == 0 && d[2] >= 0 && d[2] % 8 == 0 && d[1] >= 1 && d[1] <= TT && d[3] >= 1 && d[3] <= TT && d[4] >= 0 &&
                        d[5] >= (d[6] ? d[1] : d[3]) && d[0] + d[1] + x2_off <= ldx && d[2] + d[3] + d2_off <= ldd, -1,
                    "gemm_halves3_tn_grouped: tile %d: x_col0=%lld k_valid=%lld d_col0=%lld p_valid=%lld out_off=%lld ldo=%lld", i, (long long)d[0], (long long)d[1],
                    (long long)d[2], (long long)d[3], (long long)d[4], (long long)d[5]);
        tl.t[i] = TnTile{(int)d[0], (int)d[2], (int)d[1], (int)d[3], d[4], d[5], d[6] != 0, 0};
    }
    a.X = reinterpret_cast<const _Float16*>(X), a.D = reinterpret_cast<const _Float16*>(D), a.part = workspace, a.ldx = ldx, a.ldd = ldd;
    a.N = (int)n_rows, a.K = a.P = a.KP = a.PP = 0, a.x2_off = (int)x2_off, a.d2_off = (int)d2_off;
    a.tiles_k = n_tiles, a.tiles_p = 1;
    int splits = tn_grouped_splits(n_rows, n_tiles);
    const int rps = (int)(((n_rows + splits - 1) / splits + TBK - 1) / TBK * TBK);
    splits = (int)((n_rows + rps - 1) / rps);             // no empty split (never more than tn_grouped_splits: the workspace holds them)
    a.splits = splits, a.rows_per_split = rps;
    a.mode = mode;
    BOT_REQUIRE((int64_t)rps * std::max(ldx, ldd) * 2 < (1ll << 31), -1, "gemm_halves3_tn_grouped: a split of %d rows exceeds the 2 GiB a buffer descriptor spans", rps);
    int pmax = 0;
    for (int i = 0; i < n_tiles; ++i) pmax = std::max(pmax, tl.t[i].p_valid);
    const int PT = pmax <= 128 ? 128 : TT;      // narrow d blocks: 192 x 128 tiles
    set_kernel("bot::gemm_halves3_tn_grouped_kernel<%d>", PT);
    if (PT == 128) hipLaunchKernelGGL(gemm_halves3_tn_grouped_kernel<128>, dim3(((splits + 7) / 8) * 8 * n_tiles), dim3(512), 0, (hipStream_t)stream, a, tl);
    else hipLaunchKernelGGL(gemm_halves3_tn_grouped_kernel<192>, dim3(((splits + 7) / 8) * 8 * n_tiles), dim3(512), 0, (hipStream_t)stream, a, tl);
    hipLaunchKernelGGL(tn_reduce_h3_grouped_kernel, dim3((TT * (PT / 4) + 255) / 256, n_tiles), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, splits,
                       (int)n_tiles, tl, scale_x, scale_d, out, PT);
    return hip_status("gemm_halves3_tn_grouped");
}

extern "C" int64_t bot_gemm_halves3_tn_workspace_floats(int64_t n_rows, int64_t kp, int64_t pp) {
    const int tiles = (int)(((kp + bot::TT - 1) / bot::TT) * ((pp + bot::TT - 1) / bot::TT));
    return (int64_t)bot::tn_splits(n_rows, tiles) * kp * pp;
}

extern "C" int bot_gemm_halves3_tn_f32(int64_t n_rows, int64_t k, int64_t p, int64_t kp, int64_t pp, const float* scale_x, const float* scale_d,
                                       const uint16_t* X, int64_t ldx, int64_t x2_off, const uint16_t* D, int64_t ldd, int64_t d2_off, float* out,
                                       int64_t ldo, float* workspace, int32_t mode, bot_stream_t stream) {
    return bot_gemm_halves3_tn2_f32(n_rows, k, p, kp, pp, scale_x, scale_d, nullptr, 0, X, ldx, x2_off, D, ldd, d2_off, out, ldo, workspace, mode, stream);
}

extern "C" int bot_gemm_halves3_tn2_f32(int64_t n_rows, int64_t k, int64_t p, int64_t kp, int64_t pp, const float* scale_x, const float* scale_d,
                                        const float* scale_d2, int64_t p_split, const uint16_t* X, int64_t ldx, int64_t x2_off, const uint16_t* D, int64_t ldd,
                                        int64_t d2_off, float* out, int64_t ldo, float* workspace, int32_t mode, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(scale_d2 == nullptr || (p_split > 0 && p_split < pp && p_split % 4 == 0), -1,
                "gemm_halves3_tn2: the second scale starts at a column that is a positive multiple of 4 below pp (got %lld of %lld)", (long long)p_split, (long long)pp);
    BOT_REQUIRE(n_rows > 0 && k > 0 && p > 0 && kp >= k && pp >= p && kp % 64 == 0 && pp % 64 == 0, -1,
                "gemm_halves3_tn: need n, k, p > 0 and piece widths >= k, p that are multiples of 64");
    BOT_REQUIRE(scale_x && scale_d && X && D && out && workspace, -1, "gemm_halves3_tn: null pointer");
    BOT_REQUIRE(aligned(X, 16) && aligned(D, 16) && ldx % 8 == 0 && ldd % 8 == 0 && x2_off % 8 == 0 && d2_off % 8 == 0 && x2_off + kp <= ldx &&
                    d2_off + pp <= ldd && ldo >= p && n_rows < (1ll << 31) - 4096, -1, "gemm_halves3_tn: bad alignment or pitches");
    TnArgs3 a;
    a.X = reinterpret_cast<const _Float16*>(X), a.D = reinterpret_cast<const _Float16*>(D), a.part = workspace, a.ldx = ldx, a.ldd = ldd;
    a.N = (int)n_rows, a.K = (int)k, a.P = (int)p, a.KP = (int)kp, a.PP = (int)pp, a.x2_off = (int)x2_off, a.d2_off = (int)d2_off;
    a.tiles_k = (int)((kp + TT - 1) / TT), a.tiles_p = (int)((pp + TT - 1) / TT);
    int splits = tn_splits(n_rows, a.tiles_k * a.tiles_p);
    const int rps = (int)(((n_rows + splits - 1) / splits + TBK - 1) / TBK * TBK);
    splits = (int)((n_rows + rps - 1) / rps);             // no empty split (never more than tn_splits: the workspace holds them)
    a.splits = splits, a.rows_per_split = rps;
    a.mode = mode;
    BOT_REQUIRE((int64_t)rps * std::max(ldx, ldd) * 2 < (1ll << 31), -1, "gemm_halves3_tn: a split of %d rows exceeds the 2 GiB a buffer descriptor spans", rps);
    set_kernel("bot::gemm_halves3_tn_kernel");
    if (mode) hipLaunchKernelGGL(gemm_halves3_tn_ablate_kernel, dim3(((splits + 7) / 8) * 8 * a.tiles_k * a.tiles_p), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_halves3_tn_kernel, dim3(((splits + 7) / 8) * 8 * a.tiles_k * a.tiles_p), dim3(512), 0, (hipStream_t)stream, a);
    const int64_t n4 = k * ((p + 3) / 4);
    hipLaunchKernelGGL(tn_reduce_h3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, splits, (int)k, (int)p,
                       (int)kp, (int)pp, scale_x, scale_d, out, ldo, scale_d2, (int)p_split);
    return hip_status("gemm_halves3_tn");
}
