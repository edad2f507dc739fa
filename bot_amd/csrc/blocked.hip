// L2-blocked SpMM for DENSE graphs (mean degree in the hundreds: S-reddit 488, S-proteins 586) — gfx950.
//
// Why a second SpMM: on those graphs every source row is gathered ~500 times per pass.  The row-per-group kernel
// (spmm.hip) serves the re-reads from the 256 MiB Infinity Cache at its random-row ceiling (7.5 TB/s delivered, 15.5 ms
// for S-reddit F=256); the only faster level with useful capacity is the 4 MiB L2 of each XCD (16.8-18.8 TB/s for L2-
// resident row gathers, MI355X_MICROARCH.md "Indexed rows").  So the sweep is re-ordered to make the L2 working set small:
//
//   * sources are cut into column blocks of CB rows (CB * row bytes ~ 1-2 MB);
//   * destination rows are grouped (by degree) into tiles of T rows; ONE 256-thread workgroup owns a tile for the whole
//     pass and keeps its T output rows in LDS (T * F * 4 bytes <= 64 KB) — the neighbour tile is what is staged on chip;
//   * every workgroup walks the column blocks in the SAME order, so at any moment all workgroups of an XCD gather from the
//     same ~CB source rows, which therefore live in that XCD's L2; the host launches the tiles in rounds of one
//     resident wave of workgroups (kernel boundaries keep the rounds aligned — no in-kernel grid barrier);
//   * the workgroup is 16 wavefronts (one per CU, T = 128 rows of 1 KB in LDS); wave w owns the tile rows w, w+16, ...;
//     its edges are pre-sorted into ONE contiguous stream ordered by (column block, row, edge id), which it walks linearly,
//     8 gathers in flight: it sums a row's neighbours of the current block in registers and folds them into the LDS row
//     when the row changes (one LDS read-modify-write per (row, block) visit, none per edge); no two waves touch the same
//     LDS row: no atomics, fixed summation order (block-major, then edge id) => bitwise reproducible;
//   * the 16 waves of a workgroup cross the column blocks in lockstep (one __syncthreads per block), which makes the
//     workgroup ONE sweeper with ~T*deg/nblk edges per block: the spread between the 32 sweepers of an XCD (a random walk
//     in the per-block work) then stays a fraction of the L2.  Sweepers of different workgroups are not synchronised.
//
// Rows far above the mean degree (hubs) and everything else stay on the row-per-group kernel.
// HBM roofline unchanged: 4*[2*n*F + nnz + ...] algorithmic bytes; what changes is where the re-reads are served.
#include "common.h"

namespace bot {

struct BlockedArgs {
    const int32_t* tile_rows;  // [n_tiles * T]  destination row of each tile slot, -1 = padding
    const int32_t* ptr;        // [n_tiles * 16 + 1]  edge-stream offsets, (tile, wave)-major
    const int32_t* b_src;      // [nnz_b] source row of each blocked edge
    const uint8_t* b_lrow;     // [nnz_b] slot of the destination row inside its tile
    const int32_t* b_pos;      // [nnz_b] position in the unblocked edge order (row of w), used when weighted
    int32_t tile0, n_tiles, nblk;
    const float* x;
    int64_t ldx;
    const float* w;            // [nnz, H] or NULL
    int32_t H, D, F;           // F = H * D floats per row
    int32_t cb_shift;          // log2(source rows per column block)
    float* out;
    int64_t ldo;
    bool wstage;               // weighted, H <= 8 and 32 KB of LDS to spare: a batch's weights are staged in LDS
    const float* addend;       // optional epilogue: out[r,:] += addend[r,:] (the layer's residual branch), row stride lda
    int64_t lda;
};

constexpr int kBWaves = 16;             // wavefronts per workgroup (1024 threads): one workgroup per CU
constexpr int kBThreads = kBWaves * 64;

// Sum racc over the EPI lane groups of the wave (EPI > 1), then add it to the LDS row (lanes of group 0) and clear it.
template <int VEC, int NCHUNK, int EPI>
__device__ __forceinline__ void lds_fold(float* q_row, int lane, float (&racc)[NCHUNK][VEC]) {
    constexpr int G = 64 / EPI;
    const int li = lane & (G - 1);
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
            float v = racc[c][t];
            if constexpr (EPI >= 2) v += __shfl_xor(v, 32);
            if constexpr (EPI >= 4) v += __shfl_xor(v, 16);
            racc[c][t] = v;
        }
        if (EPI == 1 || lane < G) {
            float* q = q_row + (c * G + li) * VEC;
            float o[VEC];
            vload<VEC>(o, q);
#pragma unroll
            for (int t = 0; t < VEC; ++t) o[t] += racc[c][t];
            vstore<VEC>(q, o);
        }
#pragma unroll
        for (int t = 0; t < VEC; ++t) racc[c][t] = 0.f;
    }
}

// EPI = edges per gather instruction: rows of <= 32 (16) vector lanes are gathered 2 (4) at a time, one per 32- (16-) lane
// group of the wave; the host pads every (row, block) visit of a stream to a multiple of EPI slots (source -1 = no edge), so
// the EPI slots of one instruction always belong to the same destination row and their partial sums share `racc`.
template <int VEC, int NCHUNK, bool WEIGHTED, int T, int EPI>
__global__ __launch_bounds__(kBThreads) void spmm_blocked_kernel(BlockedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float acc_lds[];  // [T][Fp]
    static_assert(EPI == 1 || NCHUNK == 1, "lane groups only for rows that fit one chunk");
    constexpr int U = 8;
    constexpr int G = 64 / EPI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gi = lane / G, li = lane & (G - 1);
    const int tile = a.tile0 + blockIdx.x;  // grid never exceeds the tile count: every workgroup runs every barrier below
    constexpr int Fp = NCHUNK * G * VEC;
    int off[NCHUNK], hd[NCHUNK];
    bool act[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * G + li) * VEC;
        act[c] = e < a.F;
        off[c] = act[c] ? e : 0;
        hd[c] = act[c] ? e / a.D : 0;  // head of this lane's elements (D % VEC == 0: a vector never straddles heads)
    }
    // wave w owns the tile rows w, w+16, ...: it zeroes them, accumulates into them and stores them; no other wave touches them
    for (int r = wave + kBWaves * gi; r < T; r += kBWaves * EPI)
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c) {
            float z[VEC];
#pragma unroll
            for (int t = 0; t < VEC; ++t) z[t] = 0.f;
            vstore<VEC>(acc_lds + r * Fp + (c * G + li) * VEC, z);
        }
    // this wave's slot stream: the edges of its rows, sorted by (column block, row, edge id)
    int k0 = __builtin_amdgcn_readfirstlane(a.ptr[(int64_t)tile * kBWaves + wave]);
    const int end = __builtin_amdgcn_readfirstlane(a.ptr[(int64_t)tile * kBWaves + wave + 1]);
    int cur = -1;
    float racc[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int t = 0; t < VEC; ++t) racc[c][t] = 0.f;
    // current batch of up to 64 slots: (idx, lr, pos) one per lane, `i` consumed so far
    int idx = 0, lr = 0, pos = 0, i = 0, cnt = 0;
    // Weights: the H floats of a slot's edge are fetched ONCE per batch by the slot's lane and parked in LDS
    // ([wave][slot][8]); each gather then reads the weight of its lane's head from there.  (Fetching w[pos*H + head] from
    // memory per (edge, chunk) was 16 more loads per 8 edges: S-proteins 12.0 ms weighted vs 9.1 ms unweighted.)
    float* wl = acc_lds + (size_t)T * Fp + wave * (64 * 8);
    auto load_batch = [&]() {
        cnt = min(64, end - k0);
        i = 0;
        if (lane < cnt) {
            idx = a.b_src[k0 + lane];
            lr = a.b_lrow[k0 + lane];
            if constexpr (WEIGHTED) {
                pos = a.b_pos[k0 + lane];
                if (a.wstage) {
                    const float* pw = a.w + (int64_t)max(pos, 0) * a.H;
#pragma unroll
                    for (int h = 0; h < 8; ++h)
                        if (h < a.H) wl[lane * 8 + h] = pw[h];
                }
            }
        }
        if constexpr (WEIGHTED) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // lanes read each other's slots below
    };
    if (k0 < end) load_batch();
    // All 16 waves of the workgroup take the column blocks in lockstep: a wave executes barrier #b+1 when the next edge of its
    // stream lies beyond block b, so every wave runs exactly `nblk` barriers and the workgroup is ONE sweeper with T rows'
    // worth of edges per block (the spread between the sweepers of an XCD then stays a fraction of the L2).  Gather groups
    // are always full (U instructions) and may reach a few edges into the next block: those rows are about to be fetched by
    // every sweeper anyway, and no group pays the memory latency for one or two trailing edges of a block.
    int synced = 0;
    while (cnt > 0) {
        if (i == cnt) {  // batch used up
            k0 += 64;
            if (k0 >= end) break;
            load_batch();
        }
        const int fb = __builtin_amdgcn_readlane(idx, i) >> a.cb_shift;  // first slot of an instruction is a real edge
        for (; synced < fb; ++synced) __syncthreads();
        const int g = min(U, (cnt - i) / EPI);
        float v[U][NCHUNK][VEC], ww[U][NCHUNK];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(i + u * EPI, i + (g - 1) * EPI);
            int s;
            if constexpr (EPI == 1) s = __builtin_amdgcn_readlane(idx, j);
            else s = __builtin_amdgcn_ds_bpermute((j + gi) << 2, idx);
            const bool real = EPI == 1 || s >= 0;
            const float* px = a.x + (int64_t)(real ? s : 0) * a.ldx;
#pragma unroll
            for (int c = 0; c < NCHUNK; ++c) {
                vload<VEC>(v[u][c], px + off[c]);
                if constexpr (EPI > 1) {
#pragma unroll
                    for (int t = 0; t < VEC; ++t) v[u][c][t] = real ? v[u][c][t] : 0.f;
                }
            }
            if constexpr (WEIGHTED) {
                int ps;
                if (a.wstage) {
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c) ww[u][c] = wl[(j + gi) * 8 + hd[c]];
                } else {
                    if constexpr (EPI == 1) ps = __builtin_amdgcn_readlane(pos, j);
                    else ps = max(__builtin_amdgcn_ds_bpermute((j + gi) << 2, pos), 0);
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c) ww[u][c] = a.w[(int64_t)ps * a.H + hd[c]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < g) {  // wave-uniform
                const int r = __builtin_amdgcn_readlane(lr, i + u * EPI);
                if (r != cur) {
                    if (cur >= 0) lds_fold<VEC, NCHUNK, EPI>(acc_lds + cur * Fp, lane, racc);
                    cur = r;
                }
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                    for (int t = 0; t < VEC; ++t) {
                        if constexpr (WEIGHTED) racc[c][t] = fmaf(ww[u][c], v[u][c][t], racc[c][t]);
                        else racc[c][t] += v[u][c][t];
                    }
            }
        }
        i += g * EPI;
    }
    for (; synced < a.nblk; ++synced) __syncthreads();
    if (cur >= 0) lds_fold<VEC, NCHUNK, EPI>(acc_lds + cur * Fp, lane, racc);
    for (int r = wave + kBWaves * gi; r < T; r += kBWaves * EPI) {  // rows of this wave -> global, EPI rows per store
        const int row = a.tile_rows[(int64_t)tile * T + r];
        if (row < 0) continue;
#pragma unroll
        for (int c = 0; c < NCHUNK; ++c)
            if (act[c]) {
                float o[VEC];
                vload<VEC>(o, acc_lds + r * Fp + (c * G + li) * VEC);
                if (a.addend) {
                    float ad[VEC];
                    vload<VEC>(ad, a.addend + (int64_t)row * a.lda + off[c]);
#pragma unroll
                    for (int t = 0; t < VEC; ++t) o[t] += ad[t];
                }
                vstore<VEC>(a.out + (int64_t)row * a.ldo + off[c], o);
            }
    }
}

template <int VEC, int NCHUNK, int T, int EPI>
static int launch_blocked(const BlockedArgs& a0, int round_tiles, hipStream_t st) {
    size_t lds = (size_t)T * NCHUNK * (64 / EPI) * VEC * sizeof(float);
    BlockedArgs a = a0;
    a.wstage = a.w != nullptr && a.H <= 8 && lds + kBWaves * 64 * 8 * sizeof(float) <= 160 * 1024;
    if (a.wstage) lds += kBWaves * 64 * 8 * sizeof(float);
    auto k1 = spmm_blocked_kernel<VEC, NCHUNK, true, T, EPI>;
    auto k0 = spmm_blocked_kernel<VEC, NCHUNK, false, T, EPI>;
    if (lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    set_kernel("bot::spmm_blocked_kernel<%d,%d,%s,%d,%d>", VEC, NCHUNK, a.w ? "true" : "false", T, EPI);
    for (int t0 = 0; t0 < a.n_tiles; t0 += round_tiles) {  // one resident wave of workgroups per launch keeps the sweeps aligned
        a.tile0 = t0;
        const int n = a.n_tiles - t0 < round_tiles ? a.n_tiles - t0 : round_tiles;
        if (a.w) hipLaunchKernelGGL(k1, dim3(n), dim3(kBThreads), lds, st, a);
        else hipLaunchKernelGGL(k0, dim3(n), dim3(kBThreads), lds, st, a);
    }
    return hip_status("spmm_blocked launch");
}

}  // namespace bot

extern "C" {

int bot_spmm_blocked_f32(const int32_t* tile_rows, const int32_t* ptr, const int32_t* b_src, const uint8_t* b_lrow,
                         const int32_t* b_pos, int32_t n_tiles, int32_t nblk, int32_t block_rows, int32_t T, int32_t epi,
                         int32_t round_tiles, const float* x, int64_t ldx, const float* w, int32_t H, int32_t D, float* out,
                         int64_t ldo, const float* addend, int64_t lda, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_tiles >= 0 && nblk >= 1 && round_tiles >= 1, BOT_E_RANGE, "spmm_blocked: n_tiles=%d nblk=%d round=%d", n_tiles, nblk, round_tiles);
    BOT_REQUIRE(T == 256 || T == 128 || T == 64 || T == 32, BOT_E_RANGE, "spmm_blocked: tile height %d (32, 64, 128 or 256)", T);
    BOT_REQUIRE(epi == 1 || epi == 2 || epi == 4, BOT_E_RANGE, "spmm_blocked: %d edges per instruction (1, 2 or 4)", epi);
    int shift = 0;
    while ((1 << shift) < block_rows) ++shift;
    BOT_REQUIRE(block_rows >= 1 && (1 << shift) == block_rows, BOT_E_RANGE, "spmm_blocked: block_rows=%d must be a power of two", block_rows);
    BOT_REQUIRE(H >= 1 && D >= 1 && (int64_t)H * D <= 1024, BOT_E_RANGE, "spmm_blocked: H*D=%lld exceeds 1024", (long long)H * D);
    if (n_tiles == 0) return 0;
    BOT_REQUIRE(tile_rows && ptr && b_src && b_lrow && x && out && (w == nullptr || b_pos), BOT_E_NULL, "spmm_blocked: NULL pointer");
    const int F = H * D;
    BOT_REQUIRE(ldx >= F && ldo >= F && (addend == nullptr || lda >= F), BOT_E_RANGE, "spmm_blocked: row stride smaller than H*D");
    const int vec = addend ? pick_vec(D, {ldx, ldo, lda}, {x, out, addend}) : pick_vec(D, {ldx, ldo}, {x, out});
    const int L = (F + vec - 1) / vec;
    const int G = 64 / epi;
    const int nchunk = (L + G - 1) / G;
    BOT_REQUIRE(epi == 1 || nchunk == 1, BOT_E_RANGE, "spmm_blocked: rows of %d x %d-float vectors do not fit the %d-lane groups of epi=%d (plan built for another row layout?)", L, vec, G, epi);
    BOT_REQUIRE(epi > 1 || T <= 128, BOT_E_RANGE, "spmm_blocked: T=256 only with lane groups (epi > 1)");
    BOT_REQUIRE(nchunk <= 4 && (size_t)T * nchunk * G * vec * 4 <= 160 * 1024, BOT_E_RANGE, "spmm_blocked: tile does not fit LDS");
    BlockedArgs a{tile_rows, ptr, b_src, b_lrow, b_pos, 0, n_tiles, nblk, x, ldx, w, H, D, F, shift, out, ldo, false, addend, lda};
    hipStream_t st = (hipStream_t)stream;
#define BOT_BLK(V, NC)                                                        \
    do {                                                                      \
        if (T == 128) return launch_blocked<V, NC, 128, 1>(a, round_tiles, st); \
        if (T == 64) return launch_blocked<V, NC, 64, 1>(a, round_tiles, st);  \
        return launch_blocked<V, NC, 32, 1>(a, round_tiles, st);              \
    } while (0)
#define BOT_BLK_G(V, EP)                                                      \
    do {                                                                      \
        if (T == 256) return launch_blocked<V, 1, 256, EP>(a, round_tiles, st); \
        if (T == 128) return launch_blocked<V, 1, 128, EP>(a, round_tiles, st); \
        if (T == 64) return launch_blocked<V, 1, 64, EP>(a, round_tiles, st);  \
        return launch_blocked<V, 1, 32, EP>(a, round_tiles, st);              \
    } while (0)
#define BOT_BLK_V(V)                     \
    do {                                 \
        if (epi == 4) BOT_BLK_G(V, 4);   \
        if (epi == 2) BOT_BLK_G(V, 2);   \
        if (nchunk == 1) BOT_BLK(V, 1);  \
        if (nchunk == 2) BOT_BLK(V, 2);  \
        if (nchunk == 3) BOT_BLK(V, 3);  \
        BOT_BLK(V, 4);                   \
    } while (0)
    if (vec == 4) BOT_BLK_V(4);
    if (vec == 2) BOT_BLK_V(2);
    BOT_BLK_V(1);
#undef BOT_BLK_V
#undef BOT_BLK_G
#undef BOT_BLK
}

}  // extern "C"
