// bot_tn_gemm_f32:  out[kx, ky] = sum_r X[r, kx] Y[r, ky]  — weight gradients whose reduction runs over the N node rows while the
// result is small (config-2 layer 0: d W_r = h^T d out2 is [168, 768], d W_i = d x_i^T z_i is [250, 168] per head; N = 169 343).
// Library GEMMs reach ~60 TFLOP/s there (a handful of output tiles, the parallelism has to come from splitting N).
//
// The fp32 MFMA v_mfma_f32_32x32x2_f32 takes this product without any transposition: its A operand is A[i = lane & 31][k = lane >> 5]
// and its B operand B[k = lane >> 5][j = lane & 31]; with k = the row index, lane (i, k) reads X[r0 + k][kx0 + i] and
// Y[r0 + k][ky0 + i] — for each k the 32 lanes read 32 CONSECUTIVE columns of one row: coalesced 128-byte segments straight from
// global memory, no LDS, no conversion, exact fp32 products.  One wave owns a TI x TJ block of 32x32 output tiles for one chunk of
// rows (8 rows = 4 MFMA k-steps per iteration, the next iteration's loads in flight); the per-chunk partial results go to a
// workspace and a second kernel adds them in chunk order (deterministic, no atomics).
#include "common.h"

namespace bot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TnArgs {
    const float* X;
    int64_t ldx, sx;
    const float* Y;
    int64_t ldy, sy;
    float* part;           // [batch][chunks][kxp][kyp]
    int64_t n;
    int32_t kx, ky, kxp, kyp;   // kxp / kyp: kx / ky rounded up to the tile block (32 TI / 32 TJ)
    int32_t chunks, rows_per_chunk, nbi, nbj, batch;
};

template <int TI, int TJ>
__global__ __launch_bounds__(256) void tn_gemm_kernel(TnArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lk = lane >> 5;
    // wave -> (batch, row chunk, tile block): the four waves of a workgroup take consecutive tile blocks of ONE row chunk
    int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const int nblk = a.nbi * a.nbj;
    const int blk = (int)(w % nblk);
    w /= nblk;
    const int chunk = (int)(w % a.chunks);
    const int64_t z = w / a.chunks;
    if (z >= a.batch) return;
    const int bi = blk / a.nbj, bj = blk % a.nbj;
    const float* X = a.X + z * a.sx + (int64_t)bi * 32 * TI + li;
    const float* Y = a.Y + z * a.sy + (int64_t)bj * 32 * TJ + li;
    bool xok[TI], yok[TJ];
#pragma unroll
    for (int t = 0; t < TI; ++t) xok[t] = bi * 32 * TI + t * 32 + li < a.kx;
#pragma unroll
    for (int t = 0; t < TJ; ++t) yok[t] = bj * 32 * TJ + t * 32 + li < a.ky;
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int64_t r_begin = (int64_t)chunk * a.rows_per_chunk;
    const int64_t r_end = r_begin + a.rows_per_chunk < a.n ? r_begin + a.rows_per_chunk : a.n;
    constexpr int U = 4;                         // MFMA k-steps (2 rows each) per iteration
    float xa[U][TI], ya[U][TJ], xn[U][TI], yn[U][TJ];
    auto load = [&](float (&xv)[U][TI], float (&yv)[U][TJ], int64_t r0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + 2 * u + lk;
            const bool rok = r < r_end;
            const float* xr = X + r * a.ldx;
            const float* yr = Y + r * a.ldy;
#pragma unroll
            for (int t = 0; t < TI; ++t) xv[u][t] = (rok && xok[t]) ? xr[t * 32] : 0.f;
#pragma unroll
            for (int t = 0; t < TJ; ++t) yv[u][t] = (rok && yok[t]) ? yr[t * 32] : 0.f;
        }
    };
    if (r_begin < r_end) load(xa, ya, r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * U) {
        if (r0 + 2 * U < r_end) load(xn, yn, r0 + 2 * U);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u][i], ya[u][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < TI; ++t) xa[u][t] = xn[u][t];
#pragma unroll
            for (int t = 0; t < TJ; ++t) ya[u][t] = yn[u][t];
        }
    }
    // C/D map: column (j) = lane & 31, row (i) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* P = a.part + ((z * a.chunks + chunk) * a.kxp) * (int64_t)a.kyp;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int col = bj * 32 * TJ + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = bi * 32 * TI + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                P[(int64_t)row * a.kyp + col] = acc[i][j][e];
            }
        }
}

// out[z][kx][ky] = sum over chunks of part[z][chunk][kx][ky], in chunk order
__global__ __launch_bounds__(kBlock) void tn_reduce_kernel(const float* part, int chunks, int kxp, int kyp, int kx, int ky, float* out, int64_t ldo,
                                                          int64_t so) {
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int z = blockIdx.y;
    if (idx >= (int64_t)kx * ky) return;
    const int r = (int)(idx / ky), c = (int)(idx % ky);
    const float* p = part + ((int64_t)z * chunks * kxp + r) * kyp + c;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += p[(int64_t)k * kxp * kyp];
    out[z * so + (int64_t)r * ldo + c] = s;
}

constexpr int kTnTI = 2, kTnTJ = 3;

}  // namespace bot

extern "C" {

int64_t bot_tn_gemm_workspace_floats(int64_t n, int32_t kx, int32_t ky, int32_t batch) {
    using namespace bot;
    const int64_t kxp = (kx + 32 * kTnTI - 1) / (32 * kTnTI) * (32 * kTnTI), kyp = (ky + 32 * kTnTJ - 1) / (32 * kTnTJ) * (32 * kTnTJ);
    const int64_t nblk = (kxp / (32 * kTnTI)) * (kyp / (32 * kTnTJ));
    int64_t chunks = (4096 + nblk * batch - 1) / (nblk * batch);          // ~4 waves per SIMD in total
    const int64_t max_chunks = (n + 255) / 256;                           // at least 256 rows per chunk
    chunks = chunks < 1 ? 1 : (chunks > max_chunks ? (max_chunks < 1 ? 1 : max_chunks) : chunks);
    return (int64_t)batch * chunks * kxp * kyp + 2;                       // the last two slots are not used by the kernels
}

int bot_tn_gemm_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t n, int32_t kx, int32_t ky, float* out, int64_t ldo,
                    int32_t batch, int64_t stride_x, int64_t stride_y, int64_t stride_out, float* workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && kx >= 1 && ky >= 1 && batch >= 1, BOT_E_RANGE, "tn_gemm: n=%lld kx=%d ky=%d batch=%d", (long long)n, kx, ky, batch);
    BOT_REQUIRE(ldx >= kx && ldy >= ky && ldo >= ky, BOT_E_RANGE, "tn_gemm: ldx=%lld ldy=%lld ldo=%lld", (long long)ldx, (long long)ldy, (long long)ldo);
    BOT_REQUIRE(X && Y && out && workspace, BOT_E_NULL, "tn_gemm: NULL pointer");
    TnArgs a{};
    a.X = X, a.ldx = ldx, a.sx = stride_x, a.Y = Y, a.ldy = ldy, a.sy = stride_y, a.part = workspace, a.n = n, a.kx = kx, a.ky = ky;
    a.kxp = (kx + 32 * kTnTI - 1) / (32 * kTnTI) * (32 * kTnTI), a.kyp = (ky + 32 * kTnTJ - 1) / (32 * kTnTJ) * (32 * kTnTJ);
    a.nbi = a.kxp / (32 * kTnTI), a.nbj = a.kyp / (32 * kTnTJ);
    const int64_t nblk = (int64_t)a.nbi * a.nbj;
    int64_t chunks = (4096 + nblk * batch - 1) / (nblk * batch);
    const int64_t max_chunks = (n + 255) / 256;
    chunks = chunks < 1 ? 1 : (chunks > max_chunks ? (max_chunks < 1 ? 1 : max_chunks) : chunks);
    a.chunks = (int)chunks, a.batch = batch;
    a.rows_per_chunk = (int)(((n + chunks - 1) / chunks + 7) / 8 * 8);
    hipStream_t st = (hipStream_t)stream;
    const int64_t waves = nblk * chunks;                                  // per batch entry
    hipLaunchKernelGGL((tn_gemm_kernel<kTnTI, kTnTJ>), dim3((unsigned)((waves * batch + 3) / 4)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(((int64_t)kx * ky + kBlock - 1) / kBlock), (unsigned)batch), dim3(kBlock), 0, st, workspace,
                       (int)chunks, a.kxp, a.kyp, kx, ky, out, ldo, stride_out);
    set_kernel("bot::tn_gemm_kernel<%d,%d> kx=%d ky=%d chunks=%d batch=%d", kTnTI, kTnTJ, kx, ky, (int)chunks, batch);
    return hip_status("tn_gemm launch");
}

}  // extern "C"
