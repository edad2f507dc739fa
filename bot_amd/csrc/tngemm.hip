// bot_tn_gemm_f32:  out[kx, ky] = sum_r X[r, kx] Y[r, ky]  — weight gradients whose reduction runs over the N node rows while the
// result is small (config-2 layer 0: d W_r = h^T d out2 is [168, 768], d W_i = d x_i^T z_i is [250, 168] per head; N = 169 343).
// Library GEMMs reach ~60 TFLOP/s there (a handful of output tiles, the parallelism has to come from splitting N).
//
// The fp32 MFMA v_mfma_f32_32x32x2_f32 takes this product without any transposition: its A operand is A[i = lane & 31][k = lane >> 5]
// and its B operand B[k = lane >> 5][j = lane & 31]; with k = the row index, lane (i, k) reads X[r0 + k][kx0 + i] and
// Y[r0 + k][ky0 + i] — for each k the 32 lanes read 32 CONSECUTIVE columns of one row: coalesced 128-byte segments straight from
// memory, no transposition, no conversion, exact fp32 products.  The per-chunk partial results go to a workspace and a second kernel
// adds them in chunk order (deterministic, no atomics).
#include "common.h"

#include <stdlib.h>

namespace bot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TnArgs {
    const float* X;
    int64_t ldx, sx;
    const float* Y;
    int64_t ldy, sy;
    float* part;           // [batch][chunks][kxp][kyp]
    int64_t n;
    int32_t kx, ky, kxp, kyp;   // kxp / kyp: kx / ky rounded up to the workgroup block (kTnBX / kTnBY)
    int32_t chunks, rows_per_chunk, ngx, ngy, batch;
};

constexpr int kTnTI = 2, kTnTJ = 3;            // 32x32 tiles per wave: TI along X's columns, TJ along Y's
constexpr int kTnWI = 4, kTnWJ = 2;            // waves per workgroup along the two axes
constexpr int kTnBX = 32 * kTnTI * kTnWI;      // 256 columns of X per workgroup
constexpr int kTnBY = 32 * kTnTJ * kTnWJ;      // 192 columns of Y per workgroup
constexpr int kTnStage = 16;                   // rows per LDS stage (8 MFMA k-steps)
constexpr int kTnThreads = 64 * kTnWI * kTnWJ;

// One workgroup = 8 waves = a 256 x 192 block of the output for one chunk of rows.  The rows of X (256 columns) and Y (192
// columns) are staged through LDS 16 at a time (double buffered: the next stage's global loads are issued before this stage's
// MFMAs), so each operand element is read from memory once per workgroup, as whole coalesced row segments.
__global__ __launch_bounds__(kTnThreads) void tn_gemm_kernel(TnArgs a) {
    __shared__ float xs[2][kTnStage][kTnBX + 4];
    __shared__ float ys[2][kTnStage][kTnBY + 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lk = lane >> 5;
    const int wi = wave / kTnWJ, wj = wave % kTnWJ;
    int64_t w = blockIdx.x;                         // -> (batch, row chunk, X column group, Y column group)
    const int gy = (int)(w % a.ngy);
    w /= a.ngy;
    const int gxc = (int)(w % a.ngx);
    w /= a.ngx;
    const int chunk = (int)(w % a.chunks);
    const int64_t z = w / a.chunks;
    const float* X = a.X + z * a.sx + (int64_t)gxc * kTnBX;
    const float* Y = a.Y + z * a.sy + (int64_t)gy * kTnBY;
    const int xcols = a.kx - gxc * kTnBX, ycols = a.ky - gy * kTnBY;      // valid columns of this block (may exceed the block width)
    const int64_t r_begin = (int64_t)chunk * a.rows_per_chunk;
    const int64_t r_end = r_begin + a.rows_per_chunk < a.n ? r_begin + a.rows_per_chunk : a.n;
    // Staging map: thread t moves column t % B of rows t / B + (512 / B) q of a stage — X: 256 columns, rows {0,1} + 2 q, q < 8; Y: 192
    // columns on threads 0..383, rows {0,1} + 2 q, q < 8 — so a thread's loads differ by a wave-uniform row stride (one 32-bit offset each).
    constexpr int NX = 8, NY = 8;
    const int xc = threadIdx.x & 255, xr0 = threadIdx.x >> 8;
    const int yc = threadIdx.x % kTnBY, yr0 = threadIdx.x / kTnBY;           // yr0 == 2: idle for Y (threads 384..511)
    const bool xok = xc < xcols, yok = yr0 < 2 && yc < ycols;
    // TWO stages of global loads in flight (round 3): with one, a workgroup's 1.3 us of MFMAs per stage did not cover a load's latency —
    // the kernel moved 850 MB at 1.4 TB/s.  Register sets A / B alternate (static indexing: the loop body is written out twice).
    float xa_[NX], ya_[NY], xb_[NX], yb_[NY];
    auto gload = [&](float (&xr)[NX], float (&yr)[NY], int64_t r0) {
        const float* xp = X + (r0 + xr0) * a.ldx + xc;
        const float* yp = Y + (r0 + yr0) * a.ldy + yc;
        const int nrow = (int)(r_end - r0);                                   // rows of this stage that exist (>= 1)
#pragma unroll
        for (int q = 0; q < NX; ++q) xr[q] = (xok && xr0 + 2 * q < nrow) ? xp[(int64_t)(2 * q) * a.ldx] : 0.f;
#pragma unroll
        for (int q = 0; q < NY; ++q) yr[q] = (yok && yr0 + 2 * q < nrow) ? yp[(int64_t)(2 * q) * a.ldy] : 0.f;
    };
    auto lstore = [&](const float (&xr)[NX], const float (&yr)[NY], int buf) {
#pragma unroll
        for (int q = 0; q < NX; ++q) xs[buf][xr0 + 2 * q][xc] = xr[q];
        if (yr0 < 2) {
#pragma unroll
            for (int q = 0; q < NY; ++q) ys[buf][yr0 + 2 * q][yc] = yr[q];
        }
    };
    f32x16 acc[kTnTI][kTnTJ];
#pragma unroll
    for (int i = 0; i < kTnTI; ++i)
#pragma unroll
        for (int j = 0; j < kTnTJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll 2
        for (int ks = 0; ks < kTnStage / 2; ++ks) {
            float xa[kTnTI], ya[kTnTJ];
#pragma unroll
            for (int t = 0; t < kTnTI; ++t) xa[t] = xs[buf][2 * ks + lk][(wi * kTnTI + t) * 32 + li];
#pragma unroll
            for (int t = 0; t < kTnTJ; ++t) ya[t] = ys[buf][2 * ks + lk][(wj * kTnTJ + t) * 32 + li];
#pragma unroll
            for (int i = 0; i < kTnTI; ++i)
#pragma unroll
                for (int j = 0; j < kTnTJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i], ya[j], acc[i][j], 0, 0, 0);
        }
    };
    // stage s covers rows r_begin + s * kTnStage ...; LDS buffer s & 1 holds stage s while it is multiplied; the register set that holds
    // stage s + 1 was loaded one stage earlier and is stored to the other buffer after the products; stage s + 2 is requested first.
    const int64_t nst = (r_end - r_begin + kTnStage - 1) / kTnStage;
    if (nst > 0) {
        gload(xa_, ya_, r_begin);
        lstore(xa_, ya_, 0);
        if (nst > 1) gload(xb_, yb_, r_begin + kTnStage);            // stage 1 -> set B
    }
    __syncthreads();
    for (int64_t s0 = 0; s0 < nst; s0 += 2) {
        // even stage s0: LDS buffer 0; set B holds stage s0 + 1; request stage s0 + 2 into set A
        if (s0 + 2 < nst) gload(xa_, ya_, r_begin + (s0 + 2) * kTnStage);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 1 < nst) lstore(xb_, yb_, 1);
        __syncthreads();
        if (s0 + 1 >= nst) break;
        // odd stage s0 + 1: LDS buffer 1; set A holds stage s0 + 2; request stage s0 + 3 into set B
        if (s0 + 3 < nst) gload(xb_, yb_, r_begin + (s0 + 3) * kTnStage);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        if (s0 + 2 < nst) lstore(xa_, ya_, 0);
        __syncthreads();
    }
    // C/D map: column (j) = lane & 31, row (i) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* P = a.part + ((z * a.chunks + chunk) * a.kxp) * (int64_t)a.kyp;
#pragma unroll
    for (int i = 0; i < kTnTI; ++i)
#pragma unroll
        for (int j = 0; j < kTnTJ; ++j) {
            const int col = gy * kTnBY + (wj * kTnTJ + j) * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = gxc * kTnBX + (wi * kTnTI + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                P[(int64_t)row * a.kyp + col] = acc[i][j][e];
            }
        }
}

// out[z][kx][ky] = sum over chunks of part[z][chunk][kx][ky], in chunk order
__global__ __launch_bounds__(kBlock) void tn_reduce_kernel(const float* part, int chunks, int kxp, int kyp, int kx, int ky, float* out, int64_t ldo,
                                                          int64_t so, bool transposed) {
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int z = blockIdx.y;
    if (idx >= (int64_t)kx * ky) return;
    const int r = (int)(idx / ky), c = (int)(idx % ky);
    const float* p = part + ((int64_t)z * chunks * kxp + r) * kyp + c;
    const int64_t step = (int64_t)kxp * kyp;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= chunks; k += 8) {      // eight loads in flight, added in chunk order (a dependent load per term ran at 0.5 TB/s)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(k + j) * step];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < chunks; ++k) s += p[k * step];
    out[z * so + (transposed ? (int64_t)c * ldo + r : (int64_t)r * ldo + c)] = s;
}

// bot_tn_narrow_f32: the same product when X has only a handful of columns (kx <= 32: the attention columns d el / d er of the merged
// gradient against the layer input, ky <= 256): a memory-bound reduction.  One thread per column of Y, X's row is a wave-uniform
// (scalar) load, kx accumulators per thread; a workgroup takes a contiguous row range, the per-workgroup partials are added in
// workgroup order by the second kernel (deterministic).  Plain fp32 FMAs: exact products, fp32 accumulation.
constexpr int kNarrowMaxX = 32, kNarrowRows = 256;

// KX4: kx rounded up to a multiple of 4 (compile time: the accumulators are registers).  X's rows of the workgroup's range are staged in
// LDS once ([rows][KX4], zero padded; every thread then reads the same float4: a broadcast), Y is read eight rows at a time.
template <int KX4>
__global__ __launch_bounds__(256) void tn_narrow_kernel(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t n, int kx, int ky,
                                                        float* part) {
    __shared__ float xs[kNarrowRows * KX4];
    const int f = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * kNarrowRows;
    const int rows = (int)(r0 + kNarrowRows < n ? kNarrowRows : n - r0);
    for (int i = f; i < kNarrowRows * KX4; i += 256) {
        const int r = i / KX4, j = i - r * KX4;
        xs[i] = (r < rows && j < kx) ? X[(r0 + r) * ldx + j] : 0.f;
    }
    __syncthreads();
    if (f >= ky) return;
    float acc[KX4];
#pragma unroll
    for (int j = 0; j < KX4; ++j) acc[j] = 0.f;
    const float* yp = Y + r0 * ldy + f;
    constexpr int U = 8;
    int r = 0;
    for (; r + U <= rows; r += U) {
        float y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = yp[(int64_t)(r + u) * ldy];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < KX4; j += 4) {
                const float4 x4 = *reinterpret_cast<const float4*>(&xs[(r + u) * KX4 + j]);
                acc[j] = fmaf(x4.x, y[u], acc[j]), acc[j + 1] = fmaf(x4.y, y[u], acc[j + 1]);
                acc[j + 2] = fmaf(x4.z, y[u], acc[j + 2]), acc[j + 3] = fmaf(x4.w, y[u], acc[j + 3]);
            }
    }
    for (; r < rows; ++r) {
        const float y = yp[(int64_t)r * ldy];
#pragma unroll
        for (int j = 0; j < KX4; ++j) acc[j] = fmaf(xs[r * KX4 + j], y, acc[j]);
    }
    float* o = part + ((int64_t)blockIdx.x * kx) * ky + f;
#pragma unroll
    for (int j = 0; j < KX4; ++j)
        if (j < kx) o[(int64_t)j * ky] = acc[j];
}

// out[j, f] (transposed: out[f, j]) = sum_b part[b][j][f], in workgroup order: 32 outputs per workgroup, eight threads per output each
// adding a contiguous eighth of the partials, the eight sums added in order by one of them
__global__ __launch_bounds__(256) void tn_narrow_reduce_kernel(const float* part, int blocks, int kx, int ky, float* out, int64_t ldo, int transpose_out) {
    __shared__ float sums[8][32];
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    const int per = (blocks + 7) / 8, b0 = sl * per, b1 = b0 + per < blocks ? b0 + per : blocks;
    float s = 0.f;
    if (i < kx * ky) {
        const float* src = part + i;
        const int64_t stride = (int64_t)kx * ky;
        int b = b0;
        for (; b + 4 <= b1; b += 4) {           // four loads in flight, added in order
            const float v0 = src[b * stride], v1 = src[(b + 1) * stride], v2 = src[(b + 2) * stride], v3 = src[(b + 3) * stride];
            s += v0, s += v1, s += v2, s += v3;
        }
        for (; b < b1; ++b) s += src[b * stride];
    }
    sums[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < kx * ky) {
        float t = sums[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sums[k][o];
        const int j = i / ky, f = i - j * ky;
        if (transpose_out) out[(int64_t)f * ldo + j] = t;
        else out[(int64_t)j * ldo + f] = t;
    }
}

}  // namespace bot

extern "C" {

int64_t bot_tn_narrow_workspace_floats(int64_t n, int32_t kx, int32_t ky) {
    return ((n + bot::kNarrowRows - 1) / bot::kNarrowRows) * (int64_t)kx * ky;
}

int bot_tn_narrow_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t n, int32_t kx, int32_t ky, float* out, int64_t ldo,
                      int32_t transpose_out, float* workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && kx >= 1 && kx <= kNarrowMaxX && ky >= 1 && ky <= 256, BOT_E_RANGE, "tn_narrow: n=%lld kx=%d (1..%d) ky=%d (1..256)", (long long)n, kx,
                kNarrowMaxX, ky);
    BOT_REQUIRE(ldx >= kx && ldy >= ky && ldo >= (transpose_out ? kx : ky), BOT_E_RANGE, "tn_narrow: ldx=%lld ldy=%lld ldo=%lld", (long long)ldx, (long long)ldy,
                (long long)ldo);
    BOT_REQUIRE(X && Y && out && workspace, BOT_E_NULL, "tn_narrow: NULL pointer");
    const int blocks = (int)((n + kNarrowRows - 1) / kNarrowRows);
    hipStream_t st = (hipStream_t)stream;
    set_kernel("bot::tn_narrow_kernel<%d>", (kx + 3) / 4 * 4);
#define BOT_NARROW(K4) \
    case K4: hipLaunchKernelGGL(tn_narrow_kernel<K4>, dim3(blocks), dim3(256), 0, st, X, ldx, Y, ldy, n, (int)kx, (int)ky, workspace); break;
    switch ((kx + 3) / 4 * 4) {
        BOT_NARROW(4) BOT_NARROW(8) BOT_NARROW(12) BOT_NARROW(16) BOT_NARROW(20) BOT_NARROW(24) BOT_NARROW(28) BOT_NARROW(32)
    }
#undef BOT_NARROW
    hipLaunchKernelGGL(tn_narrow_reduce_kernel, dim3((kx * ky + 31) / 32), dim3(256), 0, st, (const float*)workspace, blocks, (int)kx, (int)ky, out, ldo,
                       (int)transpose_out);
    return hip_status("tn_narrow");
}

static void tn_shape(int64_t n, int32_t kx, int32_t ky, int32_t batch, int64_t* kxp, int64_t* kyp, int64_t* chunks) {
    using namespace bot;
    *kxp = (kx + kTnBX - 1) / kTnBX * kTnBX, *kyp = (ky + kTnBY - 1) / kTnBY * kTnBY;
    const int64_t groups = (*kxp / kTnBX) * (*kyp / kTnBY) * batch;
    static const int64_t target = [] {                                  // workgroups in total (BOT_TN_WGS: measurements)
        const char* e = getenv("BOT_TN_WGS");
        return e ? (int64_t)atoi(e) : (int64_t)768;
    }();
    int64_t c = (target + groups - 1) / groups;                         // ~3 workgroups per CU in total
    const int64_t max_chunks = (n + 255) / 256;                         // at least 256 rows per chunk
    *chunks = c < 1 ? 1 : (c > max_chunks ? (max_chunks < 1 ? 1 : max_chunks) : c);
}

int64_t bot_tn_gemm_workspace_floats(int64_t n, int32_t kx, int32_t ky, int32_t batch) {
    int64_t kxp, kyp, chunks;
    tn_shape(n, kx, ky, batch, &kxp, &kyp, &chunks);
    return (int64_t)batch * chunks * kxp * kyp;
}

int bot_tn_gemm_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t n, int32_t kx, int32_t ky, float* out, int64_t ldo,
                    int32_t transpose_out, int32_t batch, int64_t stride_x, int64_t stride_y, int64_t stride_out, float* workspace,
                    bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 1 && kx >= 1 && ky >= 1 && batch >= 1, BOT_E_RANGE, "tn_gemm: n=%lld kx=%d ky=%d batch=%d", (long long)n, kx, ky, batch);
    BOT_REQUIRE(ldx >= kx && ldy >= ky && ldo >= (transpose_out ? kx : ky), BOT_E_RANGE, "tn_gemm: ldx=%lld ldy=%lld ldo=%lld", (long long)ldx,
                (long long)ldy, (long long)ldo);
    BOT_REQUIRE(X && Y && out && workspace, BOT_E_NULL, "tn_gemm: NULL pointer");
    int64_t kxp, kyp, chunks;
    tn_shape(n, kx, ky, batch, &kxp, &kyp, &chunks);
    TnArgs a{};
    a.X = X, a.ldx = ldx, a.sx = stride_x, a.Y = Y, a.ldy = ldy, a.sy = stride_y, a.part = workspace, a.n = n, a.kx = kx, a.ky = ky;
    a.kxp = (int)kxp, a.kyp = (int)kyp, a.ngx = (int)(kxp / kTnBX), a.ngy = (int)(kyp / kTnBY), a.chunks = (int)chunks, a.batch = batch;
    a.rows_per_chunk = (int)(((n + chunks - 1) / chunks + kTnStage - 1) / kTnStage * kTnStage);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tn_gemm_kernel, dim3((unsigned)((int64_t)a.ngx * a.ngy * chunks * batch)), dim3(kTnThreads), 0, st, a);
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(((int64_t)kx * ky + kBlock - 1) / kBlock), (unsigned)batch), dim3(kBlock), 0, st, workspace,
                       (int)chunks, a.kxp, a.kyp, kx, ky, out, ldo, stride_out, transpose_out != 0);
    set_kernel("bot::tn_gemm_kernel kx=%d ky=%d chunks=%d batch=%d", kx, ky, (int)chunks, batch);
    return hip_status("tn_gemm launch");
}

}  // extern "C"
