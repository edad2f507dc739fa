// Merged projection weight of a GAT layer and its gradient (gfx950) — tiny tensors, few launches.
//
// bot_amd.nn.fused runs a layer's four linear maps on the layer input — fc (models.py:490-492), res_fc (:558-560) and the two
// attention scores folded through fc, el = h . (W_h^T attn_l[h]) (:517), er likewise (:521) — as ONE GEMM against
//     Wm [K, P] = [ W_fc^T | W_res^T | wl | wr | 0 ]        wl[k,h] = sum_d W_fc[h*D+d, k] * attn_l[h,d]
// (each of the two copied blocks `blk` >= H*D columns wide, zero padded: H*D rounded up to x4 keeps the residual block — which the
// backward sweep gathers from — on a 16-byte boundary)
// (the aggregate-before-project layer leaves W_fc^T out).  Built with stock tensor ops that is ~11 launches per layer forward
// and as many backward, every step; here it is one launch forward and two backward.
#include "common.h"

namespace bot {

struct MergeArgs {
    const float* W;     // [H*D, K]
    const float* Wres;  // [H*D, K] or NULL
    const float* al;    // [H*D]
    const float* ar;    // [H*D] or NULL
    int32_t H, D, K, P, with_fc;
    int32_t blk;        // column width of the W_fc^T / W_res^T blocks (>= H*D; H*D rounded up to x4 puts the residual block on a 16-byte boundary)
    float* out;         // forward: Wm [K, P]
    const float* dm;    // backward: d Wm [K, P]
    float* dW;          // [H*D, K]
    float* dWres;       // [H*D, K] or NULL
    float* dal;         // [H*D]
    float* dar;         // [H*D] or NULL
};

// Two workgroup ranges in one launch, both with coalesced reads AND writes:
//   [0, n_tiles)   32 x 32 tiles of the transposed copies W_fc^T / W_res^T through LDS;
//   [n_tiles, ..)  the score columns wl / wr (threads run along k, so the D rows of W they walk are read contiguously) and the pad.
__global__ __launch_bounds__(kBlock) void merge_fwd_kernel(MergeArgs a, int n_tiles, int tiles_k) {
    __shared__ float tile[32][33];
    const int HD = a.H * a.D;
    const int c0 = a.with_fc ? a.blk : 0, c1 = c0 + (a.Wres ? a.blk : 0), c2 = c1 + a.H, c3 = c2 + (a.ar ? a.H : 0);
    if ((int)blockIdx.x < n_tiles) {
        const int tk = blockIdx.x % tiles_k, tp = blockIdx.x / tiles_k;   // tp runs over the c1 copied columns
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8 threads
#pragma unroll
        for (int j = 0; j < 32; j += 8) {
            const int p = tp * 32 + ty + j, k = tk * 32 + tx;
            float v = 0.f;
            if (p < c1 && k < a.K) {
                const bool first = p < c0 || !a.with_fc;                  // which block, and the row inside it (rows >= H*D of a block: padding)
                const int q = p < c0 ? p : p - c0;
                if (q < HD) v = (a.with_fc && first) ? a.W[(int64_t)q * a.K + k] : a.Wres[(int64_t)q * a.K + k];
            }
            tile[ty + j][tx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; j += 8) {
            const int k = tk * 32 + ty + j, p = tp * 32 + tx;
            if (p < c1 && k < a.K) a.out[(int64_t)k * a.P + p] = tile[tx][ty + j];
        }
        return;
    }
    const int64_t gid = (int64_t)(blockIdx.x - n_tiles) * kBlock + threadIdx.x;
    const int nq = a.P - c1;
    if (gid >= (int64_t)nq * a.K) return;
    const int q = (int)(gid / a.K), k = (int)(gid - (int64_t)q * a.K), p = c1 + q;
    float v = 0.f;
    if (p < c3) {
        const bool left = p < c2;
        const int h = left ? p - c1 : p - c2;
        const float* at = left ? a.al : a.ar;
        // D independent loads per thread and only a few thousand threads: keep 8 in flight (4 partial sums)
        const float* wc = a.W + (int64_t)h * a.D * a.K + k;
        const float* ah = at + h * a.D;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int d = 0;
        for (; d + 8 <= a.D; d += 8) {
            float w[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) w[t] = wc[(int64_t)(d + t) * a.K];
            s0 = fmaf(w[0], ah[d], s0), s1 = fmaf(w[1], ah[d + 1], s1), s2 = fmaf(w[2], ah[d + 2], s2), s3 = fmaf(w[3], ah[d + 3], s3);
            s0 = fmaf(w[4], ah[d + 4], s0), s1 = fmaf(w[5], ah[d + 5], s1), s2 = fmaf(w[6], ah[d + 6], s2), s3 = fmaf(w[7], ah[d + 7], s3);
        }
        for (; d < a.D; ++d) s0 = fmaf(wc[(int64_t)d * a.K], ah[d], s0);
        v = (s0 + s1) + (s2 + s3);
    }
    a.out[(int64_t)k * a.P + p] = v;
}

// dW[p,k] = [with_fc] dm[k,p] + attn_l[p] * dm[k, c1+h] (+ attn_r[p] * dm[k, c2+h]);  dWres[p,k] = dm[k, c0+p]
__global__ __launch_bounds__(kBlock) void merge_bwd_w_kernel(MergeArgs a) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int HD = a.H * a.D;
    if (gid >= (int64_t)HD * a.K) return;
    const int p = (int)(gid / a.K), k = (int)(gid - (int64_t)p * a.K);
    const int h = p / a.D;
    const int c0 = a.with_fc ? a.blk : 0, c1 = c0 + (a.Wres ? a.blk : 0), c2 = c1 + a.H;
    const float* row = a.dm + (int64_t)k * a.P;
    float g = a.with_fc ? row[p] : 0.f;
    g = fmaf(a.al[p], row[c1 + h], g);
    if (a.ar) g = fmaf(a.ar[p], row[c2 + h], g);
    a.dW[gid] = g;
    if (a.dWres) a.dWres[gid] = row[c0 + p];
}

// dal[p] = sum_k W[p,k] * dm[k, c1+h];  dar[p] likewise with c2.  One 64-lane group per p.
__global__ __launch_bounds__(kBlock) void merge_bwd_a_kernel(MergeArgs a) {
    const int lane = threadIdx.x & 63;
    const int p = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
    const int HD = a.H * a.D;
    if (p >= HD) return;
    const int h = p / a.D;
    const int c0 = a.with_fc ? a.blk : 0, c1 = c0 + (a.Wres ? a.blk : 0), c2 = c1 + a.H;
    float sl = 0.f, sr = 0.f;
    for (int k = lane; k < a.K; k += 64) {
        const float w = a.W[(int64_t)p * a.K + k];
        sl = fmaf(w, a.dm[(int64_t)k * a.P + c1 + h], sl);
        if (a.ar) sr = fmaf(w, a.dm[(int64_t)k * a.P + c2 + h], sr);
    }
    sl = group_sum<64>(sl);
    sr = group_sum<64>(sr);
    if (lane == 0) {
        a.dal[p] = sl;
        if (a.dar) a.dar[p] = sr;
    }
}

static int merge_check(const char* who, const float* W, const float* al, int32_t H, int32_t D, int32_t K, int32_t P, const float* Wres,
                       const float* ar, int32_t with_fc, int32_t blk) {
    BOT_REQUIRE(H >= 1 && D >= 1 && K >= 1, BOT_E_RANGE, "%s: H=%d D=%d K=%d", who, H, D, K);
    BOT_REQUIRE(blk >= H * D, BOT_E_RANGE, "%s: block width %d smaller than H*D = %d", who, blk, H * D);
    BOT_REQUIRE(W && al, BOT_E_NULL, "%s: W / attn_l is NULL", who);
    const int64_t used = (int64_t)(with_fc ? blk : 0) + (Wres ? blk : 0) + H + (ar ? H : 0);
    BOT_REQUIRE(P >= used, BOT_E_RANGE, "%s: P=%d smaller than the %lld merged columns", who, P, (long long)used);
    return 0;
}

}  // namespace bot

extern "C" {

int bot_merge_weight_fwd_f32(const float* W, const float* Wres, const float* attn_l, const float* attn_r, int32_t H, int32_t D,
                             int32_t K, int32_t P, int32_t with_fc, int32_t block, float* out, bot_stream_t stream) {
    using namespace bot;
    if (int rc = merge_check("merge_weight_fwd", W, attn_l, H, D, K, P, Wres, attn_r, with_fc, block)) return rc;
    BOT_REQUIRE(out != nullptr, BOT_E_NULL, "merge_weight_fwd: out is NULL");
    MergeArgs a{W, Wres, attn_l, attn_r, H, D, K, P, with_fc, block, out, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int c1 = (with_fc ? block : 0) + (Wres ? block : 0);
    const int tiles_k = (K + 31) / 32, n_tiles = tiles_k * ((c1 + 31) / 32);
    const int64_t rest = (int64_t)(P - c1) * K;
    hipLaunchKernelGGL(merge_fwd_kernel, dim3((unsigned)(n_tiles + (rest + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, a,
                       n_tiles, tiles_k);
    return hip_status("merge_weight_fwd launch");
}

int bot_merge_weight_bwd_f32(const float* W, const float* attn_l, const float* attn_r, int32_t H, int32_t D, int32_t K, int32_t P,
                             int32_t with_fc, int32_t block, const float* d_merged, float* dW, float* dWres, float* d_attn_l, float* d_attn_r,
                             bot_stream_t stream) {
    using namespace bot;
    if (int rc = merge_check("merge_weight_bwd", W, attn_l, H, D, K, P, dWres, attn_r, with_fc, block)) return rc;
    BOT_REQUIRE(d_merged && dW && d_attn_l && ((attn_r == nullptr) == (d_attn_r == nullptr)), BOT_E_NULL, "merge_weight_bwd: NULL pointer");
    MergeArgs a{W, dWres, attn_l, attn_r, H, D, K, P, with_fc, block, nullptr, d_merged, dW, dWres, d_attn_l, d_attn_r};
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)H * D * K;
    hipLaunchKernelGGL(merge_bwd_w_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a);
    hipLaunchKernelGGL(merge_bwd_a_kernel, dim3((unsigned)(((int64_t)H * D * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a);
    return hip_status("merge_weight_bwd launch");
}

}  // extern "C"
