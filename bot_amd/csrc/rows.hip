// Row gather / scatter-add for halo packing in the 1-D vertex-partitioned mode (gfx950).
// One LANES-wide group per selected row, 16-byte lanes across the feature dimension, plain stores.
// scatter_add requires sorted-unique `rows`, so every destination row has exactly one writer:
// no atomics, bitwise reproducible.
#include "common.h"

namespace bot {

template <int VEC, bool ADD>
__global__ __launch_bounds__(kBlock) void rows_kernel(const float* src, int64_t lds_, const int32_t* rows, int64_t n_sel,
                                                     int32_t F, float* dst, int64_t ldd) {
    // GATHER (ADD=false): dst[i,:] = src[rows[i],:]      SCATTER-ADD (ADD=true): dst[rows[i],:] += src[i,:]
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / kWave;
    const int lane = threadIdx.x % kWave;
    if (wave >= n_sel) return;
    const int r = rows[wave];
    const float* s = ADD ? src + wave * lds_ : src + (int64_t)r * lds_;
    float* d = ADD ? dst + (int64_t)r * ldd : dst + wave * ldd;
    for (int e = lane * VEC; e < F; e += kWave * VEC) {
        float v[VEC];
        vload<VEC>(v, s + e);
        if constexpr (ADD) {
            float o[VEC];
            vload<VEC>(o, d + e);
#pragma unroll
            for (int t = 0; t < VEC; ++t) v[t] += o[t];
        }
        vstore<VEC>(d + e, v);
    }
}

template <bool ADD>
static int launch_rows(const char* who, const float* src, int64_t lds_, const int32_t* rows, int64_t n_sel, int32_t F,
                       float* dst, int64_t ldd, hipStream_t st) {
    BOT_REQUIRE(n_sel >= 0 && F >= 1, BOT_E_RANGE, "%s: n_sel=%lld F=%d", who, (long long)n_sel, F);
    if (n_sel == 0) return 0;
    BOT_REQUIRE(src && rows && dst, BOT_E_NULL, "%s: NULL pointer", who);
    BOT_REQUIRE(lds_ >= F && ldd >= F, BOT_E_RANGE, "%s: row stride smaller than F", who);
    const int64_t blocks = (n_sel * kWave + kBlock - 1) / kBlock;
    const int vec = pick_vec(F, {lds_, ldd}, {src, dst});
    if (vec == 4) hipLaunchKernelGGL((rows_kernel<4, ADD>), dim3((unsigned)blocks), dim3(kBlock), 0, st, src, lds_, rows, n_sel, F, dst, ldd);
    else if (vec == 2) hipLaunchKernelGGL((rows_kernel<2, ADD>), dim3((unsigned)blocks), dim3(kBlock), 0, st, src, lds_, rows, n_sel, F, dst, ldd);
    else hipLaunchKernelGGL((rows_kernel<1, ADD>), dim3((unsigned)blocks), dim3(kBlock), 0, st, src, lds_, rows, n_sel, F, dst, ldd);
    return hip_status(who);
}

}  // namespace bot

extern "C" {

int bot_gather_rows_f32(const float* x, int64_t ldx, const int32_t* rows, int64_t n_sel, int32_t F, float* out, int64_t ldo,
                        bot_stream_t stream) {
    return bot::launch_rows<false>("gather_rows", x, ldx, rows, n_sel, F, out, ldo, (hipStream_t)stream);
}

int bot_scatter_add_rows_f32(float* x, int64_t ldx, const int32_t* rows, int64_t n_sel, int32_t F, const float* vals,
                             int64_t ldv, bot_stream_t stream) {
    return bot::launch_rows<true>("scatter_add_rows", vals, ldv, rows, n_sel, F, x, ldx, (hipStream_t)stream);
}

}  // extern "C"
