// The train step's glue around the layers (round 4): what src/no-sampling/run.py does with a dozen small tensor ops per step — the
// label / prediction split of the training nodes (run.py:256-267), `add_labels` + the stack's input dropout (run.py:240-243,
// models.py:711), the per-node loss and its gradient (run.py:229-237) and the optimizer update (run.py:331-338, torch.optim.RMSprop) —
// as four launches instead of ~45 (`at::native::*` fill / index_put / cat / dropout / log_softmax / nll / where / foreach kernels:
// 0.6 ms of the 12.9 ms config-2 step, profiles/r03_bench_arxiv_kernel_stats.csv).  All deterministic: fixed-order reductions, no atomics.
#include <algorithm>

#include "common.h"

namespace bot {
namespace {

__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const uint64_t* off) {      // dense.hip: the word a captured graph bumps between replays
    return off ? seed + off[0] * 0x9E3779B97F4A7C15ull : seed;
}

__device__ __forceinline__ float uniform01(uint32_t w) { return (w >> 8) * (1.0f / 16777216.0f); }

// ---- split of the training nodes.  keep[i] = mask[i] (given) or u_i < mask_rate (Philox word i & 3 of block i >> 2).
//   use_labels:  code[train_idx[i]] = keep ? label : -1  (the node's label is an INPUT this step), wn[...] = keep ? 0 : 1 (else it is predicted)
//   otherwise:   wn[train_idx[i]] = keep ? 1 : 0         (run.py:265-267: the loss runs over train_idx[mask])
// count[0] = number of nodes with wn = 1: per-workgroup integer partials (exact, any order) folded by the last kernel of the call in
// slot order.  (One 1024-thread workgroup for all ~1e5 training nodes was a chain of dependent gathers: 0.2 ms.)
constexpr int kSplitBlocks = 128;
__global__ __launch_bounds__(256) void label_split_kernel(const int64_t* train_idx, int64_t n_train, const int64_t* labels, int64_t ldl, const uint8_t* mask,
                                                          float mask_rate, uint64_t seed, const uint64_t* seed_offset, int use_labels, int32_t* code,
                                                          float* wn, int32_t* partial) {
    __shared__ int part[4];
    const uint64_t s = eff_seed(seed, seed_offset);
    int mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_train; i += (int64_t)gridDim.x * 256) {
        bool keep;
        if (mask) {
            keep = mask[i] != 0;
        } else {
            uint32_t w[4];
            Philox::gen(s, (uint64_t)(i >> 2), w);
            keep = uniform01(w[i & 3]) < mask_rate;
        }
        const int64_t n = train_idx[i];
        const bool pred = use_labels ? !keep : keep;
        if (code) code[n] = (use_labels && keep) ? (int32_t)labels[n * ldl] : -1;
        wn[n] = pred ? 1.f : 0.f;
        mine += pred ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(64) void label_count_kernel(const int32_t* partial, int nb, float* count) {
    int t = 0;
    for (int i = threadIdx.x; i < nb; i += 64) t += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (threadIdx.x == 0) count[0] = (float)t;
}

// ---- layer-0 input: out[n, :] = dropout_p( [ feat[n, :F] | onehot(code[n])[:C] ] ), element (n, c) dropped by word c & 3 of Philox block
// n * ceil(W / 4) + c / 4 (W = F + C; the convention of the fused BatchNorm dropout), survivors scaled by 1 / (1 - p).
__global__ __launch_bounds__(256) void build_input_kernel(const float* feat, int64_t ldf, int64_t n, int F, int C, const int32_t* code, float p,
                                                          uint64_t seed, const uint64_t* seed_offset, float* out, int64_t ldo) {
    const int W = F + C, nquad = (W + 3) >> 2;
    const uint64_t s = eff_seed(seed, seed_offset);
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const int64_t total = n * (int64_t)nquad;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t r = q / nquad;
        const int c0 = (int)(q - r * nquad) * 4;
        const int lab = code ? code[r] : -1;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (p > 0.f) Philox::gen(s, (uint64_t)q, w);
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = c0 + t;
            float x = 0.f;
            if (c < F) x = feat[r * ldf + c];
            else if (c < W) x = (c - F == lab) ? 1.f : 0.f;
            v[t] = (p > 0.f && uniform01(w[t]) < p) ? 0.f : x * scale;
        }
        float* o = out + r * ldo + c0;
        if (c0 + 3 < W && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (c0 + t < W) o[t] = v[t];
        }
    }
}

// ---- per-node loss of run.py:229-236 and its gradient, 16 lanes per node (4 nodes per wave; a lane holds classes l, l + 16, ...):
//   ce = logsumexp(x) - x[label];  kind 0: y = ce;  1 (loge): y = log(eps + ce) - log(eps);  2 (savage): y = (1 - exp(-ce))^2
//   y_out[n] = wn[n] > 0 ? y : 0;   dx[n, c] = wn[n] > 0 ? (dy/dce) (softmax(x)[c] - [c == label]) / count : 0
// Nodes with wn = 0 contribute nothing whatever their label holds (placeholders: clamped into range before use).
// (One wave per node, lanes over 40 classes: 65 us at N = 169 343 - 169 k waves for 27 MB; this form: 4x fewer, every exp computed once.)
constexpr int kLossMaxPerLane = 8;          // classes per lane: C <= 128
__global__ __launch_bounds__(256) void node_loss_kernel(const float* x, int64_t ldx, int64_t n, int C, const int64_t* labels, int64_t ldl, const float* wn,
                                                        const float* count, int kind, float eps, float* y_out, float* dx, int64_t lddx, int64_t n_pad) {
    const int l16 = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (row >= n_pad) return;
    if (row >= n) {                 // padding of y_out up to a multiple of 64 (the fixed-order sum that follows reads whole rows of 64)
        if (l16 == 0) y_out[row] = 0.f;
        return;
    }
    const bool on = wn[row] > 0.f;
    const float* xr = x + row * ldx;
    float v[kLossMaxPerLane];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kLossMaxPerLane; ++k) {
        const int c = l16 + 16 * k;
        v[k] = c < C ? xr[c] : -INFINITY;
        m = fmaxf(m, v[k]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < kLossMaxPerLane; ++k) {
        v[k] = l16 + 16 * k < C ? expf(v[k] - m) : 0.f;          // e^(x - max), reused for the softmax below
        se += v[k];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o, 16);
    int lab = (int)labels[row * ldl];
    lab = min(max(lab, 0), C - 1);
    const float ce = m + logf(se) - xr[lab];
    float y, dydce;
    if (kind == 1) {
        y = logf(eps + ce) - logf(eps);
        dydce = 1.f / (eps + ce);
    } else if (kind == 2) {
        const float e = expf(-ce);
        y = (1.f - e) * (1.f - e);
        dydce = 2.f * (1.f - e) * e;
    } else {
        y = ce;
        dydce = 1.f;
    }
    if (l16 == 0) y_out[row] = on ? y : 0.f;
    if (dx) {
        const float g = on ? dydce / count[0] : 0.f;
        const float inv = 1.f / se;
        float* dr = dx + row * lddx;
#pragma unroll
        for (int k = 0; k < kLossMaxPerLane; ++k) {
            const int c = l16 + 16 * k;
            if (c < C) dr[c] = g * (v[k] * inv - (c == lab ? 1.f : 0.f));
        }
    }
}

// ---- torch.optim.RMSprop's update (momentum 0, not centered) for up to kMaxTensors parameters in one launch:
//   g += weight_decay * p;  sq = alpha * sq + (1 - alpha) * g * g;  p -= lr * g / (sqrt(sq) + eps)
// `lr_dev` (optional) overrides `lr` with a device scalar (a captured step whose learning rate moves between replays).
constexpr int kMaxTensors = 48;
struct RmsArgs {
    float* p[kMaxTensors];
    const float* g[kMaxTensors];
    float* sq[kMaxTensors];
    int first_block[kMaxTensors + 1];    // tensor t owns blocks [first_block[t], first_block[t + 1]); 1024 elements per block
    int64_t numel[kMaxTensors];
    int n_tensors;
    float lr, alpha, eps, weight_decay;
    const float* lr_dev;
};

__global__ __launch_bounds__(256) void rmsprop_kernel(RmsArgs a) {
    int t = 0;
    while (t + 1 < a.n_tensors && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
    const int64_t i0 = ((int64_t)(blockIdx.x - a.first_block[t]) * 256 + threadIdx.x) * 4;
    const float lr = a.lr_dev ? a.lr_dev[0] : a.lr;
    float* p = a.p[t];
    const float* g = a.g[t];
    float* sq = a.sq[t];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + e;
        if (i < a.numel[t]) {
            float gi = g[i];
            if (a.weight_decay != 0.f) gi = fmaf(a.weight_decay, p[i], gi);
            const float s = a.alpha * sq[i] + (1.f - a.alpha) * gi * gi;          // torch: mul_(alpha).addcmul_(g, g, value = 1 - alpha)
            sq[i] = s;
            p[i] = p[i] - lr * (gi / (sqrtf(s) + a.eps));                          // addcdiv_(g, sqrt(s) + eps, value = -lr)
        }
    }
}

}  // namespace
}  // namespace bot

extern "C" int bot_label_split_f32(const int64_t* train_idx, int64_t n_train, const int64_t* labels, int64_t ldl, const uint8_t* mask, float mask_rate,
                                   uint64_t seed, const uint64_t* seed_offset, int32_t use_labels, int32_t* code, float* wn, float* count,
                                   int32_t* workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_train >= 0 && train_idx && wn && count && workspace && (!use_labels || (code && labels)), -1, "label_split: null pointer");
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kSplitBlocks, (n_train + 255) / 256));
    hipLaunchKernelGGL(label_split_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, train_idx, n_train, labels, ldl, mask, mask_rate, seed, seed_offset,
                       (int)use_labels, code, wn, workspace);
    hipLaunchKernelGGL(label_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, nb, count);
    return hip_status("label_split");
}

extern "C" int bot_build_input_f32(const float* feat, int64_t ldf, int64_t n, int32_t F, int32_t C, const int32_t* code, float p, uint64_t seed,
                                   const uint64_t* seed_offset, float* out, int64_t ldo, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && F >= 0 && C >= 0 && F + C > 0 && feat && out && ldo >= F + C && ldf >= F && p >= 0.f && p < 1.f && (C == 0 || code), -1,
                "build_input: bad arguments");
    if (n == 0) return 0;
    const int64_t total = n * (int64_t)((F + C + 3) / 4);
    hipLaunchKernelGGL(build_input_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65536)), dim3(256), 0, (hipStream_t)stream, feat, ldf, n, (int)F,
                       (int)C, code, p, seed, seed_offset, out, ldo);
    return hip_status("build_input");
}

extern "C" int bot_node_loss_f32(const float* x, int64_t ldx, int64_t n, int32_t C, const int64_t* labels, int64_t ldl, const float* wn, const float* count,
                                 int32_t kind, float eps, float* y, int64_t n_pad, float* dx, int64_t lddx, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && C > 0 && C <= 16 * kLossMaxPerLane && x && labels && wn && count && y && n_pad >= n && kind >= 0 && kind <= 2 && (!dx || lddx >= C),
                -1, "node_loss: bad arguments (1 <= C <= %d)", 16 * kLossMaxPerLane);
    if (n_pad == 0) return 0;
    hipLaunchKernelGGL(node_loss_kernel, dim3((unsigned)((n_pad * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, n, (int)C, labels, ldl, wn,
                       count, (int)kind, eps, y, dx, lddx, n_pad);
    return hip_status("node_loss");
}

extern "C" int bot_rmsprop_step_f32(int32_t n_tensors, float* const* params, const float* const* grads, float* const* square_avg, const int64_t* numel,
                                    float lr, const float* lr_dev, float alpha, float eps, float weight_decay, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_tensors >= 0 && n_tensors <= kMaxTensors, -1, "rmsprop_step: at most %d tensors per call (got %d)", kMaxTensors, n_tensors);
    if (n_tensors == 0) return 0;
    RmsArgs a;
    int blocks = 0;
    for (int t = 0; t < n_tensors; ++t) {
        BOT_REQUIRE(params[t] && grads[t] && square_avg[t] && numel[t] >= 0, -1, "rmsprop_step: null pointer in tensor %d", t);
        a.p[t] = params[t], a.g[t] = grads[t], a.sq[t] = square_avg[t], a.numel[t] = numel[t];
        a.first_block[t] = blocks;
        blocks += (int)((numel[t] + 1023) / 1024);
    }
    a.first_block[n_tensors] = blocks;
    a.n_tensors = n_tensors, a.lr = lr, a.alpha = alpha, a.eps = eps, a.weight_decay = weight_decay, a.lr_dev = lr_dev;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hip_status("rmsprop_step");
}
