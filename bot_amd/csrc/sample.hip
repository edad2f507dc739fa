// Exact-size uniform edge subsets for the training-time edge drop of the GAT layers (gfx950).
//
// Reference (src/no-sampling/models.py:528-532, src/ogbn-proteins/models.py:120-127): `perm = torch.randperm(E);
// eids = perm[int(E * edge_drop):]` — a uniformly random subset of exactly E - int(E * p) edges takes part in the layer.
// torch.randperm sorts E random keys (6.2 ms at E = 77.7 M on MI355X, 7.8 ms with the mask scatter; six times per
// S-proteins step).  Same distribution without the sort and without any E-sized temporary:
//
//   key(e) = 64 bits of Philox4x32-10(seed, e) — recomputed wherever it is needed, never stored;
//   drop the n_drop smallest keys: a most-significant-digit radix SELECT finds the n_drop-th smallest key T in six
//   histogram passes of 12 bits (only the first one touches every element's bin; later passes count the ~n / 4096^p
//   elements that still match the prefix), then one pass writes keep[e] = key(e) > T.
//
// Elements whose key EQUALS T (one, unless two 64-bit keys collide) go through a tiny tie list resolved in index order,
// so the kept count is exact and the result is a pure function of (n, n_keep, seed).  Integer atomics only: deterministic.
#include "common.h"

namespace bot {

constexpr int kSelBits = 12;
constexpr int kSelBins = 1 << kSelBits;
constexpr int kSelPasses = 6;   // 5 x 12 + 4 bits
constexpr int kMaxTies = 256;
constexpr int kSelBlocks = 2048;

struct SelState {
    uint64_t prefix;   // the top bits of T found so far
    int64_t k_rem;     // rank (1-based) of T among the elements that share `prefix`
    uint32_t n_ties;
    uint32_t pad;
    int64_t tie_idx[kMaxTies];
};

struct SelWorkspace {
    uint32_t hist[kSelPasses][kSelBins];
    SelState st;
};

__device__ __forceinline__ int sel_shift(int pass) { return 64 - kSelBits * (pass + 1) > 0 ? 64 - kSelBits * (pass + 1) : 0; }
__device__ __forceinline__ int sel_bits(int pass) { return pass < kSelPasses - 1 ? kSelBits : 64 - kSelBits * (kSelPasses - 1); }

// keys of elements 2q and 2q+1 from one Philox block
__device__ __forceinline__ void pair_keys(uint64_t seed, int64_t q, uint64_t (&key)[2]) {
    uint32_t r[4];
    Philox::gen(seed, (uint64_t)q, r);
    key[0] = ((uint64_t)r[0] << 32) | r[1];
    key[1] = ((uint64_t)r[2] << 32) | r[3];
}

__global__ __launch_bounds__(kBlock) void sel_hist_kernel(int64_t n, uint64_t seed, int pass, SelWorkspace* ws) {
    __shared__ uint32_t h[kSelBins];
    for (int i = threadIdx.x; i < kSelBins; i += kBlock) h[i] = 0;
    __syncthreads();
    const int shift = sel_shift(pass), bits = sel_bits(pass);
    const uint64_t prefix = ws->st.prefix;  // written by the previous scan kernel (stream order)
    const int64_t pairs = (n + 1) / 2;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < pairs; q += (int64_t)gridDim.x * kBlock) {
        uint64_t key[2];
        pair_keys(seed, q, key);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (2 * q + t >= n) continue;
            if (pass > 0 && (key[t] >> (shift + bits)) != prefix) continue;
            atomicAdd(&h[(key[t] >> shift) & ((1u << bits) - 1)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSelBins; i += kBlock)
        if (h[i]) atomicAdd(&ws->hist[pass][i], h[i]);
}

// one workgroup: the bin of this pass that holds rank k_rem
__global__ __launch_bounds__(kBlock) void sel_scan_kernel(int pass, SelWorkspace* ws) {
    __shared__ int64_t part[kBlock];
    constexpr int PER = kSelBins / kBlock;
    const uint32_t* h = ws->hist[pass];
    int64_t s = 0;
    for (int i = 0; i < PER; ++i) s += h[threadIdx.x * PER + i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    int64_t k = ws->st.k_rem, cum = 0;
    int t = 0;
    for (; t < kBlock - 1 && cum + part[t] < k; ++t) cum += part[t];
    int b = t * PER;
    for (; b < t * PER + PER - 1 && cum + h[b] < k; ++b) cum += h[b];
    ws->st.prefix = (ws->st.prefix << sel_bits(pass)) | (uint64_t)b;
    ws->st.k_rem = k - cum;
}

__global__ __launch_bounds__(kBlock) void sel_write_kernel(int64_t n, uint64_t seed, SelWorkspace* ws, uint8_t* keep) {
    const uint64_t T = ws->st.prefix;
    const int64_t pairs = (n + 1) / 2;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < pairs; q += (int64_t)gridDim.x * kBlock) {
        uint64_t key[2];
        pair_keys(seed, q, key);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t e = 2 * q + t;
            if (e >= n) continue;
            keep[e] = key[t] >= T;
            if (key[t] == T) {
                const uint32_t slot = atomicAdd(&ws->st.n_ties, 1u);
                if (slot < kMaxTies) ws->st.tie_idx[slot] = e;
            }
        }
    }
}

// of the elements whose key equals T, the k_rem with the lowest index are dropped (one element unless keys collide)
__global__ void sel_ties_kernel(SelWorkspace* ws, uint8_t* keep) {
    const int m = ws->st.n_ties < (uint32_t)kMaxTies ? (int)ws->st.n_ties : kMaxTies;
    int64_t* idx = ws->st.tie_idx;
    for (int i = 1; i < m; ++i) {  // insertion sort: the append order of the atomics is not reproducible, the index order is
        const int64_t v = idx[i];
        int j = i - 1;
        for (; j >= 0 && idx[j] > v; --j) idx[j + 1] = idx[j];
        idx[j + 1] = v;
    }
    for (int i = 0; i < m && i < ws->st.k_rem; ++i) keep[idx[i]] = 0;
}

__global__ void sel_init_kernel(SelWorkspace* ws, int64_t k) { ws->st.k_rem = k; }  // after the memset that clears the rest

__global__ __launch_bounds__(kBlock) void sel_fill_kernel(int64_t n, uint8_t v, uint8_t* keep) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) keep[i] = v;
}

}  // namespace bot

extern "C" {

int64_t bot_random_keep_workspace_bytes(void) { return (int64_t)sizeof(bot::SelWorkspace); }

int bot_random_keep_u8(int64_t n, int64_t n_keep, uint64_t seed, uint8_t* keep, void* workspace, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n >= 0 && n_keep >= 0 && n_keep <= n, BOT_E_RANGE, "random_keep: n=%lld n_keep=%lld", (long long)n, (long long)n_keep);
    if (n == 0) return 0;
    BOT_REQUIRE(keep != nullptr, BOT_E_NULL, "random_keep: keep is NULL");
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > kSelBlocks) blocks = kSelBlocks;
    if (n_keep == n || n_keep == 0) {
        hipLaunchKernelGGL(sel_fill_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, n, (uint8_t)(n_keep == n), keep);
        return hip_status("random_keep fill launch");
    }
    BOT_REQUIRE(workspace != nullptr && aligned(workspace, 16), BOT_E_NULL, "random_keep: workspace is NULL or misaligned");
    SelWorkspace* ws = reinterpret_cast<SelWorkspace*>(workspace);
    if (hipMemsetAsync(ws, 0, sizeof(SelWorkspace), st) != hipSuccess) return hip_status("random_keep memset");
    hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(1), 0, st, ws, n - n_keep);
    for (int pass = 0; pass < kSelPasses; ++pass) {
        hipLaunchKernelGGL(sel_hist_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, n, seed, pass, ws);
        hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(kBlock), 0, st, pass, ws);
    }
    hipLaunchKernelGGL(sel_write_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, n, seed, ws, keep);
    hipLaunchKernelGGL(sel_ties_kernel, dim3(1), dim3(1), 0, st, ws, keep);
    return hip_status("random_keep launch");
}

}  // extern "C"
