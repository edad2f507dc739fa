// Edge-feature attention term of the ogbn-proteins GAT, fused (gfx950).
//
// Reference, per layer (src/ogbn-proteins/models.py:244-248 and :130-133):
//     emb  = relu(edge_encoder_i(efeat))        Linear(8 -> 16) + bias     [E,16]   (5 GB at E = 77.7 M)
//     ee   = attn_edge_fc(emb)                  Linear(16 -> H), no bias    [E,H]
// and `ee` is added to the attention logits.  Done with library ops that materialises [E,16] four times per layer per
// step (forward, ReLU mask, two backward intermediates).  Here the 8->16->H MLP is evaluated per edge in registers:
//
//   edge_mlp_fwd : ee[e,:] = W2 . relu(W1 . ef[e,:] + b1)        reads 32 B, writes 4H B per edge, 224 FMAs
//   edge_mlp_bwd : the three weight gradients are tiny GEMMs with K = E,
//                      dW2^T[j,h] = sum_e r[e,j] * dz[e,h]          (16 x H)
//                      [dW1 | db1][j,:] = sum_e du[e,j] * [ef[e,:] | 1]   (16 x 9),  du = (dz . W2) * [pre > 0]
//                  run on the fp32 MFMA (v_mfma_f32_16x16x4_f32: one instruction folds 4 edges into a 16x16 accumulator
//                  tile).  Lane l = (hidden unit j = l & 15, edge slot k = l >> 4) recomputes r and du for its (edge, unit)
//                  with its row of W1 / column of W2 held in registers — exactly the A-operand layout of the instruction —
//                  so nothing crosses lanes and nothing of size E x 16 ever exists in memory.  Per-wave accumulator tiles go
//                  to a workspace and are summed in wave order (deterministic, no atomics).
//
// Edge features are consumed in CSC position order (permuted once per graph; they are constant inputs), so both kernels
// stream with unit stride.  HBM roofline: fwd 4*E*(8+H) bytes, bwd 4*E*(8+H) bytes.
#include "common.h"

namespace bot {

constexpr int kI = 8;    // raw edge features   (ogbn-proteins: 8)
constexpr int kJ = 16;   // edge embedding      (ogbn-proteins/gat.py:83 edge_emb=16)
constexpr int kMlpBlocks = 2048;

template <int H>
__global__ __launch_bounds__(kBlock) void edge_mlp_fwd_kernel(const float* __restrict__ ef, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, const float* __restrict__ W2,
                                                             int64_t E, float* __restrict__ out, bool wide) {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < E; e += (int64_t)gridDim.x * kBlock) {
        float f[kI];
        vload<4>(*reinterpret_cast<float(*)[4]>(&f[0]), ef + e * kI);
        vload<4>(*reinterpret_cast<float(*)[4]>(&f[4]), ef + e * kI + 4);
        float r[kJ];
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            float acc = b1[j];
#pragma unroll
            for (int i = 0; i < kI; ++i) acc = fmaf(W1[j * kI + i], f[i], acc);
            r[j] = fmaxf(acc, 0.f);
        }
        float o[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < kJ; ++j) v = fmaf(W2[h * kJ + j], r[j], v);
            o[h] = v;
        }
        // one record per edge: 8- / 16-byte stores where H allows (a 4-byte store per head touches the line H times)
        if constexpr (H % 4 == 0) {
            if (wide) {
#pragma unroll
                for (int h = 0; h < H; h += 4) *reinterpret_cast<float4*>(out + e * H + h) = make_float4(o[h], o[h + 1], o[h + 2], o[h + 3]);
                continue;
            }
        } else if constexpr (H % 2 == 0) {
            if (wide) {
#pragma unroll
                for (int h = 0; h < H; h += 2) *reinterpret_cast<float2*>(out + e * H + h) = make_float2(o[h], o[h + 1]);
                continue;
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) out[e * H + h] = o[h];
    }
}

using f32x4 = __attribute__((ext_vector_type(4))) float;

// part[wave][0][j][n] : dW2^T tile (n = head), part[wave][1][j][n] : [dW1 | db1] tile (n < 8: input i, n == 8: bias)
template <int H>
__global__ __launch_bounds__(kBlock) void edge_mlp_bwd_kernel(const float* __restrict__ ef, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, const float* __restrict__ W2,
                                                             const float* __restrict__ dz, int64_t E, bool wide,
                                                             float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, k = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
    float w1[kI], w2[8];
#pragma unroll
    for (int i = 0; i < kI; ++i) w1[i] = W1[j * kI + i];
#pragma unroll
    for (int h = 0; h < 8; ++h) w2[h] = h < H ? W2[h * kJ + j] : 0.f;
    const float bj = b1[j];
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // One MFMA step folds 4 edges; a wave takes U steps (16 edges) per trip.  Round 6: the 16 edges' records (16 x 32 B of features, 16 x 4H B
    // of dz: both contiguous in CSC order) are fetched ONCE per wave - lanes 0..31 one 16-byte chunk of the features each, lanes 32.. one
    // 16-byte chunk of the dz span - one trip ahead, and handed to the (unit j, edge slot k) lanes through 1 KB of LDS per wave.  Before,
    // the 16 lanes of an edge slot each loaded the slot's whole record: 64 unique bytes per load instruction, the launch bound by the
    // address path (3.0 ms per call at E = 79 M, 4.4 GB); the MFMA sequence and the lane -> edge mapping are unchanged (bitwise the same sums).
    constexpr int U = 4;
    __shared__ __attribute__((aligned(16))) float stage[kBlock / 64][256];
    float* sf = stage[threadIdx.x >> 6];        // [16 edges][8] features
    float* sz = sf + 128;                       // [16 edges][H] dz records, linear
    const int zt = lane - 32;                   // lanes 32..: floats [4 zt, 4 zt + 4) of the trip's 16 H dz floats
    const int64_t EH = E * H;
    auto fetch = [&](int64_t base) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 32) {
            const int64_t e = base + (lane >> 1);
            if (e < E) v = *reinterpret_cast<const float4*>(ef + e * kI + (lane & 1) * 4);
        } else if (4 * zt < 16 * H) {
            const int64_t o = base * H + 4 * zt;        // (base is a multiple of 16: the span starts 64 H bytes into dz - 16-byte aligned with dz)
            if (wide && o + 3 < EH) {
                v = *reinterpret_cast<const float4*>(dz + o);
            } else {
                if (o < EH) v.x = dz[o];
                if (o + 1 < EH) v.y = dz[o + 1];
                if (o + 2 < EH) v.z = dz[o + 2];
                if (o + 3 < EH) v.w = dz[o + 3];
            }
        }
        return v;
    };
    const int64_t stride = n_waves * (4 * U);
    // (one trip ahead: 2.14 ms per call; three ahead: 2.34 - the launch is a balance of vector issue, LDS and MFMA, not latency-bound)
    float4 cur = fetch(wave * (4 * U));
    for (int64_t base = wave * (4 * U); base < E; base += stride) {
        const float4 nxt = base + stride < E ? fetch(base + stride) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 32) *reinterpret_cast<float4*>(sf + lane * 4) = cur;
        else if (4 * zt < 16 * H) *reinterpret_cast<float4*>(sz + zt * 4) = cur;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // lanes read each other's chunks below
        float f[U][kI], dzv[U][8], bz[U], bf[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int slot = u * 4 + k;
            live[u] = base + slot < E;          // (a tail slot holds zeros: its contributions are zeroed below as before)
            const float4 f0 = *reinterpret_cast<const float4*>(sf + slot * kI), f1 = *reinterpret_cast<const float4*>(sf + slot * kI + 4);
            f[u][0] = f0.x, f[u][1] = f0.y, f[u][2] = f0.z, f[u][3] = f0.w, f[u][4] = f1.x, f[u][5] = f1.y, f[u][6] = f1.z, f[u][7] = f1.w;
#pragma unroll
            for (int h = 0; h < 8; ++h) dzv[u][h] = 0.f;
            if constexpr (H % 4 == 0) {         // the slot's 4 H-byte record as 16- / 8-byte LDS reads where H allows
#pragma unroll
                for (int h = 0; h < H; h += 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(sz + slot * H + h);
                    dzv[u][h] = t4.x, dzv[u][h + 1] = t4.y, dzv[u][h + 2] = t4.z, dzv[u][h + 3] = t4.w;
                }
            } else if constexpr (H % 2 == 0) {
#pragma unroll
                for (int h = 0; h < H; h += 2) {
                    const float2 t2 = *reinterpret_cast<const float2*>(sz + slot * H + h);
                    dzv[u][h] = t2.x, dzv[u][h + 1] = t2.y;
                }
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h) dzv[u][h] = sz[slot * H + h];
            }
            // B operands: column n = lane & 15 of this lane's edge slot — picked out of the registers just loaded (one more LDS read each
            // instead of the select chains: 2.14 -> 2.23 ms per call, the LDS is as busy as the vector pipe here)
            float zb = 0.f, fb = j == kI ? 1.f : 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) zb = j == h ? dzv[u][h] : zb;
#pragma unroll
            for (int i = 0; i < kI; ++i) fb = j == i ? f[u][i] : fb;
            bz[u] = zb;
            bf[u] = fb;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // every read of this trip's chunks is done before the next trip overwrites them
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float pre = bj, t = 0.f;
#pragma unroll
            for (int i = 0; i < kI; ++i) pre = fmaf(w1[i], f[u][i], pre);
#pragma unroll
            for (int h = 0; h < 8; ++h) t = fmaf(dzv[u][h], w2[h], t);      // (heads beyond H: 0 x 0, kept so that the sums stay bitwise what they were)
            const float r = live[u] ? fmaxf(pre, 0.f) : 0.f;
            const float du = (live[u] && pre > 0.f) ? t : 0.f;
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(r, bz[u], acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(du, bf[u], acc1, 0, 0, 0);
        }
        cur = nxt;
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* p = part + wave * 512;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = k * 4 + r;
        p[row * 16 + j] = acc2[r];
        p[256 + row * 16 + j] = acc1[r];
    }
}

// One workgroup per output element: 256 threads add the per-wave tiles in a fixed interleaved order, then a fixed-order
// LDS combine (deterministic).
__global__ __launch_bounds__(kBlock) void edge_mlp_bwd_final_kernel(const float* part, int64_t n_waves, int32_t H, float* dW1,
                                                                   float* db1, float* dW2) {
    __shared__ double lds[kBlock];
    const int t = blockIdx.x;  // 0..511
    double s = 0.0;
    for (int64_t w = threadIdx.x; w < n_waves; w += kBlock) s += (double)part[w * 512 + t];
    lds[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    s = 0.0;
    for (int i = 0; i < kBlock; ++i) s += lds[i];
    const int which = t >> 8, row = (t & 255) >> 4, col = t & 15;  // row = hidden unit j
    if (which == 0) {
        if (col < H) dW2[col * kJ + row] = (float)s;
    } else {
        if (col < kI) dW1[row * kI + col] = (float)s;
        else if (col == kI) db1[row] = (float)s;
    }
}

}  // namespace bot

extern "C" {

int64_t bot_edge_mlp_workspace_floats(void) { return (int64_t)bot::kMlpBlocks * (bot::kBlock / 64) * 512; }

int bot_edge_mlp_fwd_f32(const float* ef, int32_t I, const float* W1, const float* b1, int32_t J, const float* W2, int32_t H,
                         int64_t n_edges, float* out, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(I == kI && J == kJ, BOT_E_RANGE, "edge_mlp: only the 8 -> 16 -> H shape of ogbn-proteins is fused (got %d -> %d)", I, J);
    BOT_REQUIRE(H >= 1 && H <= 8 && n_edges >= 0, BOT_E_RANGE, "edge_mlp: H=%d n_edges=%lld", H, (long long)n_edges);
    if (n_edges == 0) return 0;
    BOT_REQUIRE(ef && W1 && b1 && W2 && out, BOT_E_NULL, "edge_mlp_fwd: NULL pointer");
    BOT_REQUIRE(aligned(ef, 16), BOT_E_ALIGN, "edge_mlp_fwd: ef must be 16-byte aligned");
    int64_t blocks = (n_edges + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = (hipStream_t)stream;
#define BOT_MLP_FWD(HH) hipLaunchKernelGGL((edge_mlp_fwd_kernel<HH>), dim3((unsigned)blocks), dim3(kBlock), 0, st, ef, W1, b1, W2, n_edges, out, aligned(out, 16))
    switch (H) {
        case 1: BOT_MLP_FWD(1); break;
        case 2: BOT_MLP_FWD(2); break;
        case 3: BOT_MLP_FWD(3); break;
        case 4: BOT_MLP_FWD(4); break;
        case 5: BOT_MLP_FWD(5); break;
        case 6: BOT_MLP_FWD(6); break;
        case 7: BOT_MLP_FWD(7); break;
        default: BOT_MLP_FWD(8); break;
    }
#undef BOT_MLP_FWD
    return hip_status("edge_mlp_fwd launch");
}

int bot_edge_mlp_bwd_f32(const float* ef, int32_t I, const float* W1, const float* b1, int32_t J, const float* W2, int32_t H,
                         const float* dz, int64_t n_edges, float* dW1, float* db1, float* dW2, float* workspace,
                         bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(I == kI && J == kJ, BOT_E_RANGE, "edge_mlp: only the 8 -> 16 -> H shape of ogbn-proteins is fused (got %d -> %d)", I, J);
    BOT_REQUIRE(H >= 1 && H <= 8 && n_edges >= 1, BOT_E_RANGE, "edge_mlp_bwd: H=%d n_edges=%lld", H, (long long)n_edges);
    BOT_REQUIRE(ef && W1 && b1 && W2 && dz && dW1 && db1 && dW2 && workspace, BOT_E_NULL, "edge_mlp_bwd: NULL pointer");
    BOT_REQUIRE(aligned(ef, 16), BOT_E_ALIGN, "edge_mlp_bwd: ef must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const bool wide = aligned(dz, 16);
#define BOT_MLP_BWD(HH) hipLaunchKernelGGL((edge_mlp_bwd_kernel<HH>), dim3(kMlpBlocks), dim3(kBlock), 0, st, ef, W1, b1, W2, dz, n_edges, wide, workspace)
    switch (H) {
        case 1: BOT_MLP_BWD(1); break;
        case 2: BOT_MLP_BWD(2); break;
        case 3: BOT_MLP_BWD(3); break;
        case 4: BOT_MLP_BWD(4); break;
        case 5: BOT_MLP_BWD(5); break;
        case 6: BOT_MLP_BWD(6); break;
        case 7: BOT_MLP_BWD(7); break;
        default: BOT_MLP_BWD(8); break;
    }
#undef BOT_MLP_BWD
    hipLaunchKernelGGL(edge_mlp_bwd_final_kernel, dim3(512), dim3(kBlock), 0, st, workspace, (int64_t)kMlpBlocks * (kBlock / 64), H, dW1,
                       db1, dW2);
    return hip_status("edge_mlp_bwd launch");
}

}  // extern "C"
