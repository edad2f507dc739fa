// Shared helpers for the gfx950 kernels behind include/bot_gnn.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bot_gnn.h"

namespace bot {

void set_error(const char* fmt, ...);
void set_kernel(const char* fmt, ...);
// scale[0] = 2^(14 - ceil(log2 max_i part[i])), scale[1] = 1 / scale[0]  (halves.hip; part: n non-negative floats on the device)
void launch_halves_scale(const float* part, int n, float* scale, hipStream_t st);  // name of the main device kernel a launch function dispatched (bot_last_kernel)

inline int hip_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

#define BOT_REQUIRE(cond, code, ...)    \
    do {                                \
        if (!(cond)) {                  \
            bot::set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

constexpr int kBlock = 256;   // 4 wave64 per workgroup
constexpr int kWave = 64;

// ---- vector types -------------------------------------------------------------------------
template <int VEC> struct Vec;
template <> struct Vec<1> { using type = float; };
template <> struct Vec<2> { using type = float2; };
template <> struct Vec<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void vload(float (&r)[VEC], const float* p) {
    using T = typename Vec<VEC>::type;
    T v = *reinterpret_cast<const T*>(p);
    if constexpr (VEC == 1) { r[0] = v; }
    if constexpr (VEC == 2) { r[0] = v.x; r[1] = v.y; }
    if constexpr (VEC == 4) { r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
}

// Streaming variants (non-temporal hint): for data touched once per launch (index / weight streams, output rows), so that it
// does not push the gather table out of the L2 / Infinity Cache.  Enabled per kernel where measured to help.
template <int VEC>
__device__ __forceinline__ void vstore_nt(float* p, const float (&r)[VEC]) {
    typedef float nv __attribute__((ext_vector_type(VEC)));  // the builtin wants a native vector type
    nv v;
#pragma unroll
    for (int t = 0; t < VEC; ++t) v[t] = r[t];
    __builtin_nontemporal_store(v, reinterpret_cast<nv*>(p));
}

template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&r)[VEC]) {
    using T = typename Vec<VEC>::type;
    T v;
    if constexpr (VEC == 1) { v = r[0]; }
    if constexpr (VEC == 2) { v.x = r[0]; v.y = r[1]; }
    if constexpr (VEC == 4) { v.x = r[0]; v.y = r[1]; v.z = r[2]; v.w = r[3]; }
    *reinterpret_cast<T*>(p) = v;
}

// Broadcast lane `j` of a LANES-wide group.  For full-wave groups the source lane is wave-uniform,
// so v_readlane puts the value in an SGPR and the dependent address math stays scalar.
template <int LANES>
__device__ __forceinline__ int group_bcast(int v, int j) {
    if constexpr (LANES == 64) return __builtin_amdgcn_readlane(v, j);
    else return __shfl(v, j, LANES);
}
template <int LANES>
__device__ __forceinline__ float group_bcast(float v, int j) {
    return __int_as_float(group_bcast<LANES>(__float_as_int(v), j));
}

template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = LANES / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, LANES);
    return v;
}
template <int LANES>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int m = LANES / 2; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, LANES));
    return v;
}

// Reduce 4 values per lane across a LANES-wide group; on return lane l holds the total of value number l % 4.
template <int LANES>
__device__ __forceinline__ float transpose_reduce4(float (&p)[4], int lane) {
    const bool h2 = lane & 2;
    float r[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = h2 ? p[i + 2] : p[i];
        const float send = h2 ? p[i] : p[i + 2];
        r[i] = keep + __shfl_xor(send, 2, LANES);
    }
    const bool h1 = lane & 1;
    float s = (h1 ? r[1] : r[0]) + __shfl_xor(h1 ? r[0] : r[1], 1, LANES);
#pragma unroll
    for (int m = 4; m < LANES; m <<= 1) s += __shfl_xor(s, m, LANES);
    return s;
}

// Reduce 16 values per lane across a LANES-wide group (LANES >= 16); on return lane l holds the total of value l % 16.
template <int LANES>
__device__ __forceinline__ float transpose_reduce16(float (&p)[16], int lane) {
    float q[8], r[4], s2[2];
    const bool b8 = lane & 8, b4 = lane & 4, b2 = lane & 2, b1 = lane & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = (b8 ? p[i + 8] : p[i]) + __shfl_xor(b8 ? p[i] : p[i + 8], 8, LANES);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (b4 ? q[i + 4] : q[i]) + __shfl_xor(b4 ? q[i] : q[i + 4], 4, LANES);
#pragma unroll
    for (int i = 0; i < 2; ++i) s2[i] = (b2 ? r[i + 2] : r[i]) + __shfl_xor(b2 ? r[i] : r[i + 2], 2, LANES);
    float s = (b1 ? s2[1] : s2[0]) + __shfl_xor(b1 ? s2[0] : s2[1], 1, LANES);
#pragma unroll
    for (int m = 16; m < LANES; m <<= 1) s += __shfl_xor(s, m, LANES);
    return s;  // value index = 8*b8 + 4*b4 + 2*b2 + b1 = lane % 16
}

struct Philox {
    // Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0,k1)
    static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0, c[1] = n1, c[2] = n2, c[3] = n3;
    }
    static __device__ __forceinline__ void gen(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
};

// |value| maxima as a by-product of the kernel that PRODUCES a matrix (round 3: the operand of a halves GEMM needs max|x| for its
// scale; a separate pass over a 1 GB gradient buffer costs 0.2 ms).  Non-negative floats order like their bit patterns, so the
// maximum is taken with INTEGER atomics on kAbsmaxSlots words (spread by workgroup id): exact, order-free, reproducible.
// fmaxf drops NaNs, like absmax_partial_kernel does.  The caller zeroes the slots; bot_halves_scale_from_slots_f32 reads them.
constexpr int kAbsmaxSlots = 64;
#ifdef __HIPCC__
__device__ __forceinline__ float wave_absmax(float m) {            // every lane of the wave must call this
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
__device__ __forceinline__ void absmax_publish(float wave_max, uint32_t* slots) {   // wave_max: the wave-reduced value
    if ((threadIdx.x & 63) == 0) {
        const uint32_t bits = __float_as_uint(wave_max);
        uint32_t* p = slots + ((blockIdx.x + 7 * blockIdx.y) & (kAbsmaxSlots - 1));   // 2-D grids (BatchNorm: 3 x 256 blocks) spread too
        if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, bits);   // most waves stop at the load
    }
}
#endif
// max|x| of a strided [n, F] matrix into the slots (halves.hip): the fall-back where a producer has no by-product form
void launch_absmax_slots(const float* x, int64_t ldx, int64_t n, int32_t F, uint32_t* slots, hipStream_t st);

// fp16-halves operand format (halves.hip): the second half of a LEFT operand is stored times 2^11, the third piece of a RIGHT
// operand is 2^-11 h1, so that the a2 b1 term keeps its bits for rows far below the matrix maximum.
constexpr float kHalvesShift = 2048.f;

// Vector width usable for a feature slab: every stride and base must keep VEC*4-byte alignment.
inline int pick_vec(int32_t D, std::initializer_list<int64_t> strides, std::initializer_list<const void*> ptrs) {
    for (int v : {4, 2}) {
        bool ok = (D % v) == 0;
        for (int64_t s : strides) ok = ok && (s % v) == 0;
        for (const void* p : ptrs) ok = ok && (p == nullptr || aligned(p, 4 * v));
        if (ok) return v;
    }
    return 1;
}

// Workgroup -> XCD note: blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).
// The kernels below order their work head-major so that one head's source slab (N*D*4 bytes,
// 169 MB at ogbn-arxiv H=3 D=250) is what is live in the 256 MiB Infinity Cache at a time.

}  // namespace bot
