// Inference-only GAT layer for gfx950 (the forward the reference's `evaluate()` runs every epoch,
// src/no-sampling/run.py:290-322): attention logits, leaky-ReLU, per-destination softmax, weighted aggregation, residual add,
// eval-mode BatchNorm (a per-column affine) or bias, and ReLU in ONE sweep over the in-edges —
//
//   z[k,h]   = el[indices[k],h] (+ er[r,h]) (+ ee[k,h]);   e = leaky_relu(z, slope)                 models.py:517-526
//   a[k,h]   = softmax_k e[k,h]                                                                     models.py:544
//   out[r,:] = act( (sum_k a[k,h] * ew[k] * x[indices[k],h,:] + addend[r,h,:]) * scale + shift )    models.py:547-560, 726-730
//
// Nothing edge-sized is written: the softmax weights live in registers between the logit pass and the gather (the training
// forward writes a [nnz,H] and the sign bytes for a backward that never comes here), and the [n,H*D] pre-BatchNorm tensor is
// never materialised.  The softmax is the online form over 64-edge trips (running max / running sum, accumulators rescaled
// when the max moves), so a row is swept once; rows longer than the plan's chunk get their (max, 1/sum) from a small
// workgroup-per-row pass first, so that their chunks can be summed independently and combined in slot order (deterministic).
//
// Layouts as in spmm.hip: `gat_infer_rows_kernel` = one wavefront per work item covering ALL heads (a head = HL lanes or CPH
// whole chunks): a neighbour row is one contiguous H*D*4-byte read, the per-head weight of an edge reaches the lanes as a
// v_readlane broadcast (an SGPR; a v_cndmask per extra head when heads share a chunk); `gat_infer_heads_kernel` = one lane
// group per (item, head), head-major: H = 1 (the output layer), plain sums (el == NULL: GraphConv-style aggregation with
// optional per-edge weights) and shapes the all-heads layout does not fit.
//
// HBM roofline: algorithmic bytes = 4*[n_src*H*D + n*H*D (+ n*H*D addend) + nnz + (n+1) + n_src*H] — no nnz*H term.
#include "common.h"

namespace bot {

struct InferArgs {
    const int32_t* indptr;
    const int32_t* indices;
    const int4* items;
    int64_t n_items;
    const float* x;
    int64_t ldx, hsx;
    const float* el;
    int64_t ldel;
    const float* er;
    int64_t lder;
    const float* ee;   // [nnz, H] position order, may be NULL
    const float* ew;   // [nnz] position order, may be NULL
    float slope;
    int32_t H, D;
    const float* addend;
    int64_t lda, hsa;
    const float* scale;  // [H*D] or NULL (= 1)
    const float* shift;  // [H*D] or NULL (= 0)
    int32_t relu;
    float* out;
    int64_t ldo, hso;
    float* rowstat;  // [n_slots][H][2]: (max, 1/sum) of the long row a slot belongs to
    float* partial;
    int64_t ldp;
    const int32_t* long_rows;
    const int32_t* long_ptr;
    uint32_t* absmax;  // optional by-product: max|out| into kAbsmaxSlots words (common.h absmax_publish)
};

__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// epilogue of one VEC-wide piece of an output row: residual, per-column affine, ReLU
template <int VEC>
__device__ __forceinline__ void infer_epilogue(const InferArgs& a, int row, int head, int e, float (&acc)[VEC]) {
    if (a.addend) {
        float r[VEC];
        vload<VEC>(r, a.addend + (int64_t)row * a.lda + (int64_t)head * a.hsa + e);
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] += r[t];
    }
    const int col = head * a.D + e;
    if (a.scale) {
        float s[VEC];
        vload<VEC>(s, a.scale + col);
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] *= s[t];
    }
    if (a.shift) {
        float s[VEC];
        vload<VEC>(s, a.shift + col);
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] += s[t];
    }
    if (a.relu) {
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[t] = fmaxf(acc[t], 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// all-heads layout
// ---------------------------------------------------------------------------------------------------------------------
template <int VEC, int HL, int NCHUNK, int CPH>
__global__ __launch_bounds__(kBlock) void gat_infer_rows_kernel(InferArgs a) {
    static_assert(CPH == 1 || HL == 64, "multi-chunk heads use whole waves");
    static_assert(NCHUNK % CPH == 0, "whole heads only");
    constexpr int U = 4;
    constexpr int HPC = 64 / HL;                           // heads per chunk (CPH == 1)
    constexpr int NSLOT = NCHUNK / CPH;                    // head slots per lane segment
    constexpr int HT = NSLOT * HPC < 8 ? NSLOT * HPC : 8;  // heads the layout can hold (the host checks H <= HT)
    const float NEG_INF = -__builtin_inff();
    const int lane = threadIdx.x & 63;
    const int64_t item = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (item >= a.n_items) return;
    const int4 it = a.items[item];
    const int row = __builtin_amdgcn_readfirstlane(it.x), beg = __builtin_amdgcn_readfirstlane(it.y);
    const int end = __builtin_amdgcn_readfirstlane(it.z), slot = __builtin_amdgcn_readfirstlane(it.w);
    const int hl = lane & (HL - 1);
    const int seg = lane / HL;  // which of the HPC heads of a chunk this lane serves
    int xoff[NCHUNK], hd[NCHUNK], el[NCHUNK];
    bool act[NCHUNK];
    float acc[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int head = CPH > 1 ? c / CPH : c * HPC + seg;
        const int e = CPH > 1 ? ((c % CPH) * 64 + lane) * VEC : hl * VEC;
        act[c] = head < a.H && e < a.D;
        hd[c] = head < a.H ? head : 0;
        el[c] = e;
        xoff[c] = act[c] ? (int)(head * a.hsx) + e : 0;  // idle lanes re-read element 0: in bounds, never stored
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[c][t] = 0.f;
    }
    // softmax state per head, wave-uniform.  Long-row chunks (slot >= 0) start from the row's final (max, 1/sum).
    float m[HT], s[HT], erv[HT];
#pragma unroll
    for (int h = 0; h < HT; ++h) {
        const bool on = h < a.H;
        erv[h] = (a.er && on) ? a.er[(int64_t)row * a.lder + h] : 0.f;
        m[h] = (slot >= 0 && on && a.el) ? a.rowstat[((int64_t)slot * a.H + h) * 2] : NEG_INF;
        s[h] = (slot >= 0 && on && a.el) ? a.rowstat[((int64_t)slot * a.H + h) * 2 + 1] : 0.f;
    }
    for (int k0 = beg; k0 < end; k0 += 64) {
        const int k = k0 + lane;
        const bool valid = k < end;
        const int idx = valid ? a.indices[k] : 0;
        float p[HT];  // this lane's edge: weight per head (unnormalised for whole rows)
        if (a.el) {
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                float z = NEG_INF;
                if (valid && h < a.H) {
                    z = a.el[(int64_t)idx * a.ldel + h] + erv[h];
                    if (a.ee) z += a.ee[(int64_t)k * a.H + h];
                    z = z > 0.f ? z : z * a.slope;
                }
                p[h] = z;
            }
            if (slot < 0) {
                float sc[HT];
#pragma unroll
                for (int h = 0; h < HT; ++h) {
                    const float mn = fmaxf(m[h], uniform(group_max<64>(p[h])));
                    sc[h] = h < a.H ? __expf(m[h] - mn) : 1.f;          // first trip: exp(-inf) = 0, accumulators are 0 anyway
                    p[h] = (valid && h < a.H) ? __expf(p[h] - mn) : 0.f;
                    s[h] = s[h] * sc[h] + uniform(group_sum<64>(p[h]));
                    m[h] = mn;
                }
                if (k0 > beg) {
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c) {
                        float f = sc[0];
                        if constexpr (CPH > 1 || HPC == 1) {
                            f = sc[(c / CPH) < HT ? (c / CPH) : 0];
                        } else {
#pragma unroll
                            for (int g = 0; g < HPC; ++g)
                                if (c * HPC + g < HT && seg == g) f = sc[c * HPC + g < HT ? c * HPC + g : 0];
                        }
#pragma unroll
                        for (int t = 0; t < VEC; ++t) acc[c][t] *= f;
                    }
                }
            } else {
#pragma unroll
                for (int h = 0; h < HT; ++h) p[h] = (valid && h < a.H) ? __expf(p[h] - m[h]) * s[h] : 0.f;
            }
        } else {
#pragma unroll
            for (int h = 0; h < HT; ++h) p[h] = valid ? 1.f : 0.f;  // plain (optionally ew-weighted) sum
        }
        if (a.ew) {
            const float w = valid ? a.ew[k] : 0.f;
#pragma unroll
            for (int h = 0; h < HT; ++h) p[h] *= w;
        }
        const int cnt = min(64, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float v[U][NCHUNK][VEC], ww[U][NSLOT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);  // past the end: re-read a valid neighbour with weight 0
                const int sidx = __builtin_amdgcn_readlane(idx, j);
                const float* px = a.x + (int64_t)sidx * a.ldx;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + xoff[c]);
#pragma unroll
                for (int q = 0; q < NSLOT; ++q) {
                    float w = 0.f;
                    if constexpr (CPH > 1 || HPC == 1) {
                        if (q < HT) w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[q < HT ? q : 0]), j));
                    } else {
#pragma unroll
                        for (int g = 0; g < HPC; ++g)
                            if (q * HPC + g < HT) {
                                const float wg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[q * HPC + g < HT ? q * HPC + g : 0]), j));
                                if (seg == g) w = wg;
                            }
                    }
                    ww[u][q] = i + u < cnt ? w : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                    for (int t = 0; t < VEC; ++t) acc[c][t] = fmaf(ww[u][c / CPH], v[u][c][t], acc[c][t]);
        }
    }
    // normalise (whole rows), epilogue, store
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        if (!act[c]) continue;
        if (slot < 0) {
            if (a.el) {
                float sv = s[0];
                if constexpr (CPH > 1 || HPC == 1) {
                    sv = s[(c / CPH) < HT ? (c / CPH) : 0];
                } else {
#pragma unroll
                    for (int g = 0; g < HPC; ++g)
                        if (c * HPC + g < HT && seg == g) sv = s[c * HPC + g < HT ? c * HPC + g : 0];
                }
                const float inv = sv > 0.f ? 1.f / sv : 0.f;
#pragma unroll
                for (int t = 0; t < VEC; ++t) acc[c][t] *= inv;
            }
            infer_epilogue<VEC>(a, row, hd[c], el[c], acc[c]);
            vstore<VEC>(a.out + (int64_t)row * a.ldo + (int64_t)hd[c] * a.hso + el[c], acc[c]);
#pragma unroll
            for (int t = 0; t < VEC; ++t) amax = fmaxf(amax, fabsf(acc[c][t]));
        } else {
            vstore<VEC>(a.partial + (int64_t)slot * a.ldp + (int64_t)hd[c] * a.D + el[c], acc[c]);
        }
    }
    if (a.absmax) absmax_publish(wave_absmax(amax), a.absmax);     // (long rows: gat_infer_combine_kernel)
}

// ---------------------------------------------------------------------------------------------------------------------
// head-major layout: one LANES-wide group per (item, head)
// ---------------------------------------------------------------------------------------------------------------------
template <int VEC, int LANES, int NCHUNK>
__global__ __launch_bounds__(kBlock) void gat_infer_heads_kernel(InferArgs a) {
    constexpr int U = 4;
    const float NEG_INF = -__builtin_inff();
    const int lane = threadIdx.x % LANES;
    const int64_t gid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
    if (gid >= a.n_items * a.H) return;  // whole groups leave together
    int head = (int)(gid / a.n_items);
    const int64_t item = gid - (int64_t)head * a.n_items;
    const int4 it = a.items[item];
    int row = it.x, beg = it.y, end = it.z, slot = it.w;
    if constexpr (LANES == 64) {
        row = __builtin_amdgcn_readfirstlane(row);
        beg = __builtin_amdgcn_readfirstlane(beg);
        end = __builtin_amdgcn_readfirstlane(end);
        slot = __builtin_amdgcn_readfirstlane(slot);
        head = __builtin_amdgcn_readfirstlane(head);
    }
    const float* xb = a.x + (int64_t)head * a.hsx;
    int off[NCHUNK];
    bool act[NCHUNK];
    float acc[NCHUNK][VEC];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        const int e = (c * LANES + lane) * VEC;
        act[c] = e < a.D;
        off[c] = act[c] ? e : 0;
#pragma unroll
        for (int t = 0; t < VEC; ++t) acc[c][t] = 0.f;
    }
    const float erv = a.er ? a.er[(int64_t)row * a.lder + head] : 0.f;
    float m = (slot >= 0 && a.el) ? a.rowstat[((int64_t)slot * a.H + head) * 2] : NEG_INF;
    float s = (slot >= 0 && a.el) ? a.rowstat[((int64_t)slot * a.H + head) * 2 + 1] : 0.f;
    for (int k0 = beg; k0 < end; k0 += LANES) {
        const int k = k0 + lane;
        const bool valid = k < end;
        const int idx = valid ? a.indices[k] : 0;
        float p = valid ? 1.f : 0.f;
        if (a.el) {
            float z = NEG_INF;
            if (valid) {
                z = a.el[(int64_t)idx * a.ldel + head] + erv;
                if (a.ee) z += a.ee[(int64_t)k * a.H + head];
                z = z > 0.f ? z : z * a.slope;
            }
            if (slot < 0) {
                const float mn = fmaxf(m, group_max<LANES>(z));
                const float sc = __expf(m - mn);
                p = valid ? __expf(z - mn) : 0.f;
                s = s * sc + group_sum<LANES>(p);
                m = mn;
                if (k0 > beg) {
#pragma unroll
                    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                        for (int t = 0; t < VEC; ++t) acc[c][t] *= sc;
                }
            } else {
                p = valid ? __expf(z - m) * s : 0.f;
            }
        }
        if (a.ew) p *= valid ? a.ew[k] : 0.f;
        const int cnt = min(LANES, end - k0);
        for (int i = 0; i < cnt; i += U) {
            float v[U][NCHUNK][VEC], ww[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(i + u, cnt - 1);
                const int sidx = group_bcast<LANES>(idx, j);
                ww[u] = i + u < cnt ? group_bcast<LANES>(p, j) : 0.f;
                const float* px = xb + (int64_t)sidx * a.ldx;
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c) vload<VEC>(v[u][c], px + off[c]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
                    for (int t = 0; t < VEC; ++t) acc[c][t] = fmaf(ww[u], v[u][c][t], acc[c][t]);
        }
    }
    const float inv = (slot < 0 && a.el) ? (s > 0.f ? 1.f / s : 0.f) : 1.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        if (!act[c]) continue;
        if (slot < 0) {
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[c][t] *= inv;
            infer_epilogue<VEC>(a, row, head, off[c], acc[c]);
            vstore<VEC>(a.out + (int64_t)row * a.ldo + (int64_t)head * a.hso + off[c], acc[c]);
        } else {
            vstore<VEC>(a.partial + (int64_t)slot * a.ldp + (int64_t)head * a.D + off[c], acc[c]);
        }
    }
}

// (max, 1/sum) per head of every long row, written to each of the row's slots.  One workgroup per long row; heads in register
// tiles of 8.
__global__ __launch_bounds__(kBlock) void gat_infer_rowstat_kernel(InferArgs a) {
    __shared__ float lds[kBlock / 64][8];
    const float NEG_INF = -__builtin_inff();
    const int li = blockIdx.x;
    const int row = a.long_rows[li];
    const int beg = a.indptr[row], end = a.indptr[row + 1];
    const int wave = threadIdx.x >> 6;
    for (int h0 = 0; h0 < a.H; h0 += 8) {
        float m[8], s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = NEG_INF, s[j] = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = beg + (int)threadIdx.x; k < end; k += kBlock) {
                const int idx = a.indices[k];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (h0 + j >= a.H) continue;
                    float z = a.el[(int64_t)idx * a.ldel + h0 + j] + (a.er ? a.er[(int64_t)row * a.lder + h0 + j] : 0.f);
                    if (a.ee) z += a.ee[(int64_t)k * a.H + h0 + j];
                    z = z > 0.f ? z : z * a.slope;
                    if (pass == 0) m[j] = fmaxf(m[j], z);
                    else s[j] += __expf(z - m[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = pass == 0 ? group_max<64>(m[j]) : group_sum<64>(s[j]);
                __syncthreads();
                if ((threadIdx.x & 63) == 0) lds[wave][j] = v;
                __syncthreads();
                float r = lds[0][j];
#pragma unroll
                for (int w = 1; w < kBlock / 64; ++w) r = pass == 0 ? fmaxf(r, lds[w][j]) : r + lds[w][j];
                if (pass == 0) m[j] = r;
                else s[j] = r;
            }
        }
        if (threadIdx.x == 0) {
            for (int sl = a.long_ptr[li]; sl < a.long_ptr[li + 1]; ++sl)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (h0 + j < a.H) {
                        a.rowstat[((int64_t)sl * a.H + h0 + j) * 2] = m[j];
                        a.rowstat[((int64_t)sl * a.H + h0 + j) * 2 + 1] = s[j] > 0.f ? 1.f / s[j] : 0.f;
                    }
        }
    }
}

// out[row,h,d] = epilogue(partial[first slot] + ... + partial[last slot]), in slot order.
__global__ __launch_bounds__(kBlock) void gat_infer_combine_kernel(InferArgs a, int64_t n_long) {
    const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t hd = (int64_t)a.H * a.D;
    float acc[1] = {0.f};
    if (gid < n_long * hd) {
        const int64_t i = gid / hd;
        const int e = (int)(gid - i * hd);
        const int h = e / a.D, d = e - h * a.D;
        int p = a.long_ptr[i];
        const int p1 = a.long_ptr[i + 1];
        for (; p + 4 <= p1; p += 4) {       // four loads in flight, added in slot order
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a.partial[(int64_t)(p + j) * a.ldp + e];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0] += v[j];
        }
        for (; p < p1; ++p) acc[0] += a.partial[(int64_t)p * a.ldp + e];
        const int row = a.long_rows[i];
        infer_epilogue<1>(a, row, h, d, acc);
        a.out[(int64_t)row * a.ldo + (int64_t)h * a.hso + d] = acc[0];
    }
    if (a.absmax) absmax_publish(wave_absmax(fabsf(acc[0])), a.absmax);    // every lane arrives here
}

template <int VEC, int HL, int NCHUNK, int CPH>
static void launch_infer_rows(const InferArgs& a, hipStream_t st) {
    const int64_t blocks = (a.n_items * 64 + kBlock - 1) / kBlock;
    if (blocks == 0) return;
    set_kernel("bot::gat_infer_rows_kernel<%d,%d,%d,%d>", VEC, HL, NCHUNK, CPH);
    hipLaunchKernelGGL((gat_infer_rows_kernel<VEC, HL, NCHUNK, CPH>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

// Same shape rules as dispatch_spmm_rows (spmm.hip); additionally H <= 8 (softmax state per head in registers).
template <int VEC>
static bool dispatch_infer_rows(const InferArgs& a, hipStream_t st) {
    if (a.H < 2 || a.H > 8) return false;
    const int L = (a.D + VEC - 1) / VEC;
    if (L <= 8) return false;
    if (L > 64) {
        if (L > 128 || VEC == 4 || a.H > 3) return false;
        if constexpr (VEC != 4) {
            if (a.H == 2) launch_infer_rows<VEC, 64, 4, 2>(a, st);
            else launch_infer_rows<VEC, 64, 6, 2>(a, st);
        }
        return true;
    }
    const int HL = L <= 16 ? 16 : (L <= 32 ? 32 : 64);
    const int nchunk = (a.H * HL + 63) / 64;
    if (nchunk > 4) return false;
#define BOT_ROWS(HLV)                                                   \
    do {                                                                \
        if (nchunk == 1) launch_infer_rows<VEC, HLV, 1, 1>(a, st);      \
        else if (nchunk == 2) launch_infer_rows<VEC, HLV, 2, 1>(a, st); \
        else if (nchunk == 3) launch_infer_rows<VEC, HLV, 3, 1>(a, st); \
        else launch_infer_rows<VEC, HLV, 4, 1>(a, st);                  \
    } while (0)
    if (HL == 16) BOT_ROWS(16);
    else if (HL == 32) BOT_ROWS(32);
    else BOT_ROWS(64);
#undef BOT_ROWS
    return true;
}

template <int VEC, int LANES, int NCHUNK>
static void launch_infer_heads(const InferArgs& a, hipStream_t st) {
    const int64_t groups = a.n_items * a.H;
    const int64_t blocks = (groups * LANES + kBlock - 1) / kBlock;
    if (blocks == 0) return;
    set_kernel("bot::gat_infer_heads_kernel<%d,%d,%d>", VEC, LANES, NCHUNK);
    hipLaunchKernelGGL((gat_infer_heads_kernel<VEC, LANES, NCHUNK>), dim3((unsigned)blocks), dim3(kBlock), 0, st, a);
}

template <int VEC>
static void dispatch_infer_heads(const InferArgs& a, hipStream_t st) {
    const int L = (a.D + VEC - 1) / VEC;
    if (L <= 8) launch_infer_heads<VEC, 8, 1>(a, st);
    else if (L <= 16) launch_infer_heads<VEC, 16, 1>(a, st);
    else if (L <= 32) launch_infer_heads<VEC, 32, 1>(a, st);
    else if (L <= 64) launch_infer_heads<VEC, 64, 1>(a, st);
    else if (L <= 128) launch_infer_heads<VEC, 64, 2>(a, st);
    else if (L <= 192) launch_infer_heads<VEC, 64, 3>(a, st);
    else launch_infer_heads<VEC, 64, 4>(a, st);
}

}  // namespace bot

extern "C" {

int64_t bot_gat_infer_workspace_floats(int64_t n_slots, int32_t H, int32_t D) { return n_slots * (int64_t)H * (D + 2); }

int bot_gat_infer_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items,
                      int64_t n_items, const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, int64_t n_slots,
                      const float* x, int64_t ldx, int64_t hsx, const float* el, int64_t ldel, const float* er, int64_t lder,
                      const float* ee, const float* ew, float slope, int32_t H, int32_t D, const float* addend, int64_t lda,
                      int64_t hsa, const float* scale, const float* shift, int32_t relu, float* out, int64_t ldo, int64_t hso,
                      float* workspace, uint32_t* absmax_slots, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(n_rows >= 0 && nnz >= 0 && n_items >= 0 && n_long >= 0 && n_slots >= 0, BOT_E_RANGE, "gat_infer: negative size");
    BOT_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, BOT_E_RANGE, "gat_infer: int32 index range exceeded");
    BOT_REQUIRE(H >= 1 && D >= 1, BOT_E_RANGE, "gat_infer: H=%d D=%d must be >= 1", H, D);
    if (n_rows == 0) return 0;
    BOT_REQUIRE(indptr && items && x && out, BOT_E_NULL, "gat_infer: indptr/items/x/out is NULL");
    BOT_REQUIRE(nnz == 0 || indices, BOT_E_NULL, "gat_infer: indices is NULL");
    BOT_REQUIRE(el || !(er || ee), BOT_E_NULL, "gat_infer: er / ee given without el (el == NULL means a plain weighted sum)");
    BOT_REQUIRE(n_long == 0 || (long_rows && long_ptr && workspace), BOT_E_NULL, "gat_infer: long rows need long_rows/long_ptr/workspace");
    BOT_REQUIRE(ldx >= (int64_t)(H - 1) * hsx + D && ldo >= (int64_t)(H - 1) * hso + D && hsx >= D && hso >= D, BOT_E_RANGE,
                "gat_infer: strides smaller than the slab (ldx=%lld hsx=%lld ldo=%lld hso=%lld H=%d D=%d)", (long long)ldx,
                (long long)hsx, (long long)ldo, (long long)hso, H, D);
    BOT_REQUIRE(el == nullptr || ldel >= H, BOT_E_RANGE, "gat_infer: ldel=%lld < H", (long long)ldel);
    BOT_REQUIRE(er == nullptr || lder >= H, BOT_E_RANGE, "gat_infer: lder=%lld < H", (long long)lder);
    BOT_REQUIRE(addend == nullptr || (hsa >= D && lda >= (int64_t)(H - 1) * hsa + D), BOT_E_RANGE, "gat_infer: addend strides smaller than the slab");
    BOT_REQUIRE(aligned(x, 4) && aligned(out, 4) && aligned(items, 16), BOT_E_ALIGN, "gat_infer: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    float* rowstat = workspace;                                   // [n_slots][H][2]
    float* partial = workspace ? workspace + n_slots * (int64_t)H * 2 : nullptr;  // [n_slots][H*D]
    InferArgs a{indptr, indices, reinterpret_cast<const int4*>(items), n_items, x, ldx, hsx, el, ldel, er, lder, ee, ew, slope, H, D,
                addend, lda, hsa, scale, shift, relu, out, ldo, hso, rowstat, partial, (int64_t)H * D, long_rows, long_ptr};
    // every operand a lane touches with vector accesses: x, out, addend, the partials (offset 2*n_slots*H floats: even) and the
    // per-column scale / shift (column head*D + e)
    const int vec = pick_vec(D, {ldx, hsx, ldo, hso, addend ? lda : 0, addend ? hsa : 0, (n_slots * (int64_t)H * 2) % 4},
                             {x, out, addend, scale, shift, workspace});
    BOT_REQUIRE(D <= vec * 256, BOT_E_RANGE, "gat_infer: D=%d exceeds the %d floats one launch tile covers", D, vec * 256);
    if (n_long > 0 && el) {
        hipLaunchKernelGGL(gat_infer_rowstat_kernel, dim3((unsigned)n_long), dim3(kBlock), 0, st, a);
        if (int rc = hip_status("gat_infer rowstat launch")) return rc;
    }
    a.absmax = absmax_slots;            // by-product of the all-heads kernel and of the long rows' combine pass
    const bool rows = vec == 4 ? dispatch_infer_rows<4>(a, st) : (vec == 2 ? dispatch_infer_rows<2>(a, st) : dispatch_infer_rows<1>(a, st));
    if (!rows) {
        a.absmax = nullptr;             // the head-major kernel has no by-product form: a pass over the result below
        if (vec == 4) dispatch_infer_heads<4>(a, st);
        else if (vec == 2) dispatch_infer_heads<2>(a, st);
        else dispatch_infer_heads<1>(a, st);
    }
    if (int rc = hip_status("gat_infer launch")) return rc;
    if (n_long > 0) {
        const int64_t n = n_long * H * D;
        hipLaunchKernelGGL(gat_infer_combine_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a, n_long);
        if (int rc = hip_status("gat_infer combine launch")) return rc;
    }
    if (absmax_slots && !rows) {
        for (int h = 0; h < H; ++h) launch_absmax_slots(out + (int64_t)h * hso, ldo, n_rows, D, absmax_slots, st);
        if (int rc = hip_status("gat_infer absmax launch")) return rc;
    }
    return 0;
}

}  // extern "C"
