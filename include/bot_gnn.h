/*
 * bot_gnn.h — C ABI of libbot_gnn.so: gfx950 (MI355X) message-passing kernels for full-batch
 * GAT / GCN forward + backward.
 *
 * This is the drop-in boundary.  Each entry point replaces one `dgl 0.5.*` operator that the
 * reference (AiRyunn/BoT) invokes from its layer code; the citation next to each function is the
 * reference call site (paths relative to the reference repository root).  The reference's own FFI
 * for this path is DGL's Python->C++ operator registry (`dgl.ops.gspmm / gsddmm / edge_softmax`,
 * reached through `graph.update_all`, `graph.apply_edges`, `dgl.ops.edge_softmax`); the host-side
 * binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless the name ends in `_host`.  The caller allocates and
 *    owns every buffer (inputs, outputs, workspace); the library never allocates, frees, copies,
 *    synchronises or keeps a pointer past the call — every launch function is hipGraph-capturable.
 *  - Work is enqueued on `stream` (a hipStream_t passed as void*) and NOT synchronised.
 *  - Return value: 0 = ok, < 0 = invalid argument (BOT_E_*), > 0 = a hipError_t from the launch.
 *    `bot_last_error()` returns a thread-local description of the last non-zero return.
 *  - A graph direction is given in compressed-row form: `indptr[n_rows+1]`, `indices[nnz]`.
 *    For the forward pass rows are DESTINATION nodes and indices are the SOURCE of every in-edge
 *    (CSC of the adjacency); for the transposed pass rows are sources and indices destinations.
 *    Within a row, positions are in ascending edge-id order.  Index type is int32.
 *  - "position order" = the order of `indices`; `perm[k]` maps position k to a row of an
 *    edge-indexed array (the edge id, or the position in the other direction).  NULL = identity.
 *  - Node features are fp32, laid out [n, H, D] with explicit strides in floats: `ld*` between
 *    nodes, `hs*` between heads (so padded layouts are allowed).  Edge arrays are [nnz, H] dense.
 *  - No float atomics anywhere: identical inputs give bitwise identical outputs run to run.
 */
#ifndef BOT_GNN_H
#define BOT_GNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BOT_ABI_VERSION 19

#define BOT_E_NULL (-1)     /* required pointer is NULL                 */
#define BOT_E_RANGE (-2)    /* size / stride / head count out of range  */
#define BOT_E_ALIGN (-3)    /* pointer not 4-byte aligned               */
#define BOT_E_PLAN (-4)     /* inconsistent row plan                    */

typedef void* bot_stream_t; /* hipStream_t */

int bot_abi_version(void);
const char* bot_last_error(void);
/* v17: a HIP stream of the library's own (hipStreamNonBlocking; high_priority != 0: the device's HIGHEST priority, 0: its LOWEST - not the
 * default priority torch's streams have: the side stream's workgroups are dispatched behind the main stream's when both have some ready,
 * which is what the step wants of work only the optimizer waits for, and what every A/B of bot_amd.side was measured with), never destroyed -
 * the second stream of bot_amd.side (weight-gradient products beside the sparse backward).  Not one of PyTorch's 32 pooled streams, which
 * are handed out round-robin and would sooner or later alias a capture stream or a process group's collective stream. */
int bot_stream_create(int32_t high_priority, bot_stream_t* out);
/* Name (as a profiler prints it) of the main device kernel the calling thread's most recent SpMM-family launch function
 * dispatched — the template instance depends on H, D and the operands' alignment.  Diagnostic only (bench.py names the
 * kernel of its roofline line from it); thread-local like bot_last_error. */
const char* bot_last_kernel(void);
/* v19, diagnostic only: from now on a SIGABRT / std::terminate in this PROCESS first appends the native frames of the thread that raised it
 * (and the uncaught exception's type and what()) to the file `path_host`, then runs the handler that was installed before (Python's
 * faulthandler) and ends the process as it would have ended.  abort() is raised on the calling thread, so the trace names the caller
 * (HIP runtime, RCCL watchdog, libstdc++).  tests/conftest.py arms it for every GPU test process; the product never calls it. */
int bot_debug_abort_trace(const char* path_host);

/* ---------------------------------------------------------------------------------------------
 * Row plan (host side, pure integer work; built once per graph direction).
 *
 * Splits rows longer than `chunk` neighbours into chunks so that no wavefront owns more than
 * `chunk` gathers (power-law tails), and lists those long rows.  A work item is 4 x int32:
 * {row, begin, end, slot}; slot < 0: the item owns the whole row and writes the result directly;
 * slot >= 0: the item writes a partial sum into workspace slot `slot`, and the long row's partials
 * [long_ptr[i], long_ptr[i+1]) are added in slot order afterwards (deterministic).
 * Items are emitted longest-first so heavy items start early.
 * ------------------------------------------------------------------------------------------- */
int bot_row_plan_size_host(const int32_t* indptr_host, int64_t n_rows, int32_t chunk,
                           int64_t* n_items, int64_t* n_long, int64_t* n_slots);
int bot_row_plan_fill_host(const int32_t* indptr_host, int64_t n_rows, int32_t chunk,
                           int32_t* items_host /* [n_items*4] */, int32_t* long_rows_host /* [n_long] */,
                           int32_t* long_ptr_host /* [n_long+1] */);
/* chunk size recommended for a graph with nnz edges (a quarter of one wavefront's share). */
int32_t bot_row_plan_default_chunk(int64_t nnz);

/* ---------------------------------------------------------------------------------------------
 * Degrees.  Replaces graph.in_degrees() / graph.out_degrees()
 * (src/no-sampling/models.py:335,352,388,478,501,551; src/ogbn-proteins/gat.py:64).
 * deg[r] = indptr[r+1] - indptr[r], int64, bit-exact.
 * ------------------------------------------------------------------------------------------- */
int bot_degrees_i64(const int32_t* indptr, int64_t n_rows, int64_t* deg, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SpMM.  Replaces update_all(fn.copy_src/copy_u, fn.sum)   — copy_u_sum  (models.py:374,381)
 *        and      update_all(fn.u_mul_e, fn.sum)            — u_mul_e_sum (models.py:547,
 *        src/ogbn-proteins/models.py:146, src/ogbn-products/models.py:147), and serves as their
 *        backward on the transposed direction.
 *
 *   out[r,h,:] = sum_{k in row r} w[wperm[k],h] * x[indices[k],h,:]  (+ addend[r,h,:])     (w == NULL: weight 1)
 *
 * `addend` (may be NULL; strides lda/hsa) is the fused epilogue for the layer's residual branch
 * `rst = rst + res_fc(h)` (models.py:558-560).  `partial` is caller-provided workspace of bot_spmm_workspace_floats(...) floats (may be NULL when
 * the plan has no long rows).
 * ------------------------------------------------------------------------------------------- */
int64_t bot_spmm_workspace_floats(int64_t n_slots, int32_t H, int32_t D);
/* Layout hint for the calling thread's following bot_spmm_f32 calls on weighted multi-head slabs whose head width is not a multiple
 * of 4 floats (3 x 250) and whose rows are H*D contiguous floats on a 16-byte aligned pitch >= H*D rounded up to x4: 0 (default) =
 * head-segment lanes with 8-byte loads, 1 = flat 16-byte lanes (spmm_flat_kernel).  Results are bitwise identical; flat is faster when
 * the gathered rows are L2-resident (graphs numbered for locality), slightly slower when they are fabric-bound.  Speed only. */
int bot_spmm_set_layout(int32_t layout);
int bot_spmm_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                 const int32_t* items, int64_t n_items,
                 const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long,
                 const float* x, int64_t ldx, int64_t hsx,
                 const float* w, const int32_t* wperm,
                 int32_t H, int32_t D,
                 float* out, int64_t ldo, int64_t hso,
                 const float* addend, int64_t lda, int64_t hsa,
                 float* partial, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * L2-blocked SpMM for dense graphs (mean degree in the hundreds).  Same result as bot_spmm_f32 (w in position order, no
 * addend) for the destination rows listed in `tile_rows`; other rows of `out` are not touched (the caller runs hub rows
 * through bot_spmm_f32 with a plan restricted to them).  The blocked edge structure is built once per graph direction
 * by the host (bot_amd/blocked.py): destination rows grouped into tiles of T (32/64/128/256) rows, sources cut into `nblk`
 * column blocks of `block_rows` (a power of two) rows, edges sorted by (tile, wave = slot % 16, block, slot, position):
 *   tile_rows[n_tiles*T]      destination row of each tile slot (-1: padding)
 *   ptr[n_tiles*16 + 1]       offsets of each wave's edge stream, (tile, wave)-major; a stream is sorted by (block, slot, position)
 *   b_src / b_lrow / b_pos    per blocked edge: source row, tile slot of its destination, position in the unblocked order
 * Rows of at most 32 (16) vector lanes are gathered `epi` = 2 (4) edges per instruction, one per 32- (16-) lane group; the
 * host then pads every (slot, block) run of a stream to a multiple of `epi` entries with b_src = -1 (no edge), streams
 * start at multiples of `epi`, and T may be 256.  epi = 1 otherwise.
 * One 1024-thread workgroup owns a tile and keeps its T output rows in LDS; its 16 waves cross the column blocks in
 * lockstep and all workgroups walk the blocks in the same order, so the block being gathered is resident in the XCD's L2; tiles are launched `round_tiles` at a time (one resident wave
 * of workgroups per launch).  H*D <= 1024 floats, T*H*D*4 <= 160 KB.  Rows are [H*D] contiguous with row strides ldx / ldo
 * (/ lda); `addend` (may be NULL) is the same residual epilogue as in bot_spmm_f32.  Deterministic, no atomics.
 * ------------------------------------------------------------------------------------------- */
int bot_spmm_blocked_f32(const int32_t* tile_rows, const int32_t* ptr, const int32_t* b_src, const uint8_t* b_lrow,
                         const int32_t* b_pos, int32_t n_tiles, int32_t nblk, int32_t block_rows, int32_t T, int32_t epi,
                         int32_t round_tiles, const float* x, int64_t ldx, const float* w, int32_t H, int32_t D, float* out,
                         int64_t ldo, const float* addend, int64_t lda, bot_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Fused backward of u_mul_e_sum (models.py:547) in ONE sweep over the transposed direction (rows = sources u,
 * indices = destinations v): every gathered row x[v,h,:] (the upstream gradient) is used twice,
 *
 *   out[u,h,:]          = sum_{k in row u} w[wperm[k],h] * x[indices[k],h,:]      (gradient of the node features)
 *   dot_out[wperm[k],h] = < y[u,h,:] , x[indices[k],h,:] >                        (gradient of the edge weights)
 *
 * so a GAT layer's backward needs one E*H*D gather instead of the two of bot_spmm_f32 + bot_sddmm_dot_f32.
 * y is the layer's forward input slab (row-local).  D <= 1024 (16-byte aligned slabs) / 512 / 256.
 * absmax_slots (optional, NULL = off): max|out| is folded into these bot_absmax_slots() words as a by-product (see
 * "Maxima as by-products" below) — `out` then needs no separate pass before it becomes a halves-GEMM operand.
 * ------------------------------------------------------------------------------------------- */
int bot_spmm_dot_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                     const int32_t* items, int64_t n_items,
                     const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long,
                     const float* x, int64_t ldx, int64_t hsx,
                     const float* w, const int32_t* wperm,
                     const float* y, int64_t ldy, int64_t hsy,
                     int32_t H, int32_t D,
                     float* out, int64_t ldo, int64_t hso,
                     float* dot_out, float* partial, uint32_t* absmax_slots, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Aggregate-before-project forms of u_mul_e_sum (models.py:547) for layers whose input is narrower than H*D — the
 * reordering GraphConv already applies for in_feats <= out_feats (models.py:377-385), valid for GAT because the
 * aggregation is linear:  sum_e a[e,h] (W_h x[u]) = W_h (sum_e a[e,h] x[u]).  The source row has no head axis and is
 * gathered once per edge for all H <= 4 heads (D <= 1024):
 *
 *   bot_spmm_bcast_f32       out[r,h,:] = sum_{k in row r} w[wperm[k],h] * x[indices[k],:]            x: [n_src, D]
 *   bot_spmm_dot_bcast_f32   out[r,:]   = sum_{k in row r} sum_h w[wperm[k],h] * x[indices[k],h,:]    x: [n, H, D] (ldx, hsx)
 *                            dot_out[wperm[k],h] = < y[r,:] , x[indices[k],h,:] >                     y: [n_rows, D]
 *
 * The second is the backward of the first on the transposed direction (x = gradient of the aggregated slab, y = the
 * layer input).  `out` of the first may be head-outer ([H, n, D]: hso = n*D, ldo = D) — the layout batched GEMMs want.
 * ------------------------------------------------------------------------------------------- */
int bot_spmm_bcast_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                       const int32_t* items, int64_t n_items,
                       const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long,
                       const float* x, int64_t ldx, const float* w, const int32_t* wperm,
                       int32_t H, int32_t D, float* out, int64_t ldo, int64_t hso,
                       float* partial, bot_stream_t stream);
/* v16: bot_spmm_bcast_f32 with the result written as a halves GEMM operand instead of fp32 (the aggregated slab's only consumers are the
 * per-head projection and its weight gradient, csrc/halves3.hip grouped forms): hout[r, h hsh + e] = h1, hout[r, h hsh + e + h2_off] =
 * 2^11 h2 of hscale[0] * out[r,h,e]  (halves_split's order 2), e < D; zeros for D <= e < hpiece.  hpiece, hsh, h2_off, ldh multiples of 4;
 * h2_off >= (H - 1) hsh + hpiece.  hscale: a device (s, 1/s) pair that bounds the result (sum_e w <= 1 per row: the scale of x does). */
int bot_spmm_bcast_halves_f16(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                              const int32_t* items, int64_t n_items,
                              const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long,
                              const float* x, int64_t ldx, const float* w, const int32_t* wperm,
                              int32_t H, int32_t D, const float* hscale, uint16_t* hout, int64_t ldh, int64_t hsh, int32_t h2_off,
                              int32_t hpiece, float* partial, bot_stream_t stream);
int bot_spmm_dot_bcast_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                           const int32_t* items, int64_t n_items,
                           const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long,
                           const float* x, int64_t ldx, int64_t hsx, const float* w, const int32_t* wperm,
                           const float* y, int64_t ldy, int32_t H, int32_t D,
                           float* out, int64_t ldo, float* dot_out, float* partial, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SDDMM dot.  The backward of u_mul_e_sum with respect to the edge weights (models.py:547):
 *
 *   out[operm[k], h] = < x[indices[k],h,:] , y[r,h,:] >        for every position k of every row r
 *
 * `accumulate` != 0 adds into `out` (used to tile D > the per-launch limit of 1024 floats).
 * ------------------------------------------------------------------------------------------- */
int bot_sddmm_dot_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                      const int32_t* items, int64_t n_items,
                      const float* x, int64_t ldx, int64_t hsx,
                      const float* y, int64_t ldy, int64_t hsy,
                      int32_t H, int32_t D,
                      float* out, const int32_t* operm, int32_t accumulate, bot_stream_t stream);

/* The same gradient for the aggregate-before-project layer (bot_spmm_bcast_f32) when its input needs no gradient (the first
 * layer of a stack): the source row has no head axis and is gathered once per edge for all H <= 4 heads,
 *
 *   out[operm[k], h] = < x[indices[k],:] , y[r,h,:] >          x: [n_src, D] (ldx);  y: [n_rows, H, D] (ldy, hsy; may be head-outer)
 *
 * 4*D gathered bytes per edge instead of the 4*H*D of bot_spmm_dot_bcast_f32, and no transposed sweep.  D <= 1024 / 512 / 256. */
int bot_sddmm_dot_bcast_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                            const int32_t* items, int64_t n_items,
                            const float* x, int64_t ldx,
                            const float* y, int64_t ldy, int64_t hsy,
                            int32_t H, int32_t D,
                            float* out, const int32_t* operm, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SDDMM copy_u / u_add_v.  Replaces apply_edges(fn.copy_u) / apply_edges(fn.u_add_v)
 * (models.py:525 / :523; proteins models.py:127 / :125) on the COO list in edge-id order:
 *
 *   out[e, :] = x[src[e], :] (+ y[dst[e], :] when y != NULL)          width W floats per node
 * ------------------------------------------------------------------------------------------- */
int bot_sddmm_u_add_v_f32(const int32_t* src, const int32_t* dst, int64_t n_edges,
                          const float* x, const float* y, int32_t W, float* out, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Attention logits + leaky-ReLU + per-destination softmax in one sweep (models.py:517-544):
 *
 *   z[k,h] = el[indices[k],h] (+ er[r,h]) (+ ee[eperm[k],h])        el, er: [n,H]; ee: [nnz,H]
 *   a[k,h] = softmax over the positions k of row r of leaky_relu(z[k,h], slope)
 *
 * `keep` (uint8, indexed through eperm like ee; may be NULL) restates the edge-drop branch
 * models.py:528-539: positions with keep == 0 are left out of the softmax and get a = 0.
 * With el == er == NULL, slope == 1 this is dgl.ops.edge_softmax(graph, ee) (models.py:544) on
 * logits given in edge-id order.  `a` is written at aperm[k] (NULL: position order).
 * `long_rows` (rows longer than the plan's chunk) are handled by one workgroup each.
 * `zsign` (may be NULL; H <= 8): uint8 [nnz], addressed like `a`; bit h receives [z > 0], so that the backward can take the
 * leaky-ReLU derivative from one byte per edge instead of re-gathering el[src] and re-reading ee.
 * `attn_drop` > 0 with `a_drop` [nnz,H] (addressed like `a`): the dropout behind the softmax (`self.attn_drop(...)`, models.py:544)
 * fused in — a_drop = a * keep / (1 - attn_drop), keep from a Philox4x32-10 stream keyed by (drop_seed [+ *seed_offset *
 * 0x9E3779B97F4A7C15], record index, head / 4); `a` itself stays the plain softmax (the backward needs it).  The backward
 * takes the gradient of a_drop in `da` with the same (attn_drop, drop_seed, seed_offset) and regenerates the mask.
 * ------------------------------------------------------------------------------------------- */
int bot_gat_attn_fwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                         const int32_t* long_rows, int64_t n_long, int32_t chunk,
                         const float* el, const float* er, const float* ee, const int32_t* eperm,
                         const uint8_t* keep, float slope, int32_t H,
                         float* a, const int32_t* aperm, uint8_t* zsign,
                         float attn_drop, uint64_t drop_seed, const uint64_t* seed_offset, float* a_drop,
                         bot_stream_t stream);

/* Backward of the above.  Given a (forward output) and da (gradient w.r.t. a), both addressed
 * through aperm like the forward output:
 *
 *   t[r,h]   = sum_k a*da ;  de = a*(da - t) ;  dz = de * (z > 0 ? 1 : slope)
 *   dz[k,h]  written at zperm[k] (NULL: position order) — it is the gradient of ee, and the edge
 *            values whose per-source sum is the gradient of el (bot_segment_sum_f32 on the other
 *            direction);
 *   der[r,h] = sum_k dz[k,h]       (written when der != NULL)
 *
 * With el == er == NULL, slope == 1 this is the backward of dgl.ops.edge_softmax.  `zsign` (may be NULL): the sign bits the
 * forward stored; when given, el / er / ee / eperm / indices are not read.
 */
int bot_gat_attn_bwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                         const int32_t* long_rows, int64_t n_long, int32_t chunk,
                         const float* el, const float* er, const float* ee, const int32_t* eperm,
                         float slope, int32_t H,
                         const float* a, const float* da, const int32_t* aperm,
                         float* dz, const int32_t* zperm, float* der, const uint8_t* zsign,
                         float attn_drop, uint64_t drop_seed, const uint64_t* seed_offset, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Inference-only GAT layer: what `evaluate()` (src/no-sampling/run.py:290-322) needs from a layer — logits,
 * leaky-ReLU, per-destination softmax (models.py:517-544), aggregation (models.py:547), residual (models.py:558-560), then the
 * stack's eval-mode BatchNorm / bias and ReLU (models.py:726-734) — in ONE sweep over the in-edges, with nothing edge-sized
 * and no pre-BatchNorm [n,H*D] tensor written:
 *
 *   z[k,h]     = el[indices[k]*ldel + h] (+ er[r*lder + h]) (+ ee[k*H + h]);   e = leaky_relu(z, slope)
 *   a[k,h]     = softmax over the positions k of row r of e[k,h]
 *   out[r,h,:] = act( (sum_k a[k,h] * ew[k] * x[indices[k],h,:] + addend[r,h,:]) * scale[h*D+:] + shift[h*D+:] )
 *
 * el / er are node arrays with a row stride (they may be columns of the projection GEMM's output); ee [nnz,H] and ew [nnz]
 * are in position order; er, ee, ew, addend, scale, shift may each be NULL; relu != 0 applies max(.,0).  el == NULL (then er
 * and ee must be NULL) drops the softmax: out = act((sum_k ew[k] x[indices[k]] + addend) * scale + shift), the GraphConv
 * aggregation with its degree normalisation folded into ew (models.py:351-395).  Rows without in-edges give act(addend*scale
 * + shift).  Long rows (row plan) take a (max, 1/sum) pass first and are combined in slot order; `workspace` holds
 * bot_gat_infer_workspace_floats(n_slots, H, D) floats (may be NULL without long rows).  D <= 1024 / 512 / 256 by alignment.
 * Deterministic, no atomics.  Same value as bot_gat_attn_fwd_f32 + bot_spmm_f32 up to fp32 rounding of the online softmax.
 * ------------------------------------------------------------------------------------------- */
int64_t bot_gat_infer_workspace_floats(int64_t n_slots, int32_t H, int32_t D);
int bot_gat_infer_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz,
                      const int32_t* items, int64_t n_items,
                      const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, int64_t n_slots,
                      const float* x, int64_t ldx, int64_t hsx,
                      const float* el, int64_t ldel, const float* er, int64_t lder,
                      const float* ee, const float* ew, float slope, int32_t H, int32_t D,
                      const float* addend, int64_t lda, int64_t hsa,
                      const float* scale, const float* shift, int32_t relu,
                      float* out, int64_t ldo, int64_t hso,
                      float* workspace, uint32_t* absmax_slots /* optional: max|out|, "Maxima as by-products" */, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Segment sum of edge values.  Replaces update_all(fn.copy_e, fn.sum) — copy_e_sum
 * (src/ogbn-proteins/gat.py:58) — and serves the backward of copy_u / u_add_v (models.py:523,525):
 *
 *   out[r, :] = sum_{k in row r} vals[perm[k], :]                      width W floats per edge
 * ------------------------------------------------------------------------------------------- */
int bot_segment_sum_f32(const int32_t* indptr, int64_t n_rows, int64_t nnz,
                        const int32_t* long_rows, int64_t n_long, int32_t chunk,
                        const float* vals, const int32_t* perm, int32_t W,
                        float* out, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Row gather / scatter-add used by the 1-D vertex-partitioned mode to pack halo rows for the RCCL
 * all-to-all and to fold received halo gradients back into owned rows (no reference counterpart:
 * the reference is single-device).  `rows` must be sorted-unique for the add form.
 *   gather:      out[i,:] = x[rows[i],:]
 *   scatter_add: x[rows[i],:] += vals[i,:]
 * ------------------------------------------------------------------------------------------- */
int bot_gather_rows_f32(const float* x, int64_t ldx, const int32_t* rows, int64_t n_sel, int32_t F,
                        float* out, int64_t ldo, bot_stream_t stream);
int bot_scatter_add_rows_f32(float* x, int64_t ldx, const int32_t* rows, int64_t n_sel, int32_t F,
                             const float* vals, int64_t ldv, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Hidden-layer epilogue: BatchNorm over the node axis + ReLU + dropout, fused.
 * Replaces `h = self.norms[i](h); h = self.activation(h); h = self.dropout(h)`
 * (src/no-sampling/models.py:636-639 and :726-731; nn.BatchNorm1d over ALL nodes, :609 / :698).
 *
 *   colstats      mean[c] = mean_r x[r,c];  m2[c] = sum_r (x[r,c]-mean[c])^2     (two-stage, fixed order)
 *   bn_act_fwd    y = drop_p( relu?( (x-mean)*invstd*weight + bias ) )
 *   bwd_reduce    g = dy * keep/(1-p) * [bn > 0];  sum_g[c] = sum_r g;  sum_gx[c] = sum_r g*xhat
 *   bwd_apply     dx = weight*invstd*( g - sum_g/count - xhat*sum_gx/count )
 *                 (sum_g == sum_gx == NULL: statistics were constants (eval mode): dx = weight*invstd*g)
 *
 * The dropout mask is a counter-based Philox4x32-10 stream: element (r, c) takes word c % 4 of the block with
 * counter r * ceil(F/4) + c/4 under key `seed` — a function of (seed, r, c, F) only, independent of pointer
 * alignment, strides or the vector width a launch picks, so forward and backward regenerate the same mask from
 * `seed` for any operand layout; nothing is stored.  `seed_offset` (device pointer, may be NULL): one 64-bit word read by
 * the kernel and mixed into the key (seed + word * 0x9E3779B97F4A7C15): a captured hipGraph bakes `seed` in, the caller
 * bumps the word between replays.  p == 0 disables dropout.  `weight`/`bias` may be
 * NULL.  `workspace` holds bot_bn_workspace_floats(F) floats.  In the vertex-partitioned mode the caller
 * all-reduces (mean, m2, count) and (sum_g, sum_gx) between the two halves; `total_count` is the global
 * row count.  The grad of weight is sum_gx, of bias sum_g.
 * ------------------------------------------------------------------------------------------- */
int64_t bot_bn_workspace_floats(int32_t F);
int bot_colstats_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float* mean, float* m2, float* workspace,
                     bot_stream_t stream);
/* sum[c] = sum_r x[r,c], the same two-stage reduction without the pivot shift, finished in double: the bias gradient of the Linear /
 * GATConv bias over all N rows (src/no-sampling/models.py:548, src/ogbn-products/models.py:107, :262). */
int bot_colsum_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float* sum, float* workspace, bot_stream_t stream);
/* colstats + everything nn.BatchNorm1d does with them in training mode, in one call: mean, invstd = rsqrt(m2/n + eps),
 * running_mean / running_var (may both be NULL) moved by `momentum` towards the batch mean / unbiased batch variance,
 * *num_batches_tracked (may be NULL) += 1.  Single-GPU form; the partitioned mode all-reduces between the two halves. */
int bot_bn_stats_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float eps, float momentum, float* mean, float* invstd,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float* workspace,
                     bot_stream_t stream);
int bot_bn_act_fwd_f32(const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                       const float* weight, const float* bias, int32_t relu, float p, uint64_t seed,
                       const uint64_t* seed_offset, float* y, int64_t ldy, bot_stream_t stream);
/* The same two calls when the epilogue's output is the next layer's GEMM operand (fp16 halves, see bot_gemm_halves_f32 below):
 * bn_stats_halves also tracks the column extremes and derives hscale = (s, 1/s) from the bound
 * max_c (|weight_c| max(|max_c - mean_c|, |min_c - mean_c|) invstd_c + |bias_c|) / (1 - p) >= max |y| — no pass over y;
 * bn_act_fwd_halves writes y AND its halves [h1 | h1 | 2^11 h2] (pieces of `piece` = F rounded up to x64 columns, zero padded; v15:
 * `pieces` = 2 leaves the duplicate out, [h1 | 2^11 h2] with ldh >= 2 * piece — halves_split's order 2), so the
 * separate halves_scale / halves_split passes over y disappear.  Needs the 4-column form (even F, 8-byte aligned rows).
 * y == NULL (v12): ONLY the halves are written — for a hidden state whose one consumer is the next projection GEMM (the backward
 * of this pass reads x, never y), which saves the fp32 store of an [n, F] matrix nobody reads. */
int bot_bn_stats_halves_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float eps, float momentum, float* mean, float* invstd,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* weight,
                            const float* bias, float p, float* hscale, float* workspace, bot_stream_t stream);
int bot_bn_act_fwd_halves_f32(const float* x, int64_t ldx, int64_t n, int32_t F, const float* mean, const float* invstd,
                              const float* weight, const float* bias, int32_t relu, float p, uint64_t seed,
                              const uint64_t* seed_offset, float* y, int64_t ldy, const float* hscale, uint16_t* hout, int64_t ldh,
                              int32_t piece, int32_t pieces, bot_stream_t stream);   /* pieces: 3 = [h1 | h1 | 2^11 h2], 2 = [h1 | 2^11 h2] (v15) */
int bot_bn_act_bwd_reduce_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F,
                              const float* mean, const float* invstd, const float* weight, const float* bias,
                              int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, float* sum_g, float* sum_gx,
                              float* workspace, bot_stream_t stream);
int bot_bn_act_bwd_apply_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F,
                             const float* mean, const float* invstd, const float* weight, const float* bias,
                             int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, const float* sum_g,
                             const float* sum_gx, double total_count, float* dx, int64_t lddx, uint32_t* absmax_slots /* optional: max|dx|,
                             "Maxima as by-products" */, bot_stream_t stream);
/* v16: the BatchNorm backward WITHOUT a split pass behind it, for a dx whose only consumers are halves GEMMs (the aggregate-first GAT layer):
 *   bot_bn_act_bwd_reduce_max_f32   the reduce pass, which also leaves the column maxima of |g| and |xhat| in the workspace;
 *   bot_bn_bwd_bound_f32            max_c |w_c| invstd_c (max|g_c| + |sum_g_c| / n + max|xhat_c| |sum_gx_c| / n) >= max |dx| into by-product slots
 *                                   (sum_g / sum_gx: the FINAL sums, i.e. after a cross-rank reduction; NULL: eval statistics) — a scale
 *                                   for dx BEFORE dx exists (bot_halves_scale_from_slots_f32); a bound that is loose by a few binades costs
 *                                   nothing in this format (22 bits for entries down to 2^-28 of the scale);
 *   bot_bn_act_bwd_apply_halves_f32 the apply pass writing dx as a LEFT operand [h1 | 2^11 h2] (second half h2_off columns behind the first),
 *                                   columns in blocks of hD (a head; even, F % hD == 0) that start every hDP >= hD columns; the padding
 *                                   columns are NOT written (the caller zeroes them once); dx (fp32) may be NULL. */
int bot_bn_act_bwd_reduce_max_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F,
                                  const float* mean, const float* invstd, const float* weight, const float* bias,
                                  int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, float* sum_g, float* sum_gx,
                                  float* workspace, bot_stream_t stream);
int bot_bn_bwd_bound_f32(int32_t F, int64_t n, const float* workspace, const float* sum_g, const float* sum_gx, double total_count,
                         const float* weight, const float* invstd, uint32_t* absmax_slots, bot_stream_t stream);
int bot_bn_act_bwd_apply_halves_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t n, int32_t F,
                                    const float* mean, const float* invstd, const float* weight, const float* bias,
                                    int32_t relu, float p, uint64_t seed, const uint64_t* seed_offset, const float* sum_g,
                                    const float* sum_gx, double total_count, float* dx, int64_t lddx, const float* hscale, uint16_t* hout,
                                    int64_t ldh, int32_t h2_off, int32_t hD, int32_t hDP, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Merged projection weight of a GAT layer (host-side convenience of the fused layer node, not a DGL operator): the layer's
 * linear maps on its input — fc (models.py:490-492), res_fc (:558-560), and the attention scores folded through fc,
 * el = h . (W_h^T attn_l[h]) (:517), er likewise (:521) — become ONE GEMM against
 *     merged [K, P] = [ W_fc^T (with_fc) | W_res^T (Wres != NULL) | wl | wr (attn_r != NULL) | 0 ... ]
 * with wl[k,h] = sum_d W_fc[h*D+d, k] * attn_l[h*D+d].  W, Wres: [H*D, K] row-major; attn_l, attn_r: [H*D].  `block` >= H*D is the
 * column width of each of the two copied blocks (zero padded): H*D rounded up to x4 puts the residual block of the GEMM output /
 * gradient buffer on a 16-byte boundary (3 x 250: columns [0, 750) | 2 pad | [752, 1502) | 2 pad | scores).  The backward takes
 * d merged and returns the gradients of the four parameters (dWres / d_attn_r NULL when absent).
 * ------------------------------------------------------------------------------------------- */
int bot_merge_weight_fwd_f32(const float* W, const float* Wres, const float* attn_l, const float* attn_r, int32_t H, int32_t D,
                             int32_t K, int32_t P, int32_t with_fc, int32_t block, float* merged, bot_stream_t stream);
int bot_merge_weight_bwd_f32(const float* W, const float* attn_l, const float* attn_r, int32_t H, int32_t D, int32_t K,
                             int32_t P, int32_t with_fc, int32_t block, const float* d_merged, float* dW, float* dWres,
                             float* d_attn_l, float* d_attn_r, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Edge-feature attention term of the ogbn-proteins GAT, fused.  Replaces, per layer,
 *   efeat_emb = relu(edge_encoder[i](efeat))                 src/ogbn-proteins/models.py:244-248
 *   attn_edge = attn_edge_fc(efeat_emb)                      src/ogbn-proteins/models.py:130-131
 * and their autograd:   ee[e,:] = W2 . relu(W1 . ef[e,:] + b1),   ef [E,I], W1 [J,I], b1 [J], W2 [H,J].
 * Only the reference's shape I = 8, J = 16 (H <= 8) is implemented (BOT_E_RANGE otherwise; callers fall back to
 * library ops).  `ef` and `ee`/`dz` are in the SAME edge order (the layers use CSC position order).  The backward
 * returns the three weight gradients (the edge features are inputs, not parameters); it runs the K = E reductions
 * on the fp32 MFMA and sums per-wavefront tiles in a fixed order.  workspace: bot_edge_mlp_workspace_floats().
 * ------------------------------------------------------------------------------------------- */
int64_t bot_edge_mlp_workspace_floats(void);
int bot_edge_mlp_fwd_f32(const float* ef, int32_t I, const float* W1, const float* b1, int32_t J, const float* W2,
                         int32_t H, int64_t n_edges, float* ee, bot_stream_t stream);
int bot_edge_mlp_bwd_f32(const float* ef, int32_t I, const float* W1, const float* b1, int32_t J, const float* W2,
                         int32_t H, const float* dz, int64_t n_edges, float* dW1, float* db1, float* dW2,
                         float* workspace, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Training-time edge drop of the GAT layers.  Replaces
 *   perm = torch.randperm(graph.number_of_edges()); bound = int(E * edge_drop); eids = perm[bound:]
 *                                      src/no-sampling/models.py:528-532, src/ogbn-proteins/models.py:120-127
 * keep[e] = 1 for a uniformly random subset of exactly n_keep of the n edges, 0 for the others — the mask the
 * attention kernel takes (bot_gat_attn_fwd_f32 `keep`).  The subset is a pure function of (n, n_keep, seed): every
 * edge gets a 64-bit Philox4x32-10 key and the n - n_keep smallest keys are dropped, found by a radix select (no
 * sort, no n-sized temporary).  workspace: bot_random_keep_workspace_bytes() bytes, 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int64_t bot_random_keep_workspace_bytes(void);
int bot_random_keep_u8(int64_t n, int64_t n_keep, uint64_t seed, uint8_t* keep, void* workspace, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fp32 GEMMs on the fp16 matrix cores.  The dense projections of the layer (`self.fc`, `self.res_fc`, the folded attention
 * columns: src/no-sampling/models.py:490-492, :519-522, :553-557; their backward) are fp32 GEMMs of [N, 750] x [750, 1536]
 * size, MFMA-bound at gfx950's fp32 rate.  Each fp32 operand is written as two fp16 halves, x = (h1 + h2) / s with
 * h1 = fp16(s x), h2 = fp16(s x - h1), s a power of two chosen ON THE DEVICE from max|x| (halves_scale: scale[0] = s,
 * scale[1] = 1/s), and a product is evaluated as a1 b1 + a1 b2 + a2 b1 with fp32 accumulation: ONE fp16 GEMM over the
 * concatenated reduction axis.  h1 + h2 carries 22-23 of the 24 significand bits; against fp64 the result is as close as
 * hipBLASLt's fp32 GEMM (tools/exp_split_gemm*.py), at ~3x its speed.
 *
 *   halves_split  out[r, :] = [h1 | h1 | 2^11 h2] (order 0: left operands) or [h1 | h2 | 2^-11 h1] (order 1: right operands) or, v15,
 *                 [h1 | 2^11 h2] (order 2: a left operand WITHOUT the duplicate h1 piece, row pitch >= 2 * piece — only the library's
 *                 concatenated-axis GEMM reads the duplicate; bot_gemm_halves3_nt_f32 / _tn_f32 take the offset of the second half), every
 *                 piece `piece` >= F columns wide (zero padded; use a multiple of 64), out fp16 with row pitch ldo >= 3 * piece
 *                 (order 2: >= 2 * piece).
 *                 The 2^11 keeps a left operand's second half in fp16's normal range: ROWS down to 2^-28 of the matrix maximum keep
 *                 22 significant bits (one scale per matrix alone: 2^-17), and every entry is reproduced to 2^-38 of the matrix
 *                 maximum in absolute terms.  A product of two LEFT layouts (x^T d, the weight gradient) is formed from separate
 *                 GEMMs over the pieces with the 2^-11 applied by the caller (bot_amd/gemm.py:tn).
 *   gemm_halves   C[m,n] = alpha[j] * op(A)[m,k] op(B)[k,n] + beta * C[m,n], row-major, A / B fp16, C fp32, `alpha` a DEVICE vector of n
 *                 floats (one per output column; normally n copies of the product of the two operands' 1/s); trans_x != 0: the operand is stored transposed.  batch > 1: strided
 *                 batches (element strides).  Kernel choice: `algo_index` >= 0 = a solution index recorded for this shape on this library
 *                 build (bot_amd/tuning/halves_gemm.json); else with tune = 0 (what bot_amd passes by default) hipBLASLt's first heuristic
 *                 choice — no timing, no synchronisation, the same kernel every run.  EXCEPTION to the conventions at the top of this
 *                 header, opt-in only: tune = 1 / 2 makes the FIRST call per shape time the 16 heuristic candidates / all solutions on the
 *                 caller's buffers (beta == 0 only: C is overwritten; never under stream capture) — that call creates events,
 *                 synchronises, and the winner may differ from run to run (different accumulation order).  `workspace`: device
 *                 scratch for hipBLASLt (32 MiB is plenty).
 *   bot_gemm_halves_library_version: the hipBLASLt the library was COMPILED against (headers) and the one it is RUNNING on (the
 *                 first libhipblaslt.so.N the process loaded — with PyTorch in the process, the copy torch ships), both as
 *                 major * 100000 + minor * 100 + patch; runtime is 0 before the first gemm_halves call on the current device.
 * ------------------------------------------------------------------------------------------- */
int64_t bot_halves_workspace_floats(void);
int bot_halves_scale_f32(const float* x, int64_t ldx, int64_t n, int32_t F, float* scale, float* workspace, bot_stream_t stream);
/* Maxima as by-products (v12).  The scale of an operand only needs max|x|; when x is a gradient buffer that several kernels have just
 * written, they can deliver it: bot_spmm_dot_f32, bot_bn_act_bwd_apply_f32 and bot_gat_infer_f32 take `absmax_slots`, an array of bot_absmax_slots()
 * uint32 words that the CALLER ZEROES; every workgroup folds the largest |value| it stored into one of them as a bit pattern
 * (non-negative floats order like unsigned integers: an integer atomic max — exact, independent of the order of arrival; NaNs are
 * skipped as in halves_scale).  bot_absmax_slots_f32 adds a strided matrix the slow way (the few columns no producer covers);
 * bot_halves_scale_from_slots_f32 turns the slots into the same (s, 1/s) pair bot_halves_scale_f32 would have found. */
int32_t bot_absmax_slots(void);
int bot_absmax_slots_f32(const float* x, int64_t ldx, int64_t n, int32_t F, uint32_t* slots, bot_stream_t stream);
int bot_halves_scale_from_slots_f32(const uint32_t* slots, float* scale, bot_stream_t stream);
int bot_halves_split_f16(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, int32_t order, uint16_t* out,
                         int64_t ldo, int32_t piece, bot_stream_t stream);
/* The same split into a COLUMN RANGE of a wider halves operand: `out` points at the range's first column of piece 0, `width` (even,
 * F <= width <= piece) columns of each of the three pieces are written — F from x, zeros behind them — the pieces stay `piece` apart.
 * Lets several matrices that are one operand side by side (the gradients of a merged projection's column blocks) be split in place,
 * under ONE scale, without first copying them into one buffer. */
int bot_halves_split_cols_f16(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, int32_t order, uint16_t* out,
                              int64_t ldo, int32_t piece, int32_t width, bot_stream_t stream);
/* v16: x [n, H * D] as a LEFT operand without the duplicate piece whose head blocks are DP >= D columns wide: out[r, h DP + j] = h1,
 * out[r, h2_off + h DP + j] = 2^11 h2 of scale[0] * x[r, h D + j], zeros for D <= j < DP — H calls of bot_halves_split_cols_f16(order 2) on
 * column slices in one pass (the gradient operand of the aggregate-first GAT layer: every head's block starts on a 128-byte boundary). */
int bot_halves_split_heads_f16(const float* x, int64_t ldx, int64_t n, int32_t H, int32_t D, const float* scale, uint16_t* out, int64_t ldo,
                               int32_t h2_off, int32_t DP, bot_stream_t stream);
/* The weight gradient x^T d of two LEFT-layout operands, formed by the caller from row chunks (batched gemm_halves calls): a [chunks][K][2 PP]
 * = x1^T [d1 | 2^11 d2] and b [chunks][K][PP] = (2^11 x2)^T d1 per chunk, contiguous, plus optional remainder chunks rem_a [K][2 PP] / rem_b [K][PP]:
 *   out[k, p] = sum_c a[c][k][p] + (sum_c a[c][k][PP + p] + sum_c b[c][k][p]) * 2^-11      (chunk order, p < P <= PP)
 * one launch instead of two library reductions and four element-wise passes. */
int bot_halves_tn_combine_f32(const float* a, const float* b, int32_t chunks, int32_t K, int32_t PP, int32_t P, const float* rem_a,
                              const float* rem_b, float* out, int64_t ldo, bot_stream_t stream);
int bot_gemm_halves_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, const float* alpha, const uint16_t* A,
                        int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t batch, int64_t stride_a,
                        int64_t stride_b, int64_t stride_c, float beta, void* workspace, int64_t workspace_bytes, int32_t tune,
                        int32_t algo_index, bot_stream_t stream);
/* v14: the NT product of two halves operands, hand-written for gfx950 (csrc/halves3.hip) instead of the library call above:
 *   C[m, n] = scale_a[1] scale_b[1] * sum_k ( a1[m,k] b1[n,k] + a1[m,k] b2[n,k] + (2^11 a2)[m,k] (2^-11 b1)[n,k] )
 * scale_a / scale_b: the operands' (s, 1/s) device pairs from halves_scale (read by the kernel: no alpha vector, no launch to form it);
 * mode 0 (anything else is an ablation switch of the measurement tools).
 * A: a LEFT operand's buffer (row pitch lda halves; a1 at column 0, 2^11 a2 at column a2_off = 2 * piece), B: a RIGHT operand's buffer
 * (b1 at column 0, b2 at column b2_off = piece); k = the piece width (a multiple of 32).  Each operand half is staged through LDS once
 * per k-step and serves all three MFMAs (the library formulation concatenates the reduction axis three-fold and stages a1 and b1 twice).
 * Same operands, same three products, fp32 accumulation: results differ from bot_gemm_halves_f32 only in summation order.
 * Replaces the projections x W^T of src/no-sampling/models.py:490-492, :558-560 (forward) and their input gradients d W. */
int bot_gemm_halves3_nt_f32(int64_t m, int64_t n, int64_t k, const float* scale_a, const float* scale_b, const uint16_t* A, int64_t lda,
                            int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, float* C, int64_t ldc, int32_t mode,
                            bot_stream_t stream);
/* v14: the weight gradient  out[k, p] = scale_x[1] scale_d[1] * sum_n (x1 + x2)[n, k] (d1 + d2)[n, p]  (without the x2 d2 term) of two LEFT
 * operand buffers X [n_rows, ldx] (x1 at column 0, 2^11 x2 at column x2_off = 2 * piece) and D [n_rows, ldd], hand-written for gfx950
 * (csrc/halves3.hip: the reduction index runs along the ROWS of both operands; fragments by the transposing LDS read ds_read_b64_tr_b16;
 * 192 x 192 tiles, three LDS stages, split-K over row ranges with one split per XCD, partials added in split order).  kp / pp: the piece widths (multiples of 64, zero padded beyond k / p);
 * workspace: bot_gemm_halves3_tn_workspace_floats(n_rows, kp, pp) floats; mode 0 (other values: ablation switches of the measurement tools).  Replaces the batched library products + bot_halves_tn_combine_f32
 * for the weight gradients of `fc` / `res_fc` (backward of src/no-sampling/models.py:490-492, 558-560). */
int64_t bot_gemm_halves3_tn_workspace_floats(int64_t n_rows, int64_t kp, int64_t pp);
int bot_gemm_halves3_tn_f32(int64_t n_rows, int64_t k, int64_t p, int64_t kp, int64_t pp, const float* scale_x, const float* scale_d,
                            const uint16_t* X, int64_t ldx, int64_t x2_off, const uint16_t* D, int64_t ldd, int64_t d2_off, float* out,
                            int64_t ldo, float* workspace, int32_t mode, bot_stream_t stream);

/* v17: a hidden layer's gradient operand without a split pass (bot_amd/nn/fused.py:_GATHidden.backward; the backward of
 * src/no-sampling/models.py:490-492, :547, :558-560, :726-731).  The [N, P] gradient of the merged projection's output, [d ft | d res | d el | d er | 0],
 * used to be assembled in fp32 and split into halves by one more pass (1 GB read + 1 GB written at config 2).  Its two big column blocks
 * now leave their producers as halves - d res: bot_bn_act_bwd_apply_halves_f32 (fp32 dx beside it: the sweep gathers those rows), d ft:
 * bot_spmm_dot_halves_f16 - under a scale that must exist BEFORE they run, i.e. a BOUND:
 *     |d res| <= bound of bot_bn_bwd_bound_f32,    |d ft[u,h,:]| <= (sum of the edge weights out of u) max|d res| <= rowsum_bound * max|d res|
 * (bot_halves_scale_from_slots2_f32: the scale of `mult` x the slots' maximum; the format keeps 22 bits for entries down to 2^-28 of the
 * scale, a loose bound costs range, not accuracy).  The handful of attention columns, known only after the sweep and of another magnitude,
 * get a SECOND scale (bot_halves_tail_f16 writes them and the zero padding): the NT product multiplies its accumulators by the ratio of the
 * two scales (powers of two: exact) in front of the k-step where the second range starts, the TN product applies it per output column.
 *   bot_halves_scale_from_slots2_f32   scale = halves_scale(mult * max(slots)); cap_scale != NULL: at most cap_scale[0] * cap_ratio
 *   bot_halves_tail_f16                segments g < n_seg <= 8 of (col, width) = seg_cols[2 g], seg_cols[2 g + 1]: out[r, col + j] = h1,
 *                                      out[r, h2_off + col + j] = 2^11 h2 of scale[0] * srcs[g][r * src_ld[g] + j]  (srcs[g] == NULL: zeros)
 *   bot_spmm_dot_halves_f16            bot_spmm_dot_f32 with `out` replaced by the operand: hout[r, h * hsh + e] (h1) and + h2_off (2^11 h2) of
 *                                      hscale[0] * out[r,h,e]; all-heads layout only (bot_spmm_dot_halves_fits: 1 when the shape and the
 *                                      alignments are covered, else the caller keeps the fp32 form + a split pass)
 *   bot_gemm_halves3_nt2_f32 / _tn2_f32  the products above with the second scale: A's columns >= k_split (a multiple of 32) resp. D's
 *                                      columns >= p_split (a multiple of 4) are stored under scale_a2 / scale_d2 (NULL: one scale). */
int bot_halves_scale_from_slots2_f32(const uint32_t* slots, float mult, const float* cap_scale, float cap_ratio, float* scale, bot_stream_t stream);
int bot_halves_tail_f16(int64_t n, int32_t n_seg, const int64_t* seg_cols, const float* const* srcs, const int64_t* src_ld, const float* scale, uint16_t* out,
                        int64_t ldo, int32_t h2_off, bot_stream_t stream);
int bot_spmm_dot_halves_fits(const float* x, int64_t ldx, int64_t hsx, const float* y, int64_t ldy, int64_t hsy, int32_t H, int32_t D, const uint16_t* hout,
                             int64_t ldh, int64_t hsh, int32_t h2_off);
int bot_spmm_dot_halves_f16(const int32_t* indptr, const int32_t* indices, int64_t n_rows, int64_t nnz, const int32_t* items, int64_t n_items,
                            const int32_t* long_rows, const int32_t* long_ptr, int64_t n_long, const float* x, int64_t ldx, int64_t hsx, const float* w,
                            const int32_t* wperm, const float* y, int64_t ldy, int64_t hsy, int32_t H, int32_t D, const float* hscale, uint16_t* hout,
                            int64_t ldh, int64_t hsh, int32_t h2_off, float* dot_out, float* partial, bot_stream_t stream);
int bot_gemm_halves3_nt2_f32(int64_t m, int64_t n, int64_t k, const float* scale_a, const float* scale_a2, int64_t k_split, const float* scale_b,
                             const uint16_t* A, int64_t lda, int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, int32_t b_layout, float* C,
                             int64_t ldc, int32_t mode, bot_stream_t stream);
/* v18: the reduce pass of a fused BatchNorm / ReLU / dropout epilogue's backward (bot_bn_act_bwd_reduce_f32: 1 GB read per hidden layer at
 * config 2) as a BY-PRODUCT of the NT product that writes that epilogue's incoming gradient (bot_amd/nn/fused.py: the next layer's
 * `d h = d out . W^T`, backward of src/no-sampling/models.py:490-492 feeding the backward of :726-731).  With C = dy [m, n] and the
 * epilogue's input x [m, n] (n = the BatchNorm width F), the tile epilogue of bot_gemm_halves3_nt3_f32 forms, per stored element,
 *     g = dy * dropout_factor(seed, r, c) * [relu: weight_c xhat + bias_c > 0],   xhat = (x - mean_c) invstd_c
 * and leaves per 256-row tile t the partials  part[(2 t + 0) n + c] = sum_r g,  part[(2 t + 1) n + c] = sum_r g xhat  and (pmax != NULL)
 * pmax[(2 t + 0) n + c] = max_r |g|,  pmax[(2 t + 1) n + c] = max_r |xhat|  - the workspace layout of the reduce pass with one row block per
 * tile: ceil(m / 256) blocks (bot_gemm_halves3_nt_bn_rows(k): 256 when the product of piece width k takes this form - k a multiple of 64,
 * BOT_NT_KERNEL not 128x64 - else 0: the caller keeps the pass).  Finished by
 *   bot_bn_act_bwd_reduce_partials_f32  sum_g / sum_gx [F] from `part` (blocks added in order, in double: deterministic)
 *   bot_bn_bwd_bound_partials_f32       bot_bn_bwd_bound_f32 from `pmax`
 * The mask is the Philox function of (seed, r, c) the forward used (a quad of output columns = a quad of the mask).  x rows: 8-byte aligned,
 * pitch ldx >= n even; n even.  bn == NULL: bot_gemm_halves3_nt2_f32. */
typedef struct bot_bn_bwd_stats {
    const float* x;              /* [m, n] the epilogue's input (pre-BatchNorm), row pitch ldx */
    int64_t ldx;
    const float* mean;           /* [n] */
    const float* invstd;         /* [n] */
    const float* weight;         /* [n] or NULL (1) */
    const float* bias;           /* [n] or NULL (0) */
    int32_t relu;
    float p;                     /* dropout probability of the epilogue, 0: none */
    uint64_t seed;
    const uint64_t* seed_offset; /* optional device word mixed into the seed (hipGraph replays), as in bot_bn_act_fwd_f32 */
    float* part;                 /* [ceil(m / 256)][2][n] */
    float* pmax;                 /* [ceil(m / 256)][2][n] or NULL */
} bot_bn_bwd_stats_t;
int32_t bot_gemm_halves3_nt_bn_rows(int64_t k);
int bot_gemm_halves3_nt3_f32(int64_t m, int64_t n, int64_t k, const float* scale_a, const float* scale_a2, int64_t k_split, const float* scale_b,
                             const uint16_t* A, int64_t lda, int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, int32_t b_layout, float* C,
                             int64_t ldc, const bot_bn_bwd_stats_t* bn, int32_t mode, bot_stream_t stream);
int bot_bn_act_bwd_reduce_partials_f32(const float* part, int32_t nblk, int32_t F, float* sum_g, float* sum_gx, bot_stream_t stream);
int bot_bn_bwd_bound_partials_f32(int32_t F, const float* pmax, int32_t nblk, const float* sum_g, const float* sum_gx, double total_count,
                                  const float* weight, const float* invstd, uint32_t* absmax_slots, bot_stream_t stream);
/* both in ONE launch, for one rank (no cross-rank reduction of the sums in between): sum_g / sum_gx as _reduce_partials forms them, then the
 * bound from them (batch_stats = 0: eval statistics, the sums do not enter it) */
int bot_bn_bwd_partials_finish_f32(const float* part, const float* pmax, int32_t nblk, int32_t F, float* sum_g, float* sum_gx, int32_t batch_stats,
                                   double total_count, const float* weight, const float* invstd, uint32_t* absmax_slots, bot_stream_t stream);
/* v17: a RIGHT operand in FRAGMENT-MAJOR layout (b_layout = 1 of bot_gemm_halves3_nt2_f32; ldb / b2_off unused): the 16 bytes lane l of an
 * MFMA fragment holds - row 16 t + (l & 15), columns 32 s + 8 (l >> 4) .. + 7 - at halves ((t T + s) 64 + l) 8, T = piece / 32, h1 in the
 * first region, h2 ceil(n / 16) T 512 halves behind it: the 64 lanes of a fragment load read one contiguous KB (the NT kernel pays for the
 * bytes a CU pulls from its L2 by their address pattern: profiles/r05_nt64.txt).  out: 2 * ceil(n / 16) * (piece / 32) * 512 halves, 16-byte
 * aligned; piece a multiple of 64; rows beyond n and columns beyond F are zeros. */
int bot_halves_split_frag_f16(const float* x, int64_t ldx, int64_t n, int32_t F, const float* scale, uint16_t* out, int32_t piece, bot_stream_t stream);
int bot_gemm_halves3_tn2_f32(int64_t n_rows, int64_t k, int64_t p, int64_t kp, int64_t pp, const float* scale_x, const float* scale_d, const float* scale_d2,
                             int64_t p_split, const uint16_t* X, int64_t ldx, int64_t x2_off, const uint16_t* D, int64_t ldd, int64_t d2_off, float* out,
                             int64_t ldo, float* workspace, int32_t mode, bot_stream_t stream);

/* v16: GROUPED forms of the two products above — the column tiles (NT) / output tiles (TN) of ONE launch are a host-side list, each entry
 * with its own operand columns and output block.  They carry the aggregate-first GAT layer (fused.py:_GATHiddenAggFirst; models.py:490-492,
 * :547, :558-560 reordered by linearity): per head h  rst_h = [x | z_h] [Wres_h | W_h]^T  (one reduction over two column ranges of the
 * left operand), d z_h = d rst_h W_h, and the weight gradients d W_h = d rst_h^T z_h, d Wres = d rst^T x as blocks of one launch.
 *
 *   nt_grouped  groups[g] = (b_row0, n_valid, a_col0, a_col1, k_steps, c_off):
 *     C[r * ldc + c_off + j] = scale_a[1] scale_b[1] * sum_{t < k_steps} sum_{i < 32} A3[r, (t < k_seg ? a_col0 : a_col1) + 32 t + i] . B3[b_row0 + j, 32 t + i]
 *     for r < m, j < n_valid <= 256 (A3 . B3: the three products a1 b1 + a1 b2 + a2 b1; a1 at the given column, 2^11 a2 a2_off behind it;
 *     b1 at column 32 t + i of B's row, b2 b2_off behind it).  1 <= n_groups <= 12; a_col0, a_col1 multiples of 8; b_rows = rows of B.
 *     Optional epilogue (the eval-mode layer, models.py:726-734): col_scale / col_shift (either may be NULL), indexed by c_off + j — groups
 *     that write the columns of ONE [m, ldc] matrix —, then ReLU if `relu`, then max|C| into `absmax_slots` (NULL: none):
 *     C = relu?(C * col_scale + col_shift).
 *   tn_grouped  tiles[i] = (x_col0, k_valid, d_col0, p_valid, out_off, ldo, transposed):
 *     out[out_off + k * ldo + p] = scale_x[1] scale_d[1] * sum_n X3[n, x_col0 + k] . D3[n, d_col0 + p]   for k < k_valid <= 192, p < p_valid <= 192
 *     (transposed != 0: out[out_off + p * ldo + k]); 1 <= n_tiles <= 16; workspace: bot_gemm_halves3_tn_grouped_workspace_floats(n_rows, n_tiles) floats.
 * `groups` / `tiles` are HOST arrays of 6 resp. 7 int64 per entry (copied into the kernel arguments). */
int bot_gemm_halves3_nt_grouped_f32(int64_t m, int64_t b_rows, const float* scale_a, const float* scale_b, const uint16_t* A, int64_t lda,
                                    int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, float* C, int64_t ldc, int32_t n_groups,
                                    const int64_t* groups, int32_t k_seg, const float* col_scale, const float* col_shift, int32_t relu,
                                    uint32_t* absmax_slots, int32_t mode, bot_stream_t stream);
/* v17: nt_grouped with column statistics of the stored values as a by-product (BatchNorm's batch statistics without a pass over the
 * layer's output): per 256-row tile t and output column c = c_off + j, stats_pivot[t stats_F + c] = the tile's FIRST stored value of the
 * column (v19: written by the launch - a pivot inside the data keeps the sum of squares from cancelling for columns whose mean is far
 * from zero; v17/18 read one caller-given pivot per column), stats_part[(2 t + 0) stats_F + c] = sum over the tile's rows of
 * (C - pivot), [(2 t + 1) stats_F + c] the sum of squares, stats_minmax likewise the column minimum / maximum - ceil(m / 256) row blocks in
 * the layout of bot_bn_stats_halves_partials_f32, which combines the tiles exactly (in double) into mean / invstd / running statistics /
 * the epilogue's halves scale.  All three NULL: no statistics (= bot_gemm_halves3_nt_grouped_f32). */
int bot_gemm_halves3_nt_grouped2_f32(int64_t m, int64_t b_rows, const float* scale_a, const float* scale_b, const uint16_t* A, int64_t lda,
                                     int64_t a2_off, const uint16_t* B, int64_t ldb, int64_t b2_off, float* C, int64_t ldc, int32_t n_groups,
                                     const int64_t* groups, int32_t k_seg, const float* col_scale, const float* col_shift, int32_t relu,
                                     uint32_t* absmax_slots, float* stats_part, float* stats_minmax, float* stats_pivot, int32_t stats_F,
                                     int32_t mode, bot_stream_t stream);
/* (v19) part / minmax [nblk][2][F], pivot [nblk][F], nblk = ceil(n / 256) row blocks of 256 rows (the last one shorter). */
int bot_bn_stats_halves_partials_f32(const float* part, const float* minmax, int32_t nblk, const float* pivot, int64_t n, int32_t F, float eps, float momentum,
                                     float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                     const float* weight, const float* bias, float p, float* hscale, float* bound_workspace, bot_stream_t stream);
int64_t bot_gemm_halves3_tn_grouped_workspace_floats(int64_t n_rows, int32_t n_tiles);
int bot_gemm_halves3_tn_grouped_f32(int64_t n_rows, const float* scale_x, const float* scale_d, const uint16_t* X, int64_t ldx, int64_t x2_off,
                                    const uint16_t* D, int64_t ldd, int64_t d2_off, float* out, int32_t n_tiles, const int64_t* tiles,
                                    float* workspace, int32_t mode, bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * v14: the train step's glue (csrc/step.hip) — what src/no-sampling/run.py does around the model with a dozen small tensor ops per
 * step, as four launches.  Deterministic (fixed-order reductions, no atomics); Philox streams as in bot_bn_act_fwd_f32 (`seed`,
 * optional device word `seed_offset` for hipGraph replays).
 *
 *   label_split   run.py:256-267.  keep_i = mask[i] (uint8, may be NULL) or u_i < mask_rate.  use_labels != 0: code[train_idx[i]] =
 *                 keep_i ? label : -1 (the node's label is an input feature this step, run.py:259-263) and wn[train_idx[i]] = !keep_i
 *                 (it is a prediction node); use_labels == 0: wn[train_idx[i]] = keep_i (run.py:265-267).  count[0] = sum of the wn written.
 *                 `code` / `wn` are [N] arrays the CALLER initialised once to -1 / 0 (entries of non-training nodes never change);
                 `workspace`: 128 int32 words (per-workgroup counts, folded in slot order: two launches).
 *   build_input   `add_labels` (run.py:240-243) + the stack's input dropout (models.py:711) in one pass:
 *                 out[n, :] = dropout_p([feat[n, :F] | onehot(code[n])[:C]]), survivors scaled by 1 / (1 - p); C may be 0.
 *   node_loss     run.py:229-236 per node and its gradient: ce = logsumexp(x) - x[label]; kind 0 logit: y = ce, 1 loge: y = log(eps + ce)
 *                 - log eps, 2 savage: y = (1 - exp(-ce))^2;  y_out[n] = wn[n] > 0 ? y : 0 (entries n .. n_pad - 1 are zeroed: pad to a
 *                 multiple of 64 for the fixed-order sum that follows);  dx[n, c] = wn[n] > 0 ? y'(ce) (softmax(x)[c] - [c == label]) /
 *                 count[0] : 0 (dx may be NULL); C <= 128.  Labels of nodes with wn = 0 may be placeholders (clamped into range, never used).
 *   rmsprop_step  torch.optim.RMSprop (run.py:331-333; momentum 0, not centered) for n_tensors <= 48 parameters in ONE launch:
 *                 g += weight_decay p; sq = alpha sq + (1 - alpha) g^2; p -= lr g / (sqrt(sq) + eps).  `params` / `grads` / `square_avg`
 *                 / `numel` are HOST arrays (of device pointers / element counts), copied into the launch; lr_dev (may be NULL): a device
 *                 scalar that overrides `lr` (captured steps).
 * ------------------------------------------------------------------------------------------- */
int bot_label_split_f32(const int64_t* train_idx, int64_t n_train, const int64_t* labels, int64_t ldl, const uint8_t* mask, float mask_rate,
                        uint64_t seed, const uint64_t* seed_offset, int32_t use_labels, int32_t* code, float* wn, float* count, int32_t* workspace,
                        bot_stream_t stream);
int bot_build_input_f32(const float* feat, int64_t ldf, int64_t n, int32_t F, int32_t C, const int32_t* code, float p, uint64_t seed,
                        const uint64_t* seed_offset, float* out, int64_t ldo, bot_stream_t stream);
int bot_node_loss_f32(const float* x, int64_t ldx, int64_t n, int32_t C, const int64_t* labels, int64_t ldl, const float* wn, const float* count,
                      int32_t kind, float eps, float* y, int64_t n_pad, float* dx, int64_t lddx, bot_stream_t stream);
int bot_rmsprop_step_f32(int32_t n_tensors, float* const* params, const float* const* grads, float* const* square_avg, const int64_t* numel,
                         float lr, const float* lr_dev, float alpha, float eps, float weight_decay, bot_stream_t stream);

/* solution index (hipblaslt_ext::getIndexFromAlgo) and search time in ms of the kernel the last gemm_halves call used; what
 * tools/tune_halves_gemm.py records into bot_amd/tuning/halves_gemm.json and passes back as `algo_index` (-1: none) */
int bot_gemm_halves_last_algo(int32_t* index, float* ms);
int bot_gemm_halves_library_version(int32_t* compiled, int32_t* runtime);

/* ---------------------------------------------------------------------------------------------
 * Small-K projections: C[m,n] = (accumulate ? C : 0) + A[m,k] op(B), k <= 256, m huge — the per-head products of the
 * aggregate-before-project layer (W_h . sum_u a x_u: src/no-sampling/models.py:490-492 applied after :547, the reordering
 * GraphConv does at :377-385) and its residual / score columns (:519-522, :553-557).  fp32 operands are split in registers into
 * three bf16 terms each (exactly; bf16 has fp32's exponent range, so no scale), six bf16 MFMA products per tile
 * (a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1, dropped terms <= 2^-24 relative), fp32 accumulation: fp32-GEMM accuracy without
 * any pass over the operands.  b_is_kn: B is stored [k, n] (1) or [n, k] (0, nn.Linear's weight layout), row-major with
 * pitch ldb.  batch > 1: strided batches (element strides), e.g. the H heads writing side by side into the columns of one matrix.
 * ------------------------------------------------------------------------------------------- */
int bot_skinny_gemm_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int32_t b_is_kn, float* C, int64_t ldc, int64_t m,
                        int32_t n, int32_t k, int32_t accumulate, int32_t batch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                        bot_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradients of the small-K projections: out[kx, ky] = sum_r X[r, kx] Y[r, ky] over the n node rows, kx / ky a few hundred
 * (d W_r = h^T d out2, d W_i = d x_i^T z_i of the aggregate-before-project layer; the backward of the nn.Linear calls at
 * src/no-sampling/models.py:490-492, 553-557).  Exact fp32 MFMA (v_mfma_f32_32x32x2_f32: with the reduction index on the rows both
 * operands are read as coalesced row segments, no transposition), one workgroup per 256 x 192 output block and row chunk with the rows
 * staged through LDS, transpose_out: out[ky, kx] instead (put the wider operand first: blocks are 256 of X by 192 of Y), per-chunk
 * partials in `workspace` (bot_tn_gemm_workspace_floats) added in chunk order: deterministic.  batch > 1: element strides.
 * ------------------------------------------------------------------------------------------- */
/* v16: the same product for an X of a handful of columns (kx <= 32, ky <= 256: the attention columns of the merged gradient against the
 * layer input): plain fp32 FMAs, one thread per column of Y, per-workgroup partials added in workgroup order.
 * workspace: bot_tn_narrow_workspace_floats(n, kx, ky) floats. */
int64_t bot_tn_narrow_workspace_floats(int64_t n, int32_t kx, int32_t ky);
int bot_tn_narrow_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t n, int32_t kx, int32_t ky, float* out, int64_t ldo,
                      int32_t transpose_out, float* workspace, bot_stream_t stream);
int64_t bot_tn_gemm_workspace_floats(int64_t n, int32_t kx, int32_t ky, int32_t batch);
int bot_tn_gemm_f32(const float* X, int64_t ldx, const float* Y, int64_t ldy, int64_t n, int32_t kx, int32_t ky, float* out, int64_t ldo,
                    int32_t transpose_out, int32_t batch, int64_t stride_x, int64_t stride_y, int64_t stride_out, float* workspace,
                    bot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BOT_GNN_H */
