/*
 * TEST INFRASTRUCTURE ONLY — plain-C restatement of the CPU kernels behind the `dgl 0.5.*` operators
 * the reference calls (copy_u_sum / u_mul_e_sum SpMM, dot / u_add_v SDDMM, edge_softmax), used
 *   (1) as a second, independent oracle next to oracle/ref_ops.py (tests/test_oracle_c.py), and
 *   (2) as the timed "DGL-equivalent CPU restatement" (`cpu_baseline.kind = "port"`) in bench.py.
 * PARITY UNPINNED (the arithmetic lives in the un-vendored `dgl 0.5.*`, reference README.md:9): the
 * loop structure follows DGL's published CPU kernels [upstream-DGL, recalled: src/array/cpu/spmm.h,
 * sddmm.h] — OpenMP parallel-for over destination rows, sequential fp32 sum inside a row in CSR
 * position (= ascending edge-id) order, feature loop innermost; ids are int64 like DGL's default
 * idtype.  Never linked into, or loaded by, the product (bot_amd).
 *
 * Reference call sites: src/no-sampling/models.py:374,381 (copy_u_sum), :547 (u_mul_e_sum),
 * :523,525 (u_add_v / copy_u), :537,544 (edge_softmax); src/ogbn-proteins/gat.py:58 (copy_e_sum).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* REAL = float: the restatement proper (liboracle.so).  -DORACLE_F64 builds the same loops in double (liboracle_f64.so): the
 * "exact" side when two fp32 implementations have to be ranked against each other (tests/full_size.py). */
#ifdef ORACLE_F64
#define REAL double
#define FMAX fmax
#define EXP exp
#else
#define REAL float
#define FMAX fmaxf
#define EXP expf
#endif

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* out[r,h,:] = sum_{k in row r} w[eid[k],h] * x[indices[k],h,:]     (w == NULL: copy_u_sum) */
void oracle_spmm(const int64_t* indptr, const int64_t* indices, const int64_t* eid, int64_t n_rows, const REAL* x,
                 const REAL* w, int64_t H, int64_t D, REAL* out) {
    const int64_t F = H * D;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        REAL* o = out + r * F;
        memset(o, 0, sizeof(REAL) * F);
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
            const REAL* xs = x + indices[k] * F;
            if (w) {
                const REAL* we = w + eid[k] * H;
                for (int64_t h = 0; h < H; ++h)
                    for (int64_t d = 0; d < D; ++d) o[h * D + d] += xs[h * D + d] * we[h];
            } else {
                for (int64_t j = 0; j < F; ++j) o[j] += xs[j];
            }
        }
    }
}

/* out[eid[k],h] = < x[indices[k],h,:], y[r,h,:] > */
void oracle_sddmm_dot(const int64_t* indptr, const int64_t* indices, const int64_t* eid, int64_t n_rows, const REAL* x,
                      const REAL* y, int64_t H, int64_t D, REAL* out) {
    const int64_t F = H * D;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        const REAL* yr = y + r * F;
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
            const REAL* xs = x + indices[k] * F;
            for (int64_t h = 0; h < H; ++h) {
                REAL s = (REAL)0;
                for (int64_t d = 0; d < D; ++d) s += xs[h * D + d] * yr[h * D + d];
                out[eid[k] * H + h] = s;
            }
        }
    }
}

/* out[e,:] = x[src[e],:] (+ y[dst[e],:]) */
void oracle_u_add_v(const int64_t* src, const int64_t* dst, int64_t n_edges, const REAL* x, const REAL* y, int64_t W,
                    REAL* out) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n_edges; ++e)
        for (int64_t j = 0; j < W; ++j) out[e * W + j] = x[src[e] * W + j] + (y ? y[dst[e] * W + j] : (REAL)0);
}

/* a[eid[k],h] = softmax over the positions k of row r of e[eid[k],h]; keep == 0 edges are excluded, a = 0 */
void oracle_edge_softmax_fwd(const int64_t* indptr, const int64_t* eid, int64_t n_rows, const REAL* e,
                             const uint8_t* keep, int64_t H, REAL* a) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        for (int64_t h = 0; h < H; ++h) {
            REAL m = -INFINITY, s = (REAL)0;
            for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
                if (!keep || keep[eid[k]]) m = FMAX(m, e[eid[k] * H + h]);
            for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
                if (!keep || keep[eid[k]]) s += EXP(e[eid[k] * H + h] - m);
            for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
                a[eid[k] * H + h] = (!keep || keep[eid[k]]) ? EXP(e[eid[k] * H + h] - m) / s : (REAL)0;
        }
    }
}

/* de = a*da - a * sum_row(a*da) */
void oracle_edge_softmax_bwd(const int64_t* indptr, const int64_t* eid, int64_t n_rows, const REAL* a, const REAL* da,
                             int64_t H, REAL* de) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        for (int64_t h = 0; h < H; ++h) {
            REAL t = (REAL)0;
            for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) t += a[eid[k] * H + h] * da[eid[k] * H + h];
            for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k) {
                const int64_t o = eid[k] * H + h;
                de[o] = a[o] * da[o] - a[o] * t;
            }
        }
    }
}

/* out[r,:] = sum_{k in row r} vals[eid[k],:] */
void oracle_segment_sum(const int64_t* indptr, const int64_t* eid, int64_t n_rows, const REAL* vals, int64_t W,
                        REAL* out) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rows; ++r) {
        REAL* o = out + r * W;
        memset(o, 0, sizeof(REAL) * W);
        for (int64_t k = indptr[r]; k < indptr[r + 1]; ++k)
            for (int64_t j = 0; j < W; ++j) o[j] += vals[eid[k] * W + j];
    }
}
