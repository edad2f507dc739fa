// bot_gemm_halves_f32: C[m,n] = alpha[n] * op(A)[m,k] op(B)[k,n] + beta * C for row-major fp16 operands with fp32 accumulation and output,
// alpha a DEVICE vector over the n output columns (the product of two halves_scale reciprocals, never seen by the host;
// hipBLASLt's device-SCALAR pointer mode is not honoured by the library build torch ships, the device-vector mode is) — hipBLASLt does the
// MFMA work (a plain library GEMM; the halves format around it is halves.hip).  Optional strided batches (the row chunks of a
// weight gradient).  Kernel choice per shape, first call: a recorded solution index when the caller has one (algo_index >= 0);
// else hipBLASLt's first heuristic choice (tune == 0, the default of the Python binding: no timing, no synchronisation,
// reproducible).  Opt-in: (tune == 1) the fastest of the 16 heuristic candidates timed on the caller's buffers, or (tune == 2, the
// tuning tool) of ALL the library's solutions for these types; only for beta == 0, where repeated runs are idempotent, and never
// under stream capture (top heuristic then).
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt-version.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

namespace bot {
namespace {

using Key = std::tuple<int, int, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int64_t, int64_t, int64_t, int, bool>;

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool tuned = false;
    int index = -1;        // solution index of `algo` when known
    float ms = 0.f;        // its time in the search that picked it
    int searched = 0;      // candidates timed by the exhaustive search
    std::vector<hipblasLtMatmulHeuristicResult_t> cand;
};

int g_last_index = -1;     // solution index / time of the kernel the last call selected (bot_gemm_halves_last_algo)
float g_last_ms = 0.f;
std::mutex g_mu;
std::map<Key, Plan> g_plans;
std::map<int, hipblasLtHandle_t> g_handles;   // one library handle per device
int g_runtime_version = 0;                    // hipblasLtGetVersion of the library actually loaded

void destroy(Plan& p) {
    if (p.la) hipblasLtMatrixLayoutDestroy(p.la);
    if (p.lb) hipblasLtMatrixLayoutDestroy(p.lb);
    if (p.lc) hipblasLtMatrixLayoutDestroy(p.lc);
    if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
    p = Plan();
}

// Shapes vary without bound when the row count does (partitions, scaled graphs): keep the plan table small.  Evicted shapes are
// simply planned again.
constexpr size_t MAX_PLANS = 512;

#define LT_CHECK(expr, what)                                                   \
    do {                                                                       \
        hipblasStatus_t st_ = (expr);                                          \
        if (st_ != HIPBLAS_STATUS_SUCCESS) {                                   \
            set_error("gemm_halves: %s failed (hipblasStatus %d)", what, (int)st_); \
            return 1000 + (int)st_;                                            \
        }                                                                      \
    } while (0)

int make_layout(hipblasLtMatrixLayout_t* l, hipDataType t, int64_t rows, int64_t cols, int64_t ld, int32_t batch, int64_t stride) {
    LT_CHECK(hipblasLtMatrixLayoutCreate(l, t, (uint64_t)rows, (uint64_t)cols, ld), "MatrixLayoutCreate");
    if (batch > 1) {
        LT_CHECK(hipblasLtMatrixLayoutSetAttribute(*l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch)), "layout batch");
        LT_CHECK(hipblasLtMatrixLayoutSetAttribute(*l, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &stride, sizeof(stride)), "layout stride");
    }
    return 0;
}

}  // namespace
}  // namespace bot

extern "C" {

int bot_gemm_halves_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k, const float* alpha, const uint16_t* A,
                        int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int32_t batch, int64_t stride_a,
                        int64_t stride_b, int64_t stride_c, float beta, void* workspace, int64_t workspace_bytes, int32_t tune,
                        int32_t algo_index, bot_stream_t stream) {
    using namespace bot;
    BOT_REQUIRE(m >= 1 && n >= 1 && k >= 1 && batch >= 1, BOT_E_RANGE, "gemm_halves: m=%lld n=%lld k=%lld batch=%d", (long long)m,
                (long long)n, (long long)k, batch);
    BOT_REQUIRE(lda >= (trans_a ? m : k) && ldb >= (trans_b ? k : n) && ldc >= n, BOT_E_RANGE, "gemm_halves: lda=%lld ldb=%lld ldc=%lld",
                (long long)lda, (long long)ldb, (long long)ldc);
    BOT_REQUIRE(alpha && A && B && C && (workspace || workspace_bytes == 0), BOT_E_NULL, "gemm_halves: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipblasLtHandle_t& g_handle = g_handles[dev];
    if (!g_handle) {
        LT_CHECK(hipblasLtCreate(&g_handle), "hipblasLtCreate");
        int v = 0;
        if (hipblasLtGetVersion(g_handle, &v) == HIPBLAS_STATUS_SUCCESS) g_runtime_version = v;
    }
    const Key key{trans_a, trans_b, m, n, k, lda, ldb, ldc, batch, stride_a, stride_b, stride_c, dev, beta != 0.f};
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        // Built in a local and moved into the table only when complete: an error on the way leaves no half-initialised entry
        // behind (descriptor without layouts / algorithm) for later calls with the same key to run.
        Plan q;
        struct Guard {
            Plan& q;
            hipblasLtMatmulPreference_t pref = nullptr;
            bool done = false;
            ~Guard() {
                if (pref) hipblasLtMatmulPreferenceDestroy(pref);
                if (!done) destroy(q);
            }
        } guard{q};
        // row-major C = op(A) op(B)  <=>  column-major C^T = op(B)^T op(A)^T: B is hipBLASLt's first operand, A its second
        LT_CHECK(hipblasLtMatmulDescCreate(&q.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F), "MatmulDescCreate");
        const hipblasOperation_t op1 = trans_b ? HIPBLAS_OP_T : HIPBLAS_OP_N, op2 = trans_a ? HIPBLAS_OP_T : HIPBLAS_OP_N;
        LT_CHECK(hipblasLtMatmulDescSetAttribute(q.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op1, sizeof(op1)), "desc transA");
        LT_CHECK(hipblasLtMatmulDescSetAttribute(q.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op2, sizeof(op2)), "desc transB");
        const int32_t pm = HIPBLASLT_POINTER_MODE_ALPHA_DEVICE_VECTOR_BETA_HOST;
        LT_CHECK(hipblasLtMatmulDescSetAttribute(q.desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &pm, sizeof(pm)), "desc pointer mode");
        int rc;
        if ((rc = make_layout(&q.la, HIP_R_16F, trans_b ? k : n, trans_b ? n : k, ldb, batch, stride_b))) return rc;   // first = B
        if ((rc = make_layout(&q.lb, HIP_R_16F, trans_a ? m : k, trans_a ? k : m, lda, batch, stride_a))) return rc;   // second = A
        if ((rc = make_layout(&q.lc, HIP_R_32F, n, m, ldc, batch, stride_c))) return rc;
        LT_CHECK(hipblasLtMatmulPreferenceCreate(&guard.pref), "PreferenceCreate");
        const uint64_t wsb = (uint64_t)workspace_bytes;
        LT_CHECK(hipblasLtMatmulPreferenceSetAttribute(guard.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)), "pref workspace");
        q.cand.resize(16);
        int found = 0;
        LT_CHECK(hipblasLtMatmulAlgoGetHeuristic(g_handle, q.desc, q.la, q.lb, q.lc, q.lc, guard.pref, (int)q.cand.size(), q.cand.data(), &found),
                 "AlgoGetHeuristic");
        q.cand.resize(found);
        if (found == 0) {
            set_error("gemm_halves: hipBLASLt has no kernel for m=%lld n=%lld k=%lld", (long long)m, (long long)n, (long long)k);
            return 2;
        }
        q.algo = q.cand[0].algo;
        q.ws = q.cand[0].workspaceSize;
        if (g_plans.size() >= MAX_PLANS) {
            for (auto& kv : g_plans) destroy(kv.second);
            g_plans.clear();
        }
        guard.done = true;
        it = g_plans.emplace(key, std::move(q)).first;
    }
    Plan& p = it->second;
    auto run = [&](const hipblasLtMatmulAlgo_t& algo) {
        return hipblasLtMatmul(g_handle, p.desc, alpha, B, p.la, A, p.lb, &beta, C, p.lc, C, p.lc, &algo, workspace, (size_t)workspace_bytes, st);
    };
    if (!p.tuned && algo_index >= 0) {   // a recorded selection (bot_amd/tuning/halves_gemm.json, made by tune == 2 on this library build)
        std::vector<int> idx{algo_index};
        std::vector<hipblasLtMatmulHeuristicResult_t> res;
        size_t ws = 0;
        if (hipblaslt_ext::getAlgosFromIndex(g_handle, idx, res) == HIPBLAS_STATUS_SUCCESS && !res.empty() &&
            hipblaslt_ext::matmulIsAlgoSupported(g_handle, p.desc, alpha, p.la, p.lb, &beta, p.lc, p.lc, res[0].algo, ws) == HIPBLAS_STATUS_SUCCESS &&
            ws <= (size_t)workspace_bytes) {
            p.algo = res[0].algo, p.ws = ws, p.tuned = true, p.index = algo_index;
            p.cand.clear();
        }
    }
    if (!p.tuned && tune == 2 && beta == 0.f) {   // exhaustive: every solution of the library for these types that supports the problem
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        if (cs == hipStreamCaptureStatusNone) {
            const hipblasOperation_t op1 = trans_b ? HIPBLAS_OP_T : HIPBLAS_OP_N, op2 = trans_a ? HIPBLAS_OP_T : HIPBLAS_OP_N;
            std::vector<hipblasLtMatmulHeuristicResult_t> all;
            (void)hipblaslt_ext::getAllAlgos(g_handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, op1, op2, HIP_R_16F, HIP_R_16F, HIP_R_32F, HIP_R_32F,
                                             HIPBLAS_COMPUTE_32F, all);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            auto time_of = [&](hipblasLtMatmulAlgo_t& algo, int reps) {
                (void)hipEventRecord(e0, st);
                for (int i = 0; i < reps; ++i)
                    if (run(algo) != HIPBLAS_STATUS_SUCCESS) return 1e30f;
                (void)hipEventRecord(e1, st);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                return ms / reps;
            };
            std::vector<std::pair<float, size_t>> first;   // (one-run time, position in `all`)
            std::vector<size_t> wss(all.size(), 0);
            for (size_t i = 0; i < all.size(); ++i) {
                size_t ws = 0;
                if (hipblaslt_ext::matmulIsAlgoSupported(g_handle, p.desc, alpha, p.la, p.lb, &beta, p.lc, p.lc, all[i].algo, ws) != HIPBLAS_STATUS_SUCCESS ||
                    ws > (size_t)workspace_bytes)
                    continue;
                wss[i] = ws;
                if (run(all[i].algo) != HIPBLAS_STATUS_SUCCESS) continue;    // warm-up
                first.emplace_back(time_of(all[i].algo, 1), i);
            }
            std::sort(first.begin(), first.end());
            float best = 1e30f;
            for (size_t j = 0; j < first.size() && j < 12; ++j) {            // the dozen fastest once more, five runs each
                const size_t i = first[j].second;
                const float ms = time_of(all[i].algo, 5);
                if (ms < best) best = ms, p.algo = all[i].algo, p.ws = wss[i], p.index = hipblaslt_ext::getIndexFromAlgo(all[i].algo);
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (best < 1e29f) p.tuned = true, p.ms = best, p.cand.clear();
            p.searched = (int)first.size();
        }
    }
    if (!p.tuned && tune && beta == 0.f && p.cand.size() > 1) {   // timing runs overwrite C: only when nothing is accumulated
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        if (cs == hipStreamCaptureStatusNone) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            float best = 1e30f;
            for (const auto& c : p.cand) {
                if (c.state != HIPBLAS_STATUS_SUCCESS || c.workspaceSize > (size_t)workspace_bytes) continue;
                if (run(c.algo) != HIPBLAS_STATUS_SUCCESS) continue;          // warm-up (and: does it launch at all)
                (void)hipEventRecord(e0, st);
                bool ok = true;
                for (int i = 0; i < 3 && ok; ++i) ok = run(c.algo) == HIPBLAS_STATUS_SUCCESS;
                (void)hipEventRecord(e1, st);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ok && ms < best) best = ms, p.algo = c.algo, p.ws = c.workspaceSize, p.ms = ms / 3.f;
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            p.tuned = true;
            p.cand.clear();
        }
    }
    set_kernel("hipblaslt_f16_f32 m=%lld n=%lld k=%lld batch=%d", (long long)m, (long long)n, (long long)k, batch);
    if (p.index < 0) p.index = hipblaslt_ext::getIndexFromAlgo(p.algo);
    g_last_index = p.index, g_last_ms = p.ms;
    LT_CHECK(run(p.algo), "hipblasLtMatmul");
    return hip_status("gemm_halves launch");
}

int bot_gemm_halves_library_version(int32_t* compiled, int32_t* runtime) {
    std::lock_guard<std::mutex> lock(bot::g_mu);
    if (compiled) *compiled = HIPBLASLT_VERSION_MAJOR * 100000 + HIPBLASLT_VERSION_MINOR * 100 + HIPBLASLT_VERSION_PATCH;
    if (runtime) *runtime = bot::g_runtime_version;
    return 0;
}

int bot_gemm_halves_last_algo(int32_t* index, float* ms) {
    std::lock_guard<std::mutex> lock(bot::g_mu);
    if (index) *index = bot::g_last_index;
    if (ms) *ms = bot::g_last_ms;
    return 0;
}

}  // extern "C"
