// Host-side pieces of libbot_gnn.so: error reporting, ABI version, row plan (integer work only).
#include <cxxabi.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdarg.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <exception>
#include <typeinfo>
#include <vector>

#include "common.h"

namespace bot {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static thread_local char g_kernel[128] = "";
void set_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
}  // namespace bot

extern "C" {

int bot_abi_version(void) { return BOT_ABI_VERSION; }
const char* bot_last_error(void) { return bot::g_err; }
const char* bot_last_kernel(void) { return bot::g_kernel; }

// A HIP stream of the library's OWN (v17): bot_amd.side runs the weight-gradient products on it.  PyTorch hands out its streams from a pool
// of 32 per device round-robin: a pooled "second stream" held for the process lifetime is sooner or later the SAME stream a later
// torch.cuda.Stream() / the graph-capture stream / the process group's collective stream gets - a fork that waits on itself, or captured
// work interleaved with a collective's.  Non-blocking (no implicit synchronisation with the legacy default stream).  high_priority == 0 is
// the device's LOWEST priority, deliberately below torch's default-priority streams (include/bot_gnn.h).
int bot_stream_create(int32_t high_priority, bot_stream_t* out) {
    using namespace bot;
    BOT_REQUIRE(out != nullptr, BOT_E_NULL, "stream_create: NULL pointer");
    int lo = 0, hi = 0;
    hipStream_t s = nullptr;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);       // lo: the numerically largest = lowest priority
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo);
    BOT_REQUIRE(e == hipSuccess, (int)e, "stream_create: %s", hipGetErrorString(e));
    *out = (bot_stream_t)s;
    return 0;
}

// ---- v19 diagnostics: who called abort()? -------------------------------------------------------------------------------------
// Round 5's GPU suite died with SIGABRT inside a hipGraph capture and left only Python frames (faulthandler): abort() is raised on the
// CALLING thread, so a handler that prints that thread's native frames names the caller - the HIP runtime, RCCL's watchdog, libstdc++'s
// terminate.  Async-signal-safe calls only (backtrace_symbols_fd writes straight to the descriptor); then the previous handler runs
// (Python's faulthandler) and the default action ends the process as before.
namespace {
int g_trace_fd = -1;
struct sigaction g_prev_abrt;
std::terminate_handler g_prev_term = nullptr;

void put(const char* s) { if (g_trace_fd >= 0) (void)!write(g_trace_fd, s, strlen(s)); }
void put_num(long v) {
    char b[24];
    int i = 23;
    b[i] = 0;
    if (v == 0) b[--i] = '0';
    for (; v > 0 && i > 0; v /= 10) b[--i] = (char)('0' + v % 10);
    put(b + i);
}
void dump_frames(const char* why) {
    void* fr[96];
    put("\n=== libbot_gnn abort trace: ");
    put(why);
    put(" pid ");
    put_num((long)getpid());
    put(" tid ");
    put_num((long)syscall(SYS_gettid));
    put(" ===\n");
    int n = backtrace(fr, 96);
    if (g_trace_fd >= 0) backtrace_symbols_fd(fr, n, g_trace_fd);
    put("=== end of trace ===\n");
}
void on_abort(int sig, siginfo_t* info, void* ctx) {
    dump_frames("SIGABRT");
    sigaction(SIGABRT, &g_prev_abrt, nullptr);          // hand over: faulthandler's dump, then the default action
    if ((g_prev_abrt.sa_flags & SA_SIGINFO) && g_prev_abrt.sa_sigaction) g_prev_abrt.sa_sigaction(sig, info, ctx);
    else if (g_prev_abrt.sa_handler != SIG_DFL && g_prev_abrt.sa_handler != SIG_IGN && g_prev_abrt.sa_handler) g_prev_abrt.sa_handler(sig);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}
void on_terminate() {
    put("\n=== libbot_gnn: std::terminate");
    if (std::type_info* t = abi::__cxa_current_exception_type()) {
        put(", uncaught ");
        put(t->name());
        try {
            throw;
        } catch (const std::exception& e) {
            put(": ");
            put(e.what());
        } catch (...) {
        }
    }
    put(" ===\n");
    dump_frames("std::terminate");
    if (g_prev_term) g_prev_term();
    abort();
}
}  // namespace

int bot_debug_abort_trace(const char* path) {
    using namespace bot;
    BOT_REQUIRE(path != nullptr, BOT_E_NULL, "debug_abort_trace: NULL path");
    int fd = open(path, O_WRONLY | O_CREAT | O_APPEND | O_CLOEXEC, 0644);
    BOT_REQUIRE(fd >= 0, BOT_E_RANGE, "debug_abort_trace: cannot open %s", path);
    void* warm[4];
    backtrace(warm, 4);                                  // loads libgcc's unwinder now: dlopen inside a signal handler is not safe
    const bool first = g_trace_fd < 0;
    if (!first) close(g_trace_fd);
    g_trace_fd = fd;
    if (first) {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = on_abort;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGABRT, &sa, &g_prev_abrt);
        g_prev_term = std::set_terminate(on_terminate);
    }
    return 0;
}

int32_t bot_row_plan_default_chunk(int64_t nnz) {
    // cdna_hip_programming.md Appendix B "Scatter / gather": split lists longer than a quarter of one
    // wavefront's share of the rows; 256 CUs x 16 waves in flight.
    int64_t share = nnz / (256 * 16) / 4;
    int32_t c = 64;
    while (c * 2 <= share && c < 512) c *= 2;
    return c;
}

static int plan_check(const int32_t* indptr, int64_t n_rows, int32_t chunk) {
    BOT_REQUIRE(indptr != nullptr, BOT_E_NULL, "row plan: indptr_host is NULL");
    BOT_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX, BOT_E_RANGE, "row plan: n_rows=%lld out of range", (long long)n_rows);
    BOT_REQUIRE(chunk >= 1, BOT_E_RANGE, "row plan: chunk=%d must be >= 1", chunk);
    for (int64_t r = 0; r < n_rows; ++r)
        BOT_REQUIRE(indptr[r + 1] >= indptr[r], BOT_E_PLAN, "row plan: indptr not monotone at row %lld", (long long)r);
    return 0;
}

int bot_row_plan_size_host(const int32_t* indptr, int64_t n_rows, int32_t chunk, int64_t* n_items, int64_t* n_long,
                           int64_t* n_slots) {
    if (int rc = plan_check(indptr, n_rows, chunk)) return rc;
    BOT_REQUIRE(n_items && n_long && n_slots, BOT_E_NULL, "row plan: output pointer is NULL");
    int64_t items = 0, longs = 0, slots = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t deg = (int64_t)indptr[r + 1] - indptr[r];
        if (deg > chunk) {
            int64_t c = (deg + chunk - 1) / chunk;
            items += c, slots += c, longs += 1;
        } else {
            items += 1;
        }
    }
    *n_items = items, *n_long = longs, *n_slots = slots;
    return 0;
}

int bot_row_plan_fill_host(const int32_t* indptr, int64_t n_rows, int32_t chunk, int32_t* items, int32_t* long_rows,
                           int32_t* long_ptr) {
    if (int rc = plan_check(indptr, n_rows, chunk)) return rc;
    BOT_REQUIRE(items && long_ptr, BOT_E_NULL, "row plan: output pointer is NULL");
    // 1. long rows, in row order; their chunks come first (every chunk but the last is full-size).
    int64_t it = 0, nl = 0, slot = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        int32_t beg = indptr[r], end = indptr[r + 1];
        if ((int64_t)end - beg <= chunk) continue;
        BOT_REQUIRE(long_rows != nullptr, BOT_E_NULL, "row plan: long_rows_host is NULL but long rows exist");
        long_rows[nl] = (int32_t)r;
        long_ptr[nl] = (int32_t)slot;
        for (int32_t b = beg; b < end; b += chunk) {
            int32_t* q = items + 4 * it++;
            q[0] = (int32_t)r, q[1] = b, q[2] = std::min<int64_t>((int64_t)b + chunk, end), q[3] = (int32_t)slot++;
        }
        ++nl;
    }
    long_ptr[nl] = (int32_t)slot;
    // 2. whole rows, longest first (counting sort on the degree, stable in the row id) so that the
    //    groups sharing a wavefront have similar trip counts and heavy items start early.
    std::vector<int64_t> start(chunk + 2, 0);
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t deg = (int64_t)indptr[r + 1] - indptr[r];
        if (deg <= chunk) start[chunk - deg + 1] += 1;  // bucket 0 = degree `chunk`
    }
    for (int32_t b = 0; b <= chunk; ++b) start[b + 1] += start[b];
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t deg = (int64_t)indptr[r + 1] - indptr[r];
        if (deg > chunk) continue;
        int32_t* q = items + 4 * (it + start[chunk - deg]++);
        q[0] = (int32_t)r, q[1] = indptr[r], q[2] = indptr[r + 1], q[3] = -1;
    }
    return 0;
}

}  // extern "C"
