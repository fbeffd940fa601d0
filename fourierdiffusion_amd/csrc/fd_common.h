// fd_common.h -- shared host-side plumbing for libfdiff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fdiff_hip.h"

struct fd_ctx {
    int device = 0;
    std::string err;
    // first error raised by a void helper deep inside a multi-launch call (fixed-order reductions whose scratch could not be
    // allocated): the enclosing API call returns it (fd_take_deferred) instead of FD_OK with gradients partly unwritten
    int deferred_rc = 0;
    // grow-only scratch arena; carved per call by fd_ws
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // bumped whenever a call (re)carves or regrows the arena: the training forward stamps it and the backward verifies
    // that nothing else used the arena in between (the saved activations live there)
    uint64_t ws_gen = 0;
    // small pinned staging buffer for per-call coefficient tables
    void* comm = nullptr;   // ncclComm_t when fd_comm_init succeeded
    int rank = 0, nranks = 1;
    int num_cu = 256;
    // split-K partial sums of the exact-f32 GEMMs (allocated on first use, freed with the context)
    float* gemm_scratch = nullptr;
    size_t gemm_scratch_floats = 0;
    // per-block partial sums of the two-stage (fixed-order, atomic-free) reductions: column sums, LayerNorm parameter
    // gradients, gradient norm
    float* red_scratch = nullptr;
    size_t red_scratch_floats = 0;
    // side stream + events: the training backward runs each layer's weight-gradient kernel beside the (latency-bound)
    // input-gradient chain of the earlier layers
    hipStream_t side_stream = nullptr;
    hipStream_t mask_stream = nullptr;    // dropout decisions of the NEXT training step: their own stream, so that they start behind the last
                                          // reader of the decision buffers instead of behind the previous step's last weight-gradient launch
    hipStream_t side_stream2 = nullptr;   // weight-gradient launches alternate between the two (they are latency-bound on ~80 CUs each)
    std::vector<hipEvent_t> side_events;
    // recorded behind the last reader (training forward / backward) of the dropout-decision buffers, which live in the arena;
    // tr_readers_gen = ws_gen at that moment: a different ws_gen at the next training forward means some other call carved the
    // arena in between and may still be running on the caller's stream, so the side stream must wait for a FRESH event
    hipEvent_t tr_readers_event[2] = {nullptr, nullptr};      // one per set of decision buffers (alternating training steps)
    bool tr_readers_event_valid[2] = {false, false};
    uint64_t tr_readers_gen = 0;
    unsigned tr_mask_steps = 0;
    const void* tr_last_model = nullptr;      // (model, batch) of the last bf16 training forward: another one lays the arena out differently
    int tr_last_B = 0;
    // F-split of the training FFN kernels (fd_train_bf16.hip, struct FSplit): partial accumulators handed from the producer to the
    // finisher workgroup of a token block, one flag per token tile (zeroed at allocation; a launch writes its own epoch)
    float* tr_ypart = nullptr;
    unsigned* tr_yflag = nullptr;
    size_t tr_fsplit_blocks = 0;
    unsigned tr_epoch = 0;
    // error word of the device-side bounded waits (pinned host memory mapped into the device; a kernel sets it with a
    // system-scope atomic, the host reads it without synchronising): fd_train_async_check turns it into FD_ERR_STATE
    unsigned* tr_err_host = nullptr;
    unsigned* tr_err_dev = nullptr;
    unsigned* tr_err_gpu = nullptr;      // device-resident copy (set with the host word): what the optimizer kernel tests in stream order
    // persistent training forward (fd_train_persist.hip): one 64-bit flag per (series, token tile), zeroed at allocation, and the launch
    // counter their values are built from
    unsigned long long* trp_flags = nullptr;
    size_t trp_flag_count = 0;
    unsigned long long trp_epoch = 0;
    bool trp_disabled = false;            // a cluster wait of the persistent forward timed out once on this context: per-layer kernels from then on
    // FFT twiddle tables (T, device pointer), built on first use of a length
    std::vector<std::pair<int, void*>> fft_tw;
    // measurement hooks (fd_prof_begin / fd_prof_end)
    // several kernels may be bracketed in one window (the training step brackets its forward FFN kernel and the weight-gradient
    // kernel); fd_prof_end reports the one with the largest TOTAL time.  At most 1024 launches per kernel name are bracketed (a
    // timing event is a barrier packet: bracketing all 20 000 layer launches of a step-by-step sampler run would slow the run
    // it measures; every bracketed name of a window is launched equally often, so the totals stay comparable)
    bool prof_on = false;
    unsigned long long* prof_clk = nullptr;   // device, 4 words: the persistent kernel's clock stamps of the window's last launch (fd_mega_params::clk_out)
    double prof_clock_mhz = 0.0;              // shader clock of that launch (fd_prof_end), 0 when none was measured
    struct prof_kernel { std::string name; double flops; int scopes = 0; int seen = 0; };
    int prof_stride = 1;       // bracket every prof_stride-th launch of a name (fd_prof_stride)
    std::vector<prof_kernel> prof_kernels;
    struct prof_event { int kernel; hipEvent_t a, b; };
    std::vector<prof_event> prof_events;
};

float* fd_gemm_scratch(fd_ctx* ctx, size_t* n_floats);   // fd_ctx.hip
// FD_OK, or FD_ERR_STATE when a kernel of an EARLIER training call on this context gave up a bounded inter-workgroup wait
// (fd_train_bf16.hip, struct FSplit); called at the entry of every training / optimizer call, never synchronises (fd_ctx.hip)
int fd_train_async_check(fd_ctx* ctx);
float* fd_red_scratch(fd_ctx* ctx, size_t n_floats);      // fd_ctx.hip; nullptr when the allocation fails
// out[n] (+)= sum_r part[r][n], r in ascending order (fd_score_bwd.hip)
void fd_sum_rows(const float* part, int R, int N, float* out, bool accumulate, hipStream_t s);
// out[n] += sum_m x[m][n] without atomics: per-block partials, then fd_sum_rows (fd_score_bwd.hip)
int fd_colsum_det(fd_ctx* ctx, const float* x, float* out, int M, int N, hipStream_t s);

// Bracket one launch of the dominant kernel with events on its stream (no-op unless profiling is on).
struct fd_prof_scope {
    fd_ctx* ctx;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    int kernel = -1;
    fd_prof_scope(fd_ctx* c, hipStream_t st, const char* name, double flops) : ctx(c), s(st) {
        if (!ctx->prof_on || ctx->prof_events.size() >= 8192) { ctx = nullptr; return; }
        for (size_t i = 0; i < ctx->prof_kernels.size(); ++i)
            if (ctx->prof_kernels[i].name == name) kernel = (int)i;
        if (kernel >= 0 && ctx->prof_kernels[kernel].scopes >= 1024) { ctx = nullptr; return; }
        if (kernel < 0) { ctx->prof_kernels.push_back({name, flops}); kernel = (int)ctx->prof_kernels.size() - 1; }
        ctx->prof_kernels[kernel].flops = flops;
        // a timing event is a barrier packet on the stream (~2.5 us of device time each): the training step launches its two
        // bracketed kernels 20 times, so bracketing every launch cost the step it measures 0.1 ms; bench.py samples them
        if (ctx->prof_kernels[kernel].seen++ % ctx->prof_stride != 0) { ctx = nullptr; return; }
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { ctx = nullptr; return; }
        ++ctx->prof_kernels[kernel].scopes;
        (void)hipEventRecord(a, s);
    }
    ~fd_prof_scope() {
        if (!ctx) return;
        (void)hipEventRecord(b, s);
        ctx->prof_events.push_back({kernel, a, b});
    }
};

inline int fd_fail(fd_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

// hipFuncSetAttribute is per device: a call site keeps one `static` mask with a bit per device ordinal and sets the attribute
// the first time it sees a device (not on every launch: it is a driver call)
inline bool fd_first_on_device(unsigned long long& mask, int device) {
    const unsigned long long bit = 1ull << (device & 63);
    if (mask & bit) return false;
    mask |= bit;
    return true;
}

inline void fd_defer(fd_ctx* ctx, int rc) {
    if (rc && ctx && !ctx->deferred_rc) ctx->deferred_rc = rc;
}
inline int fd_take_deferred(fd_ctx* ctx) {
    const int rc = ctx->deferred_rc;
    ctx->deferred_rc = 0;
    return rc;
}

#define FD_HIP(ctx, expr)                                                                     \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fd_fail((ctx), FD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                 \
                           hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

#define FD_LAUNCH_CHECK(ctx)                                                                  \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return fd_fail((ctx), FD_ERR_HIP, "kernel launch failed: %s (%s:%d)",            \
                           hipGetErrorString(_e), __FILE__, __LINE__);                        \
    } while (0)

#define FD_REQUIRE(ctx, cond, ...)                                                            \
    do {                                                                                      \
        if (!(cond)) return fd_fail((ctx), FD_ERR_ARG, __VA_ARGS__);                          \
    } while (0)

// Ensure the ctx arena holds >= bytes.  Growth frees + reallocates (device-synchronising);
// steady-state calls never allocate.
int fd_ws_reserve(fd_ctx* ctx, size_t bytes);

// Bump allocator over the arena for one API call.
struct fd_ws {
    char* base;
    size_t off = 0, cap;
    explicit fd_ws(fd_ctx* c, bool reader = false) : base((char*)c->ws), cap(c->ws_bytes) {
        if (!reader) ++c->ws_gen;          // reader = re-derives pointers of an earlier carve without writing a new layout
    }
    template <typename T>
    T* take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        T* p = (T*)(base + off);
        off += bytes;
        return p;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~size_t(255); }
};

static inline int fd_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
