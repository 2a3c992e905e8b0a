// Context, error plumbing and cached device tables (twiddles, windows) for libmmwgpu.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/mmwgpu.h"
#include "mmw_fft.h"

namespace mmw {

inline thread_local std::string g_last_error;

inline int set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define MMW_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return mmw::set_error(MMW_ERR_HIP, "%s failed: %s (%s:%d)", #call,               \
                                  hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)

#define MMW_REQUIRE(cond, ...)                                                               \
    do {                                                                                     \
        if (!(cond)) return mmw::set_error(MMW_ERR_INVALID, __VA_ARGS__);                    \
    } while (0)

#define MMW_TRY(expr)                                                                        \
    do {                                                                                     \
        int rc_ = (expr);                                                                    \
        if (rc_ != MMW_OK) return rc_;                                                       \
    } while (0)

struct ProfileSlot {
    double total_ms = 0.0;
    int launches = 0;
};

enum TableKind { TAB_TWIDDLE = 0, TAB_HANN = 1, TAB_HAMMING = 2, TAB_DFTMAT = 3 };   // DFTMAT: W_N^(jk), [N][N]

constexpr int PIPE_RING_MAX = 4;      // deepest ring of RD chunks of the overlapped chain

// tables of one chirp-z frequency list (mmw_czt.h), cached per context
struct CztPlan {
    std::vector<double> freq;
    int n_used = 0, L = 0, n_seg = 0;
    void *d_segs = nullptr, *d_tabs = nullptr;
};

// a device-synchronised chain call whose hand-off has not been checked yet (mmw_chain_settle re-runs it if it was aborted)
struct ChainCall {
    const void *d_cubes;
    void *d_out;
    int ntx, nrx, n_frames, V, S, C, A, flags, i16;
};

struct PendingSpan {
    const char *family;
    hipEvent_t e0, e1;
};

}  // namespace mmw

struct mmw_ctx {
    int device = 0;
    int num_cu = 0;
    int active_cus = 0;                         // CUs of the queue being launched on (0 = all), set by the chain
    hipStream_t stream = nullptr;
    // overlapped chain (DESIGN.md "chain schedule"): two CU-masked queues + ordering events, created lazily
    hipStream_t q_rd = nullptr, q_ang = nullptr, q_ang2 = nullptr;
    int q_rd_cus = 0;
    hipEvent_t pipe_rd[mmw::PIPE_RING_MAX] = {}, pipe_ang[mmw::PIPE_RING_MAX] = {}, pipe_begin = nullptr;
    bool pipe_ang_used[mmw::PIPE_RING_MAX] = {};
    size_t pipe_slot_bytes = 0;  // ring layout of the last overlapped chain call (a change forces a full hand-over)
    int pipe_ring = 0;
    bool pipe_unavailable = false;   // queue creation failed once: the chain stays on its serial schedule
    // device-synchronised chain (ChainSync in mmw_fft_fused.h): control block + host mirror of the monotone counters
    unsigned *chain_ctl = nullptr;
    unsigned long long chain_g = 0;          // frames handed through the ring since the layout was set up
    unsigned chain_rd_base = 0, chain_ang_base = 0;
    long chain_layout[6] = {0, 0, 0, 0, 0, 0};   // V, S*C, vskip, ring, tiles, ring base address
    bool chain_dirty = false;                // sync-mode work was enqueued since the abort word was last checked
    std::vector<mmw::ChainCall> chain_calls; // ... these calls
    int chain_fallbacks = 0;                 // calls re-run on the event schedule after a hand-off timeout
    bool chain_settling = false;             // inside mmw_chain_settle (its own entry-point calls must not recurse)
    bool rd_attr_set = false;    // hipFuncSetAttribute(max dynamic LDS) done for this context's device
    bool pipe_pending = false;   // chain work in flight on q_rd/q_ang that the context stream has not joined yet
    // overlapped detection pipeline (mmw_detect.h): producer / consumer queues on disjoint CU sets + their join events
    hipStream_t q_drd = nullptr, q_dscr = nullptr;
    int q_drd_cus = 0;
    hipEvent_t det_begin = nullptr, det_rd_done = nullptr, det_scr_done = nullptr;
    hipStream_t q_side = nullptr;               // refinement of the flagged argmax evaluations, beside the exact CFAR cells
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    // deferred tail of mmw_detect_points (the exact cells + list insertion, beside the refinement on q_side): the context stream
    // does not wait for it at the end of a call, so the range-Doppler kernel of the NEXT call runs beside it; every other entry
    // point joins it first (join_pipe), as with the chain's queues
    hipStream_t q_tail = nullptr;
    hipEvent_t tail_done = nullptr;
    hipEvent_t help_begin = nullptr, help_done = nullptr;      // the helper range-Doppler launch behind a pending tail (mmw_detect_points)
    unsigned *help_sync = nullptr;              // its ticket counter + per-frame counters (not in the scratch: the pending tail owns that)
    size_t help_sync_words = 0;
    bool tail_pending = false;
    int rd_leave_cus = 0;                       // CUs the persistent range-Doppler launch leaves free (for a tail running beside it)
    std::vector<std::pair<const char *, size_t>> tail_bufs;     // what the pending tail reads / writes (besides the scratch)
    bool det_unavailable = false;
    hipStream_t q_copy = nullptr;               // copy queue of the host-streaming API (mmw_memcpy_async), created lazily
    std::vector<void *> host_owned;             // mmw_host_alloc'ed pinned blocks still alive
    hipEvent_t t0 = nullptr, t1 = nullptr;      // mmw_timer_*
    bool profiling = false;                     // per-family kernel timing (mmw_profile_*)
    int prof_every = 1;                         // time every n-th launch group of a family (event records cost ~us)
    std::map<std::string, int> prof_seen;
    std::map<std::string, mmw::ProfileSlot> prof;
    std::vector<hipEvent_t> ev_pool;            // idle events
    std::vector<mmw::PendingSpan> ev_pending;   // recorded, not yet read back
    // (kind, N, is_double) -> device table
    std::map<std::tuple<int, int, int>, void *> tables;
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    std::vector<void *> owned;  // mmw_malloc'ed blocks still alive (freed at destroy)
    std::map<std::string, int> opts;            // mmw_diag_set_option: per-context tuning / test switches (before the environment)
    std::deque<mmw::CztPlan> czt_plans;         // (deque: pointers to cached plans survive later insertions)
    void *capon_z = nullptr;                    // exp(-j pi sin(theta_t)) of the last Capon angle grid ...
    std::vector<double> capon_key;              // ... and the grid it was built from
};

namespace mmw {

// np.hanning / np.hamming: w[i] = a + b*cos(pi*(2i-(M-1))/(M-1)), M == 1 -> 1.0
// (the reference builds its windows with these: processors/range_doppler_resp.py:62,66;
//  simple_synthetic_array_beamformer_processor_multiFrame.py:537,567)
inline double np_window(int kind, int i, int M) {
    if (M == 1) return 1.0;
    const double n = (double)(1 - M + 2 * i);
    const double c = std::cos(M_PI * n / (double)(M - 1));
    return kind == TAB_HANN ? 0.5 + 0.5 * c : 0.54 + 0.46 * c;
}

template <typename T> int get_table(mmw_ctx *ctx, int kind, int N, const void **out) {
    const auto key = std::make_tuple(kind, N, (int)(sizeof(T) == 8));
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return MMW_OK;
    }
    // DFTMAT rows are read up to 8 entries past the end by dft_level_big: zero padding
    const size_t elems = (kind == TAB_TWIDDLE) ? 2 * (size_t)N : (kind == TAB_DFTMAT ? 2 * ((size_t)N * N + 8) : (size_t)N);
    std::vector<T> h(elems, (T)0);
    if (kind == TAB_TWIDDLE || kind == TAB_DFTMAT) {
        std::vector<T> w(2 * (size_t)N);
        for (int m = 0; m < N; ++m) {
            // exact octant symmetries keep W^0, W^(N/4), ... exact
            const long double ang = -2.0L * M_PIl * (long double)m / (long double)N;
            long double c = cosl(ang), s = sinl(ang);
            if ((4 * m) % N == 0) {  // quarter turns are exact
                const int q = (4 * m) / N;
                c = (q == 0) ? 1.0L : (q == 2 ? -1.0L : 0.0L);
                s = (q == 1) ? -1.0L : (q == 3 ? 1.0L : 0.0L);
            }
            w[2 * m] = (T)c;
            w[2 * m + 1] = (T)s;
        }
        if (kind == TAB_TWIDDLE) {
            h = w;
        } else {
            for (int k = 0; k < N; ++k)
                for (int j = 0; j < N; ++j) {
                    const int m = (int)(((long)j * k) % N);
                    h[2 * ((size_t)k * N + j)] = w[2 * m];
                    h[2 * ((size_t)k * N + j) + 1] = w[2 * m + 1];
                }
        }
    } else {
        for (int i = 0; i < N; ++i) h[i] = (T)np_window(kind, i, N);
    }
    void *d = nullptr;
    if (hipMalloc(&d, elems * sizeof(T)) != hipSuccess)
        return set_error(MMW_ERR_NOMEM, "hipMalloc(%zu) for table failed", elems * sizeof(T));
    MMW_HIP(hipMemcpyAsync(d, h.data(), elems * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));  // h goes out of scope
    ctx->tables[key] = d;
    *out = d;
    return MMW_OK;
}

// Order the context stream after any overlapped-chain work still running on the two chain queues.  The
// chain does not do this itself so that back-to-back mmw_chain3d calls keep the RD || angle pipeline full;
// every other entry point that touches the context stream calls it first.
int chain_settle(mmw_ctx *ctx);     // mmwgpu.hip
inline int join_tail(mmw_ctx *ctx) {
    if (ctx->tail_pending) {
        MMW_HIP(hipStreamWaitEvent(ctx->stream, ctx->tail_done, 0));
        ctx->tail_pending = false;
    }
    return MMW_OK;
}
inline int join_pipe(mmw_ctx *ctx, bool keep_tail = false) {
    MMW_HIP(hipSetDevice(ctx->device));     // several contexts (devices) may live in one process
    if (!keep_tail) MMW_TRY(join_tail(ctx));
    if (ctx->pipe_pending) {
        for (int i = 0; i < PIPE_RING_MAX; ++i)
            if (ctx->pipe_ang_used[i]) MMW_HIP(hipStreamWaitEvent(ctx->stream, ctx->pipe_ang[i], 0));
        ctx->pipe_pending = false;
    }
    // Device-synchronised chain calls are checked (and, had their hand-off timed out, re-run) before anything else of
    // this context consumes or downloads their output: one host synchronisation behind a batch of chain calls.
    if (ctx->chain_dirty && !ctx->chain_settling) return chain_settle(ctx);
    return MMW_OK;
}
#define MMW_JOIN(ctx) MMW_TRY(mmw::join_pipe(ctx))

inline int ensure_scratch(mmw_ctx *ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return MMW_OK;
    if (ctx->scratch) {
        if (ctx->q_rd) {
            MMW_HIP(hipStreamSynchronize(ctx->q_rd));
            MMW_HIP(hipStreamSynchronize(ctx->q_ang));
            MMW_HIP(hipStreamSynchronize(ctx->q_ang2));
        }
        if (ctx->q_side) MMW_HIP(hipStreamSynchronize(ctx->q_side));
        if (ctx->q_tail) MMW_HIP(hipStreamSynchronize(ctx->q_tail));
        ctx->tail_pending = false;
        MMW_HIP(hipStreamSynchronize(ctx->stream));
        MMW_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    if (hipMalloc(&ctx->scratch, bytes) != hipSuccess)
        return set_error(MMW_ERR_NOMEM, "hipMalloc(%zu) for scratch failed", bytes);
    ctx->scratch_bytes = bytes;
    return MMW_OK;
}

// Profiling bracket: records an event pair around a launch group on the ctx stream WITHOUT a host
// sync, so the timed region is not perturbed; mmw_profile_get drains the pairs afterwards.
inline hipEvent_t take_event(mmw_ctx *ctx) {
    if (!ctx->ev_pool.empty()) {
        hipEvent_t e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

inline void drain_profile(mmw_ctx *ctx) {
    for (auto &sp : ctx->ev_pending) {
        float ms = 0.f;
        if (sp.e0 && sp.e1 && hipEventSynchronize(sp.e1) == hipSuccess &&
            hipEventElapsedTime(&ms, sp.e0, sp.e1) == hipSuccess) {
            auto &s = ctx->prof[sp.family];
            s.total_ms += ms;
            s.launches += 1;
        }
        if (sp.e0) ctx->ev_pool.push_back(sp.e0);
        if (sp.e1) ctx->ev_pool.push_back(sp.e1);
    }
    ctx->ev_pending.clear();
}

struct ProfScope {
    mmw_ctx *ctx;
    PendingSpan span;
    bool on;
    ProfScope(mmw_ctx *c, const char *f) : ctx(c), span{f, nullptr, nullptr}, on(c->profiling) {
        if (on && c->prof_every > 1) on = (c->prof_seen[f]++ % c->prof_every) == 0;
        if (!on) return;
        span.e0 = take_event(ctx);
        span.e1 = take_event(ctx);
        if (span.e0) (void)hipEventRecord(span.e0, ctx->stream);
    }
    ~ProfScope() {
        if (!on) return;
        if (span.e1) (void)hipEventRecord(span.e1, ctx->stream);
        ctx->ev_pending.push_back(span);
    }
};

// tuning knob read once per process from the environment (experiments only; defaults are the product path)
inline int tune_int(const char *name, int dflt) {
    static std::map<std::string, int> cache;
    static std::mutex mu;                       // contexts of different threads share the cache
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(name);
    if (it != cache.end()) return it->second;
    const char *s = std::getenv(name);
    const int v = (s && *s) ? std::atoi(s) : dflt;
    cache[name] = v;
    return v;
}

// per-context option (mmw_diag_set_option), else the process environment, else the default
inline int opt_int(const mmw_ctx *ctx, const char *name, int dflt) {
    if (ctx) {
        auto it = ctx->opts.find(name);
        if (it != ctx->opts.end()) return it->second;
    }
    const char *s = std::getenv(name);
    return (s && *s) ? std::atoi(s) : dflt;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(MMW_ERR_HIP, "launch %s failed: %s", what, hipGetErrorString(e));
    return MMW_OK;
}

}  // namespace mmw
