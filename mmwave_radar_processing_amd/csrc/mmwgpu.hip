// libmmwgpu.so -- extern "C" entry points declared in include/mmwgpu.h.
#include "mmw_ctx.h"
#include "mmw_launch.h"
#include "mmw_cfar.h"
#include "mmw_misc.h"
#include "mmw_czt.h"
#include "mmw_beamform.h"
#include "mmw_detect.h"
#include "mmw_cells64.h"

#include <algorithm>
#include <climits>
#include <chrono>
#include <memory>
#include <cstdlib>
#include <thread>

using namespace mmw;

namespace {

int env_int(const char *name, int dflt) {
    const char *s = std::getenv(name);
    return (s && *s) ? std::atoi(s) : dflt;
}

bool fused_rd_ok(int S, int C) { return rd_fused_supported(S, C); }

int abs_c64(mmw_ctx *ctx, const void *d_in, float *d_out, size_t n) {
    if (n == 0) return MMW_OK;
    hipLaunchKernelGGL(k_abs_c64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float2 *)d_in, d_out, n);
    return check_launch("abs_c64");
}

// Range FFT (windows on both axes folded into the load) then in-place Doppler FFT + fftshift.
// Frames go through in chunks of ~96 MB of output so that the Doppler pass finds the range
// pass's result still in the 256 MB Infinity Cache instead of re-reading it from HBM.
static int range_doppler_generic_chunk(mmw_ctx *ctx, const void *d_cubes, void *d_out, int F, int V, int S, int C) {
    FftArgs a{};
    a.in = d_cubes;
    a.out = d_out;
    a.outer = F * V;
    a.inner = C;
    a.n_in = S;
    a.in_outer_stride = a.out_outer_stride = (long)S * C;
    a.in_axis_stride = a.out_axis_stride = C;
    a.in_inner_stride = a.out_inner_stride = 1;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &a.win_axis));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, C, &a.win_inner));
    a.scale = 1.0;
    MMW_TRY((launch_fft_axis<float, float>(ctx, a, S, false)));
    FftArgs b{};
    b.in = d_out;
    b.out = d_out;  // each workgroup reads its rows completely before it writes them
    b.outer = F * V * S;
    b.inner = 1;
    b.n_in = C;
    b.in_outer_stride = b.out_outer_stride = C;
    b.in_axis_stride = b.out_axis_stride = 1;
    b.scale = 1.0;
    b.shift = 1;
    return launch_fft_axis<float, float>(ctx, b, C, true);
}

int range_doppler_generic(mmw_ctx *ctx, const void *d_cubes, void *d_out, int F, int V, int S, int C) {
    const size_t cube_bytes = (size_t)V * S * C * sizeof(float2);
    long chunk = (long)(((size_t)96 << 20) / cube_bytes);
    if (chunk < 1) chunk = 1;
    for (long f0 = 0; f0 < F; f0 += chunk) {
        const int nf = (int)((F - f0 < chunk) ? F - f0 : chunk);
        MMW_TRY(range_doppler_generic_chunk(ctx, (const char *)d_cubes + (size_t)f0 * cube_bytes,
                                            (char *)d_out + (size_t)f0 * cube_bytes, nf, V, S, C));
    }
    return MMW_OK;
}

template <typename T>
int range_profile_impl(mmw_ctx *ctx, const void *d_cubes, T *d_out, int n_frames, int V, int S, int C,
                       int chirp_idx) {
    MMW_REQUIRE(ctx && d_cubes && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0, "bad shape");
    MMW_REQUIRE(chirp_idx >= -C && chirp_idx < C, "chirp_idx %d out of range", chirp_idx);
    if (chirp_idx < 0) chirp_idx += C;  // numpy negative indexing
    if (n_frames == 0) return MMW_OK;
    MMW_TRY(ensure_scratch(ctx, (size_t)n_frames * V * S * sizeof(cplx<T>)));
    FftArgs a{};
    a.in = (const cplx<float> *)d_cubes + chirp_idx;
    a.out = ctx->scratch;
    a.outer = n_frames * V;
    a.inner = 1;
    a.n_in = S;
    a.in_outer_stride = (long)S * C;
    a.in_axis_stride = C;
    a.in_inner_stride = 1;
    a.out_outer_stride = S;
    a.out_axis_stride = 1;
    a.out_inner_stride = 1;
    MMW_TRY(get_table<T>(ctx, TAB_HANN, S, &a.win_axis));
    a.scale = 1.0;
    MMW_TRY((launch_fft_axis<T, float>(ctx, a, S, false)));
    const long n = (long)n_frames * S;
    hipLaunchKernelGGL((k_mean_abs_over_v<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const cplx<T> *)ctx->scratch, d_out, n_frames, V, S);
    return check_launch("mean_abs_over_v");
}

}  // namespace

static void sync_slot_forget(mmw_ctx *ctx);

extern "C" {

const char *mmw_version(void) { return "mmwgpu 0.3 (gfx950)"; }
int mmw_abi_version(void) { return MMWGPU_ABI_VERSION; }
const char *mmw_last_error(void) { return g_last_error.c_str(); }

int mmw_device_count(int *count) {
    MMW_REQUIRE(count, "count is null");
    MMW_HIP(hipGetDeviceCount(count));
    return MMW_OK;
}

int mmw_device_info(int device, char *name, int name_len, char *arch, int arch_len, int *num_cu,
                    size_t *total_mem) {
    hipDeviceProp_t prop;
    MMW_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        // hipDeviceProp_t::name comes back empty on this image's MI355X boxes: ask the runtime's other query, and failing that
        // name the product from what identifies it (gfx950 with 256 compute units is the MI355X / MI350X package)
        char buf[256] = {0};
        snprintf(buf, sizeof(buf), "%s", prop.name);
        if (!buf[0] && hipDeviceGetName(buf, (int)sizeof(buf), device) != hipSuccess) buf[0] = 0;
        (void)hipGetLastError();
        if (!buf[0] && !std::strncmp(prop.gcnArchName, "gfx950", 6))
            snprintf(buf, sizeof(buf), "AMD Instinct MI355X (identified by arch %s, %d CUs, %.0f GiB: the runtime reports no name)",
                     prop.gcnArchName, prop.multiProcessorCount, (double)prop.totalGlobalMem / (1 << 30));
        snprintf(name, name_len, "%s", buf);
    }
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    if (num_cu) *num_cu = prop.multiProcessorCount;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return MMW_OK;
}

int mmw_ctx_create(mmw_ctx **out, int device) {
    MMW_REQUIRE(out, "out is null");
    int n = 0;
    MMW_HIP(hipGetDeviceCount(&n));
    MMW_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    MMW_HIP(hipSetDevice(device));
    // released to the caller on success; any early return below tears the half-built context down again
    std::unique_ptr<mmw_ctx, int (*)(mmw_ctx *)> c(new mmw_ctx(), mmw_ctx_destroy);
    c->device = device;
    hipDeviceProp_t prop;
    MMW_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    MMW_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    MMW_HIP(hipEventCreate(&c->t0));
    MMW_HIP(hipEventCreate(&c->t1));
    *out = c.release();
    return MMW_OK;
}

int mmw_ctx_destroy(mmw_ctx *ctx) {
    if (!ctx) return MMW_OK;
    sync_slot_forget(ctx);
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->tables) (void)hipFree(kv.second);
    for (auto &pl : ctx->czt_plans) {
        (void)hipFree(pl.d_segs);
        (void)hipFree(pl.d_tabs);
    }
    for (void *p : ctx->owned) (void)hipFree(p);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->capon_z) (void)hipFree(ctx->capon_z);
    if (ctx->chain_ctl) (void)hipFree(ctx->chain_ctl);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    drain_profile(ctx);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->q_rd) {
        (void)hipStreamSynchronize(ctx->q_rd);
        (void)hipStreamSynchronize(ctx->q_ang);
        (void)hipStreamSynchronize(ctx->q_ang2);
        (void)hipStreamDestroy(ctx->q_rd);
        (void)hipStreamDestroy(ctx->q_ang);
        (void)hipStreamDestroy(ctx->q_ang2);
    }
    if (ctx->pipe_begin) {
        for (int i = 0; i < PIPE_RING_MAX; ++i) {
            if (ctx->pipe_rd[i]) (void)hipEventDestroy(ctx->pipe_rd[i]);
            if (ctx->pipe_ang[i]) (void)hipEventDestroy(ctx->pipe_ang[i]);
        }
        (void)hipEventDestroy(ctx->pipe_begin);
    }
    if (ctx->q_drd) {
        (void)hipStreamSynchronize(ctx->q_drd);
        (void)hipStreamSynchronize(ctx->q_dscr);
        (void)hipStreamDestroy(ctx->q_drd);
        (void)hipStreamDestroy(ctx->q_dscr);
    }
    if (ctx->q_side) {
        (void)hipStreamSynchronize(ctx->q_side);
        (void)hipStreamDestroy(ctx->q_side);
    }
    if (ctx->q_tail) {
        (void)hipStreamSynchronize(ctx->q_tail);
        (void)hipStreamDestroy(ctx->q_tail);
    }
    if (ctx->help_sync) (void)hipFree(ctx->help_sync);
    for (hipEvent_t e : {ctx->det_begin, ctx->det_rd_done, ctx->det_scr_done, ctx->side_fork, ctx->side_join, ctx->tail_done,
                         ctx->help_begin, ctx->help_done})
        if (e) (void)hipEventDestroy(e);
    if (ctx->q_copy) {
        (void)hipStreamSynchronize(ctx->q_copy);
        (void)hipStreamDestroy(ctx->q_copy);
    }
    for (void *p : ctx->host_owned) (void)hipHostFree(p);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MMW_OK;
}

int mmw_sync(mmw_ctx *ctx) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    MMW_HIP(hipStreamSynchronize(ctx->stream));     // (MMW_JOIN above has settled any device-synchronised chain call)
    return MMW_OK;
}

int mmw_malloc(mmw_ctx *ctx, void **d_ptr, size_t bytes) {
    MMW_REQUIRE(ctx && d_ptr, "null argument");
    MMW_HIP(hipSetDevice(ctx->device));
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess)
        return set_error(MMW_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    ctx->owned.push_back(p);
    *d_ptr = p;
    return MMW_OK;
}

int mmw_free(mmw_ctx *ctx, void *d_ptr) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    if (!d_ptr) return MMW_OK;
    auto it = std::find(ctx->owned.begin(), ctx->owned.end(), d_ptr);
    MMW_REQUIRE(it != ctx->owned.end(), "pointer was not allocated by this context");
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    MMW_HIP(hipFree(d_ptr));
    ctx->owned.erase(it);
    return MMW_OK;
}

int mmw_memcpy_h2d(mmw_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    MMW_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "null argument");
    MMW_JOIN(ctx);
    if (!bytes) return MMW_OK;
    MMW_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    return MMW_OK;
}

int mmw_memcpy_d2h(mmw_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    MMW_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), "null argument");
    MMW_JOIN(ctx);
    if (!bytes) return MMW_OK;
    MMW_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    return MMW_OK;
}

int mmw_memset(mmw_ctx *ctx, void *d_dst, int value, size_t bytes) {
    MMW_REQUIRE(ctx && (bytes == 0 || d_dst), "null argument");
    MMW_JOIN(ctx);
    if (!bytes) return MMW_OK;
    MMW_HIP(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return MMW_OK;
}

// ------------------------------------------------------------------ host streaming (pinned staging, copy queue, events)
// A frame loop that receives its cubes on the host (scripts/test_vel_estimation.py:145-151 in the reference) uploads chunk
// k + 1 while chunk k is processed: pinned host blocks, a second queue for the copies, events to order the two.
int mmw_host_alloc(mmw_ctx *ctx, void **h_ptr, size_t bytes) {
    MMW_REQUIRE(ctx && h_ptr, "null argument");
    MMW_HIP(hipSetDevice(ctx->device));
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
        return set_error(MMW_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    ctx->host_owned.push_back(p);
    *h_ptr = p;
    return MMW_OK;
}

int mmw_host_free(mmw_ctx *ctx, void *h_ptr) {
    MMW_REQUIRE(ctx, "ctx is null");
    if (!h_ptr) return MMW_OK;
    auto it = std::find(ctx->host_owned.begin(), ctx->host_owned.end(), h_ptr);
    MMW_REQUIRE(it != ctx->host_owned.end(), "pointer was not allocated by mmw_host_alloc of this context");
    MMW_HIP(hipSetDevice(ctx->device));
    if (ctx->q_copy) MMW_HIP(hipStreamSynchronize(ctx->q_copy));
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    MMW_HIP(hipHostFree(h_ptr));
    ctx->host_owned.erase(it);
    return MMW_OK;
}

static int queue_of(mmw_ctx *ctx, int queue, hipStream_t *out) {
    MMW_REQUIRE(queue == MMW_QUEUE_COMPUTE || queue == MMW_QUEUE_COPY, "queue must be MMW_QUEUE_COMPUTE or MMW_QUEUE_COPY");
    MMW_HIP(hipSetDevice(ctx->device));
    if (queue == MMW_QUEUE_COPY && !ctx->q_copy) MMW_HIP(hipStreamCreateWithFlags(&ctx->q_copy, hipStreamNonBlocking));
    *out = queue == MMW_QUEUE_COPY ? ctx->q_copy : ctx->stream;
    return MMW_OK;
}

int mmw_memcpy_async(mmw_ctx *ctx, void *dst, const void *src, size_t bytes, int to_host, int queue) {
    MMW_REQUIRE(ctx && (bytes == 0 || (dst && src)), "null argument");
    hipStream_t q;
    MMW_TRY(queue_of(ctx, queue, &q));
    if (queue == MMW_QUEUE_COMPUTE) MMW_JOIN(ctx);
    if (!bytes) return MMW_OK;
    MMW_HIP(hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, q));
    return MMW_OK;
}

int mmw_event_create(mmw_ctx *ctx, void **event) {
    MMW_REQUIRE(ctx && event, "null argument");
    MMW_HIP(hipSetDevice(ctx->device));
    hipEvent_t e = nullptr;
    MMW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return MMW_OK;
}

int mmw_event_destroy(mmw_ctx *ctx, void *event) {
    MMW_REQUIRE(ctx, "ctx is null");
    if (event) MMW_HIP(hipEventDestroy((hipEvent_t)event));
    return MMW_OK;
}

int mmw_event_record(mmw_ctx *ctx, void *event, int queue) {
    MMW_REQUIRE(ctx && event, "null argument");
    hipStream_t q;
    MMW_TRY(queue_of(ctx, queue, &q));
    if (queue == MMW_QUEUE_COMPUTE) MMW_JOIN(ctx);          // chain work of the context is part of "the compute queue so far"
    MMW_HIP(hipEventRecord((hipEvent_t)event, q));
    return MMW_OK;
}

int mmw_queue_wait_event(mmw_ctx *ctx, int queue, void *event) {
    MMW_REQUIRE(ctx && event, "null argument");
    hipStream_t q;
    MMW_TRY(queue_of(ctx, queue, &q));
    MMW_HIP(hipStreamWaitEvent(q, (hipEvent_t)event, 0));
    return MMW_OK;
}

int mmw_event_sync(mmw_ctx *ctx, void *event) {
    MMW_REQUIRE(ctx && event, "null argument");
    MMW_HIP(hipEventSynchronize((hipEvent_t)event));
    return MMW_OK;
}

int mmw_timer_start(mmw_ctx *ctx) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    MMW_HIP(hipEventRecord(ctx->t0, ctx->stream));
    return MMW_OK;
}

int mmw_timer_stop(mmw_ctx *ctx, float *elapsed_ms) {
    MMW_REQUIRE(ctx && elapsed_ms, "null argument");
    MMW_JOIN(ctx);
    MMW_HIP(hipEventRecord(ctx->t1, ctx->stream));
    MMW_HIP(hipEventSynchronize(ctx->t1));
    MMW_HIP(hipEventElapsedTime(elapsed_ms, ctx->t0, ctx->t1));
    return MMW_OK;
}

int mmw_profile_enable(mmw_ctx *ctx, int on) {
    MMW_REQUIRE(ctx, "ctx is null");
    ctx->profiling = on != 0;
    ctx->prof_every = on > 1 ? on : 1;      // on = n > 1: sample every n-th launch group per family
    ctx->prof_seen.clear();
    return MMW_OK;
}

int mmw_profile_reset(mmw_ctx *ctx) {
    MMW_REQUIRE(ctx, "ctx is null");
    drain_profile(ctx);
    ctx->prof.clear();
    return MMW_OK;
}

int mmw_profile_get(mmw_ctx *ctx, const char *family, float *total_ms, int *launches) {
    MMW_REQUIRE(ctx && family, "null argument");
    drain_profile(ctx);
    auto it = ctx->prof.find(family);
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.f : (float)it->second.total_ms;
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return MMW_OK;
}

// ------------------------------------------------------------------ input staging
int mmw_synth_cubes(mmw_ctx *ctx, void *d_cubes, int n_frames, int V, int S, int C, uint64_t seed0,
                    int num_targets, float noise_sigma) {
    MMW_REQUIRE(ctx && d_cubes, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0, "bad shape");
    MMW_REQUIRE(num_targets >= 0 && num_targets <= SYNTH_MAX_TARGETS, "num_targets must be 0..%d", SYNTH_MAX_TARGETS);
    const long total = (long)n_frames * V * S * C;
    if (!total) return MMW_OK;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (float2 *)d_cubes, total, V, S, C, (unsigned long long)seed0, num_targets, noise_sigma);
    return check_launch("synth");
}

int mmw_virtual_array_reformat(mmw_ctx *ctx, const void *d_raw, void *d_virt, int n_frames, int num_rx,
                               int num_tx, int S, int loops) {
    MMW_REQUIRE(ctx && d_raw && d_virt, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && num_rx > 0 && num_tx > 0 && S > 0 && loops > 0, "bad shape");
    const long total = (long)n_frames * num_rx * num_tx * S * loops;
    if (!total) return MMW_OK;
    hipLaunchKernelGGL(k_reformat, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float2 *)d_raw, (float2 *)d_virt, total, num_rx, num_tx, S, loops);
    return check_launch("reformat");
}

int mmw_virtual_array_reformat_i16(mmw_ctx *ctx, const void *d_raw_i16, void *d_virt, int n_frames, int num_rx,
                                   int num_tx, int S, int loops) {
    MMW_REQUIRE(ctx && d_raw_i16 && d_virt, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && num_rx > 0 && num_tx > 0 && S > 0 && loops > 0, "bad shape");
    const long total = (long)n_frames * num_rx * num_tx * S * loops;
    if (!total) return MMW_OK;
    hipLaunchKernelGGL(k_reformat_i16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const short2 *)d_raw_i16, (float2 *)d_virt, total, num_rx, num_tx, S, loops);
    return check_launch("reformat_i16");
}

// ------------------------------------------------------------------ FFT chain
// rv.ntx > 1: d_cubes is the raw [F][num_rx][S][num_tx * C] cube and the virtual-array de-interleave is folded into
// the load of the single-pass kernels (V == rv.ntx * rv.nrx).
static int plane_l1_impl(mmw_ctx *ctx, const void *d_cubes, float *d_l1, int n_frames, int V, int S, int C);

// d_l1 != nullptr: also the per-plane L1 norms of the windowed input (mmw_plane_l1), from inside the RD kernel where it
// can produce them, else by a pass of k_plane_l1.
static int range_doppler_impl(mmw_ctx *ctx, const void *d_cubes, void *d_out, void *d_mag_f32, int n_frames, int V,
                             int S, int C, RawView rv = RawView{1, 0}, float *d_l1 = nullptr) {
    MMW_REQUIRE(ctx && d_cubes && d_out, "null argument");
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0, "bad shape");
    if (n_frames == 0) return MMW_OK;
    {
        ProfScope ps(ctx, "rd");
        bool l1_done = false;
        // int16 raw cubes: folded into the loads of the 256 x 128 kernel and of the compile-time mixed-radix kernels (the
        // plane shapes of every shipped cfg); anything else converts + de-interleaves first (one extra pass)
        const bool i16_folded = rv.i16 && rv.ntx > 1 &&
                                ((fused_rd_ok(S, C) && !env_int("MMW_NO_FUSED_RD", 0)) ||
                                 (rd_mixed_ct_supported(S, C) && !env_int("MMW_NO_MIXED_RD", 0)));
        if (rv.i16 && !i16_folded) {
            MMW_REQUIRE(rv.ntx >= 1 && rv.nrx >= 1, "int16 cubes are raw cubes");
            const long total = (long)n_frames * V * S * C;
            hipLaunchKernelGGL(k_reformat_i16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const short2 *)d_cubes, (float2 *)d_out, total, rv.nrx, rv.ntx, S, C);
            MMW_TRY(check_launch("reformat_i16"));
            // every kernel below may run in place on a virtual-array cube: a workgroup has read all of its plane (and a
            // persistent one prefetches only planes it will write itself) before it stores
            d_cubes = d_out;
            rv = RawView{1, 0, rv.vskip, 0};
        }
        if (fused_rd_ok(S, C) && !env_int("MMW_NO_FUSED_RD", 0))
            MMW_TRY(launch_rd_fused(ctx, d_cubes, d_out, n_frames * V, S, C, rv, rv.ntx > 1 ? nullptr : d_l1, &l1_done));
        else if (rd_lds_supported(S, C) && !env_int("MMW_NO_FUSED_RD", 0) && !rd_mixed_ct_supported(S, C))      // the compile-time kernel is faster where both exist (64 x 64: 6.0 vs 4.4 TB/s)
            MMW_TRY(launch_rd_lds(ctx, d_cubes, d_out, n_frames * V, S, C, rv));
        else if (rd_mixed_ct_supported(S, C) && !env_int("MMW_NO_MIXED_RD", 0)) {
            // compile-time kernel; on virtual-array cubes it also leaves the planes' L1 norms when asked
            float *l1 = rv.ntx > 1 ? nullptr : d_l1;
            MMW_TRY(launch_rd_mixed_ct(ctx, d_cubes, (long)S * C, d_out, n_frames * V, S, C, rv, nullptr, 0, nullptr, false, l1));
            l1_done = l1 != nullptr;
        } else if (rd_mixed_supported(S, C) && !env_int("MMW_NO_MIXED_RD", 0))
            MMW_TRY((launch_rd_mixed<float, false>(ctx, d_cubes, (long)S * C, d_out, n_frames * V, S, C, rv)));
        else if (rv.ntx <= 1 && rd_split_ct_supported(S, C) && !env_int("MMW_NO_SPLIT_RD", 0))
            MMW_TRY(launch_rd_split_ct(ctx, d_cubes, d_out, n_frames * V, S, C, rv));
        else if (rv.ntx > 1) {
            // no single-pass kernel for this plane: de-interleave into the output buffer, then transform it in place
            const long total = (long)n_frames * V * S * C;
            hipLaunchKernelGGL(k_reformat, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const float2 *)d_cubes, (float2 *)d_out, total, rv.nrx, rv.ntx, S, C);
            MMW_TRY(check_launch("reformat"));
            // (the split kernel reads its whole plane before its first store, so it may run in place as well)
            if (rd_split_ct_supported(S, C) && !env_int("MMW_NO_SPLIT_RD", 0))
                MMW_TRY(launch_rd_split_ct(ctx, d_out, d_out, n_frames * V, S, C, RawView{1, 0, rv.vskip}));
            else
                MMW_TRY(range_doppler_generic(ctx, d_out, d_out, n_frames, V, S, C));
        } else
            MMW_TRY(range_doppler_generic(ctx, d_cubes, d_out, n_frames, V, S, C));
        if (d_l1 && !l1_done) {
            MMW_REQUIRE(rv.ntx <= 1, "plane L1 norms are defined on virtual-array cubes");
            MMW_TRY(plane_l1_impl(ctx, d_cubes, d_l1, n_frames, V, S, C));
        }
    }
    if (d_mag_f32) MMW_TRY(abs_c64(ctx, d_out, (float *)d_mag_f32, (size_t)n_frames * V * S * C));
    return MMW_OK;
}

int mmw_range_doppler(mmw_ctx *ctx, const void *d_cubes, void *d_out, void *d_mag_f32, int n_frames, int V,
                      int S, int C) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    return range_doppler_impl(ctx, d_cubes, d_out, d_mag_f32, n_frames, V, S, C);
}

static int range_doppler_mag64_impl(mmw_ctx *ctx, const void *d_cubes, double *d_mag, int n_frames, int V, int S,
                                   int C, int rx_idx) {
    MMW_REQUIRE(ctx && d_cubes && d_mag, "null argument");
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && rx_idx >= 0 && rx_idx < V, "bad shape / rx_idx");
    if (n_frames == 0) return MMW_OK;
    ProfScope ps(ctx, "rd64");
    if (S == R64_S && C == R64_C && n_frames >= 8 && opt_int(ctx, "MMW_RD64_FUSED", 1)) {
        // one launch, the complex128 intermediate in per-workgroup scratch that never leaves the caches (mmw_cells64.h); a whole
        // frame per workgroup: batches only (a single frame is spread over the chip by the two generic launches)
        const int grid = std::min(n_frames, ctx->num_cu);
        MMW_TRY(ensure_scratch(ctx, (size_t)grid * S * C * sizeof(cplx<double>)));
        Rd64Args ra{};
        ra.cubes = (const float2 *)d_cubes + (long)rx_idx * S * C;
        ra.frame_stride = (long)V * S * C;
        ra.mag = d_mag;
        ra.scratch = (cplx<double> *)ctx->scratch;
        ra.n_frames = n_frames;
        const void *p;
        MMW_TRY(get_table<double>(ctx, TAB_HANN, S, &p));
        ra.ws = (const double *)p;
        MMW_TRY(get_table<double>(ctx, TAB_HANN, C, &p));
        ra.wc = (const double *)p;
        MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, S, &p));
        ra.twS = (const cplx<double> *)p;
        MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, C, &p));
        ra.twC = (const cplx<double> *)p;
        const size_t lds = rd64_lds();
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rd_mag64_256x128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_rd_mag64_256x128, dim3((unsigned)grid), dim3(C64_NT), lds, ctx->stream, ra);
        return check_launch("rd_mag64_256x128");
    }
    RdMixedPlan mp;
    // LDS-resident single-pass kernel for every plane that fits in float64 except small power-of-two ones, where the
    // two-kernel register-FFT path is as fast (64 x 64, 512 x 8: equal; 128 x 64, 256 x 32: single pass +18 %)
    if (!is_pow2(S) || !is_pow2(C) || (long)S * C >= 8192)
        if (rd_mixed_plan(S, C, sizeof(cplx<double>), &mp) && !env_int("MMW_NO_MIXED_RD", 0))
            return launch_rd_mixed<double, true>(ctx, (const cplx<float> *)d_cubes + (long)rx_idx * S * C,
                                                 (long)V * S * C, d_mag, n_frames, S, C);
    // two passes with a complex128 intermediate: chunks of frames small enough for the intermediate to stay in the
    // Infinity Cache between them (128 MB; it also bounds the scratch: all 1250 frames of the bench at once were 640 MB)
    const size_t plane_bytes = (size_t)S * C * sizeof(cplx<double>);
    long chunk = (long)(((size_t)128 << 20) / plane_bytes);
    if (chunk < 1) chunk = 1;
    if (chunk > n_frames) chunk = n_frames;
    MMW_TRY(ensure_scratch(ctx, (size_t)chunk * plane_bytes));
    FftArgs a{};
    a.out = ctx->scratch;
    a.inner = C;
    a.n_in = S;
    a.in_outer_stride = (long)V * S * C;
    a.out_outer_stride = (long)S * C;
    a.in_axis_stride = a.out_axis_stride = C;
    a.in_inner_stride = a.out_inner_stride = 1;
    MMW_TRY(get_table<double>(ctx, TAB_HANN, S, &a.win_axis));
    MMW_TRY(get_table<double>(ctx, TAB_HANN, C, &a.win_inner));
    a.scale = 1.0;
    FftArgs b{};
    b.in = ctx->scratch;
    b.inner = 1;
    b.n_in = C;
    b.in_outer_stride = b.out_outer_stride = C;
    b.in_axis_stride = b.out_axis_stride = 1;
    b.scale = 1.0;
    b.shift = 1;
    b.magnitude = 1;
    for (long f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = (int)std::min<long>(chunk, n_frames - f0);
        a.in = (const cplx<float> *)d_cubes + (f0 * V + rx_idx) * (long)S * C;
        a.outer = nf;
        MMW_TRY((launch_fft_axis<double, float>(ctx, a, S, false)));
        b.out = d_mag + f0 * (long)S * C;
        b.outer = nf * S;
        MMW_TRY((launch_fft_axis<double, double>(ctx, b, C, true)));
    }
    return MMW_OK;
}

int mmw_range_doppler_mag64(mmw_ctx *ctx, const void *d_cubes, double *d_mag, int n_frames, int V, int S,
                            int C, int rx_idx) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    return range_doppler_mag64_impl(ctx, d_cubes, d_mag, n_frames, V, S, C, rx_idx);
}

static bool angle_fast_path(int V, long bins, int A, bool mag);

static int angle_fft_impl(mmw_ctx *ctx, const void *d_rd, void *d_out, int n_frames, int V, int S, int C, int A,
                         int flags) {
    MMW_REQUIRE(ctx && d_rd && d_out, "null argument");
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && A >= V, "bad shape (need A >= V)");
    MMW_REQUIRE((flags & ~7) == 0, "unknown flag bits %d", flags);
    if (n_frames == 0) return MMW_OK;
    const bool magnitude = flags & MMW_ANGLE_MAGNITUDE, window = !(flags & MMW_ANGLE_NO_WINDOW),
               shift = !(flags & MMW_ANGLE_NO_SHIFT);
    ProfScope ps(ctx, "angle");
    const long bins = (long)S * C;
    if (n_frames <= 65535 && angle_fast_path(V, bins, A, magnitude)) {
        float h[16];
        for (int i = 0; i < V; ++i) h[i] = window ? (float)np_window(TAB_HANN, i, V) : 1.f;
        switch (V) {
            case 4: return launch_angle64<4>(ctx, d_rd, d_out, n_frames, bins, magnitude, h, shift);
            case 8: return launch_angle64<8>(ctx, d_rd, d_out, n_frames, bins, magnitude, h, shift);
            case 12: return launch_angle64<12>(ctx, d_rd, d_out, n_frames, bins, magnitude, h, shift);
            default: return launch_angle64<16>(ctx, d_rd, d_out, n_frames, bins, magnitude, h, shift);
        }
    }
    FftArgs a{};
    a.in = d_rd;
    a.out = d_out;
    a.outer = n_frames;
    a.inner = (int)bins;
    a.n_in = V;
    a.in_outer_stride = (long)V * bins;
    a.out_outer_stride = (long)A * bins;
    a.in_axis_stride = a.out_axis_stride = bins;
    a.in_inner_stride = a.out_inner_stride = 1;
    if (window) MMW_TRY(get_table<float>(ctx, TAB_HANN, V, &a.win_axis));
    a.scale = 1.0;
    a.shift = shift;
    a.magnitude = magnitude;
    return launch_fft_axis<float, float>(ctx, a, A, false);
}

int mmw_angle_fft(mmw_ctx *ctx, const void *d_rd, void *d_out, int n_frames, int V, int S, int C, int A,
                  int flags) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    return angle_fft_impl(ctx, d_rd, d_out, n_frames, V, S, C, A, flags);
}

int mmw_dbs_gather(mmw_ctx *ctx, const float *d_mag, const int *h_ang_idx, const int *h_vel_idx, float *d_out,
                   int n_frames, int A, int S, int C, int n_out) {
    MMW_REQUIRE(ctx && d_mag && h_ang_idx && h_vel_idx && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && A > 0 && S > 0 && C > 0 && n_out > 0, "bad shape");
    for (int i = 0; i < n_out; ++i)
        MMW_REQUIRE(h_ang_idx[i] >= 0 && h_ang_idx[i] < A && h_vel_idx[i] >= 0 && h_vel_idx[i] < C,
                    "gather index %d out of range", i);
    if (n_frames == 0) return MMW_OK;
    MMW_TRY(ensure_scratch(ctx, (size_t)2 * n_out * sizeof(int)));
    int *d_idx = (int *)ctx->scratch;
    MMW_HIP(hipMemcpyAsync(d_idx, h_ang_idx, (size_t)n_out * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipMemcpyAsync(d_idx + n_out, h_vel_idx, (size_t)n_out * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));     // the index arrays are caller-owned host memory
    const long total = (long)n_frames * S * n_out;
    hipLaunchKernelGGL(k_dbs_gather, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_mag, d_idx,
                       d_idx + n_out, d_out, n_frames, A, S, C, n_out);
    return check_launch("dbs_gather");
}

int mmw_mean_over_range(mmw_ctx *ctx, const float *d_mag, float *d_out, int n_frames, int A, int S, int C,
                        int s_lo, int s_hi) {
    MMW_REQUIRE(ctx && d_mag && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && A > 0 && S > 0 && C > 0, "bad shape");
    MMW_REQUIRE(0 <= s_lo && s_lo < s_hi && s_hi <= S, "empty or out-of-range row interval [%d, %d)", s_lo, s_hi);
    const long total = (long)n_frames * A * C;
    if (!total) return MMW_OK;
    hipLaunchKernelGGL(k_mean_over_range, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_mag, d_out,
                       n_frames, A, S, C, s_lo, s_hi);
    return check_launch("mean_over_range");
}

// |angle FFT| averaged over the range rows [s_lo, s_hi): d_rd [nf][V][S][C] c64 -> d_out [nf][C][A] float32.
// Fused single pass (k_angle64_rmean) when the angle fast path applies, else |angle FFT| cube + k_mean_over_range.
// d_work: scratch of angle_mean_work_bytes(...) bytes.
static size_t angle_mean_work_bytes(int nf, int V, int S, int C, int A, int rows) {
    const bool fused = A == 64 && (V == 4 || V == 8 || V == 12 || V == 16) && !env_int("MMW_NO_FUSED_ANGLE", 0);
    if (fused) return (size_t)nf * rmean_partitions(nf, S, C, rows) * RMEAN_RL * 64 * C * sizeof(float);
    return (size_t)nf * A * S * C * sizeof(float);
}

static int angle_mean_impl(mmw_ctx *ctx, const void *d_rd, void *d_work, size_t work_bytes, float *d_out, int nf, int V,
                           int S, int C, int A, int s_lo, int s_hi, int flags) {
    const bool window = !(flags & MMW_ANGLE_NO_WINDOW), shift = !(flags & MMW_ANGLE_NO_SHIFT);
    const bool fused = A == 64 && (V == 4 || V == 8 || V == 12 || V == 16) && !env_int("MMW_NO_FUSED_ANGLE", 0);
    if (fused) {
        ProfScope ps(ctx, "angle_rmean");
        float h[16];
        for (int i = 0; i < V; ++i) h[i] = window ? (float)np_window(TAB_HANN, i, V) : 1.f;
        switch (V) {
            case 4: return launch_angle64_rmean<4>(ctx, d_rd, (float *)d_work, work_bytes, d_out, nf, S, C, s_lo, s_hi, h, shift);
            case 8: return launch_angle64_rmean<8>(ctx, d_rd, (float *)d_work, work_bytes, d_out, nf, S, C, s_lo, s_hi, h, shift);
            case 12: return launch_angle64_rmean<12>(ctx, d_rd, (float *)d_work, work_bytes, d_out, nf, S, C, s_lo, s_hi, h, shift);
            default: return launch_angle64_rmean<16>(ctx, d_rd, (float *)d_work, work_bytes, d_out, nf, S, C, s_lo, s_hi, h, shift);
        }
    }
    MMW_TRY(angle_fft_impl(ctx, d_rd, d_work, nf, V, S, C, A, (flags & (MMW_ANGLE_NO_WINDOW | MMW_ANGLE_NO_SHIFT)) | MMW_ANGLE_MAGNITUDE));
    const long total = (long)nf * A * C;
    hipLaunchKernelGGL(k_mean_over_range, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float *)d_work, d_out, nf, A, S, C, s_lo, s_hi);
    return check_launch("mean_over_range");
}

int mmw_doppler_azimuth(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C, int A,
                        int s_lo, int s_hi, int flags) {
    MMW_REQUIRE(ctx && d_cubes && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && A >= V, "bad shape (need A >= V)");
    MMW_REQUIRE(0 <= s_lo && s_lo < s_hi && s_hi <= S, "empty or out-of-range row interval [%d, %d)", s_lo, s_hi);
    MMW_REQUIRE((flags & ~(MMW_ANGLE_NO_WINDOW | MMW_ANGLE_NO_SHIFT)) == 0, "unknown flag bits %d", flags);
    if (n_frames == 0) return MMW_OK;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t cube_bytes = (size_t)V * S * C * sizeof(float2);
    long chunk = std::min<long>(n_frames, 65535);
    while (chunk > 1 && up(chunk * cube_bytes) + angle_mean_work_bytes((int)chunk, V, S, C, A, s_hi - s_lo) > ((size_t)1 << 30))
        chunk = (chunk + 1) / 2;
    const size_t work_bytes = angle_mean_work_bytes((int)chunk, V, S, C, A, s_hi - s_lo);
    MMW_TRY(ensure_scratch(ctx, up(chunk * cube_bytes) + work_bytes));
    char *d_rd = (char *)ctx->scratch, *d_work = d_rd + up(chunk * cube_bytes);
    for (long f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = (int)std::min<long>(chunk, n_frames - f0);
        MMW_TRY(range_doppler_impl(ctx, (const char *)d_cubes + (size_t)f0 * cube_bytes, d_rd, nullptr, nf, V, S, C));
        MMW_TRY(angle_mean_impl(ctx, d_rd, d_work, work_bytes, d_out + (size_t)f0 * C * A, nf, V, S, C, A, s_lo, s_hi, flags));
    }
    return MMW_OK;
}

int mmw_doppler_azimuth_zoom(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C, int A,
                             int s_lo, int s_hi, int n_used, const double *h_freq, int M, int flags) {
    MMW_REQUIRE(ctx && d_cubes && d_out && h_freq, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && A >= V && M > 0, "bad shape (need A >= V, M > 0)");
    MMW_REQUIRE(0 <= s_lo && s_lo < s_hi && s_hi <= S, "empty or out-of-range row interval [%d, %d)", s_lo, s_hi);
    MMW_REQUIRE(n_used > 0 && n_used <= C, "zoom transform defined for %d chirps, the cube has %d", n_used, C);
    MMW_REQUIRE((flags & ~(MMW_ANGLE_NO_WINDOW | MMW_ANGLE_NO_SHIFT)) == 0, "unknown flag bits %d", flags);
    constexpr int RB = 16;
    MMW_REQUIRE((size_t)RB * n_used * sizeof(float2) <= 64 * 1024, "too many chirps for the zoom row buffer");
    if (n_frames == 0) return MMW_OK;
    ProfScope ps(ctx, "dopaz_zoom");
    const int Sk = s_hi - s_lo;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_freq = up((size_t)M * sizeof(double)), b_tab = up((size_t)n_used * M * sizeof(float2));
    const size_t f_rng = (size_t)V * S * C * sizeof(float2), f_zoom = (size_t)V * Sk * M * sizeof(float2);
    long chunk = std::min<long>(n_frames, 65535);
    auto need = [&](long c) { return up(c * f_rng) + up(c * f_zoom) + up(angle_mean_work_bytes((int)c, V, Sk, M, A, Sk)); };
    while (chunk > 1 && need(chunk) > ((size_t)1 << 30)) chunk = (chunk + 1) / 2;    // <= 1 GiB of intermediates per pass
    MMW_TRY(ensure_scratch(ctx, b_freq + b_tab + need(chunk)));
    char *base = (char *)ctx->scratch;
    double *d_freq = (double *)base;
    float2 *d_tab = (float2 *)(base + b_freq);
    char *d_rng = base + b_freq + b_tab;
    char *d_zoom = d_rng + up((size_t)chunk * f_rng);
    char *d_mag = d_zoom + up((size_t)chunk * f_zoom);
    // chirp-z form (mmw_czt.h) unless the list has no usable plan; MMW_ZOOM_DIRECT=1 keeps the direct n_used x M table
    const CztPlan *plan = nullptr;
    if (czt_length(n_used) > 0 && !env_int("MMW_ZOOM_DIRECT", 0)) MMW_TRY(czt_plan(ctx, h_freq, M, n_used, &plan));
    if (!plan) {
        MMW_HIP(hipMemcpyAsync(d_freq, h_freq, (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        MMW_HIP(hipStreamSynchronize(ctx->stream));     // h_freq is caller-owned host memory
        const long tab_n = (long)n_used * M;
        hipLaunchKernelGGL(k_zoom_table, dim3((unsigned)((tab_n + 255) / 256)), dim3(256), 0, ctx->stream, d_freq, d_tab,
                           n_used, M);
        MMW_TRY(check_launch("zoom_table"));
    }
    for (long f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = (int)((n_frames - f0 < chunk) ? n_frames - f0 : chunk);
        FftArgs a{};                                 // range FFT, Hann(S) x Hann(C) folded into the load
        a.in = (const char *)d_cubes + (size_t)f0 * f_rng;
        a.out = d_rng;
        a.outer = nf * V;
        a.inner = C;
        a.n_in = S;
        a.in_outer_stride = a.out_outer_stride = (long)S * C;
        a.in_axis_stride = a.out_axis_stride = C;
        a.in_inner_stride = a.out_inner_stride = 1;
        MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &a.win_axis));
        MMW_TRY(get_table<float>(ctx, TAB_HANN, C, &a.win_inner));
        a.scale = 1.0;
        MMW_TRY((launch_fft_axis<float, float>(ctx, a, S, false)));
        const long rows = (long)nf * V * Sk;
        if (plan) {
            CztArgs z{};
            z.x = (const float2 *)d_rng;
            z.outer_stride = (long)S * C;
            z.inner_stride = C;
            z.elem_stride = 1;
            z.s_lo = s_lo;
            z.s_keep = Sk;
            z.M = M;
            z.rows = rows;
            z.out = (float2 *)d_zoom;
            MMW_TRY(launch_czt_rows(ctx, *plan, z));
        } else {
            hipLaunchKernelGGL((k_zoom_rows<RB>), dim3((unsigned)((rows + RB - 1) / RB), (unsigned)((M + 255) / 256)), dim3(256),
                               (size_t)RB * n_used * sizeof(float2), ctx->stream, (const float2 *)d_rng, d_tab,
                               (float2 *)d_zoom, S, C, s_lo, Sk, n_used, M, rows);
            MMW_TRY(check_launch("zoom_rows"));
        }
        // antenna window + zero-padded angle FFT + |.| on [nf][V][Sk][M] and the mean over the kept range bins
        MMW_TRY(angle_mean_impl(ctx, d_zoom, d_mag, up(angle_mean_work_bytes((int)chunk, V, Sk, M, A, Sk)), d_out + (size_t)f0 * M * A, nf, V, Sk, M,
                                A, 0, Sk, flags));
    }
    return MMW_OK;
}

// Lazily create the two CU-masked queues of the overlapped chain: the RD queue owns the first rd_cus
// CU-mask bits, the angle queue the rest, so workgroups of the two kernels are co-resident on the chip.
// rd_cus == 0: no masks (both queues may use every CU).  A failure leaves the context without chain queues
// (pipe_unavailable) and the chain falls back to its serial schedule.
static int ensure_pipe_queues(mmw_ctx *ctx, int rd_cus) {
    if (ctx->pipe_unavailable) return set_error(MMW_ERR_UNSUPPORTED, "chain queues unavailable on this runtime");
    if (ctx->q_rd && ctx->q_rd_cus == rd_cus) return MMW_OK;
    if (ctx->q_rd) {
        MMW_HIP(hipStreamSynchronize(ctx->q_rd));
        MMW_HIP(hipStreamSynchronize(ctx->q_ang));
        MMW_HIP(hipStreamSynchronize(ctx->q_ang2));
        MMW_HIP(hipStreamDestroy(ctx->q_rd));
        MMW_HIP(hipStreamDestroy(ctx->q_ang));
        MMW_HIP(hipStreamDestroy(ctx->q_ang2));
        ctx->q_rd = ctx->q_ang = ctx->q_ang2 = nullptr;
    }
    if (!ctx->pipe_begin) {                         // events are created once per context
        for (int i = 0; i < PIPE_RING_MAX; ++i) {
            MMW_HIP(hipEventCreateWithFlags(&ctx->pipe_rd[i], hipEventDisableTiming));
            MMW_HIP(hipEventCreateWithFlags(&ctx->pipe_ang[i], hipEventDisableTiming));
        }
        MMW_HIP(hipEventCreateWithFlags(&ctx->pipe_begin, hipEventDisableTiming));
    }
    // Two angle queues on the same CUs take alternate chunks: the barrier / marker packets around one launch are
    // processed while the other queue's kernel runs (one queue alone idles ~15 us per launch).
    hipError_t e1, e2 = hipSuccess, e3 = hipSuccess;
    if (rd_cus > 0) {
        const int words = (ctx->num_cu + 31) / 32;
        std::vector<uint32_t> m_rd(words, 0u), m_ang(words, 0u);
        for (int i = 0; i < ctx->num_cu; ++i) ((i < rd_cus) ? m_rd : m_ang)[i / 32] |= 1u << (i % 32);
        e1 = hipExtStreamCreateWithCUMask(&ctx->q_rd, (uint32_t)words, m_rd.data());
        if (e1 == hipSuccess) e2 = hipExtStreamCreateWithCUMask(&ctx->q_ang, (uint32_t)words, m_ang.data());
        if (e1 == hipSuccess && e2 == hipSuccess) e3 = hipExtStreamCreateWithCUMask(&ctx->q_ang2, (uint32_t)words, m_ang.data());
    } else {
        e1 = hipStreamCreateWithFlags(&ctx->q_rd, hipStreamNonBlocking);
        if (e1 == hipSuccess) e2 = hipStreamCreateWithFlags(&ctx->q_ang, hipStreamNonBlocking);
        if (e1 == hipSuccess && e2 == hipSuccess) e3 = hipStreamCreateWithFlags(&ctx->q_ang2, hipStreamNonBlocking);
    }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        if (ctx->q_rd) (void)hipStreamDestroy(ctx->q_rd);
        if (ctx->q_ang) (void)hipStreamDestroy(ctx->q_ang);
        ctx->q_rd = ctx->q_ang = ctx->q_ang2 = nullptr;
        ctx->pipe_unavailable = true;
        (void)hipGetLastError();
        return set_error(MMW_ERR_UNSUPPORTED, "chain queue creation failed: %s",
                         hipGetErrorString(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3)));
    }
    ctx->q_rd_cus = rd_cus;
    for (int i = 0; i < PIPE_RING_MAX; ++i) ctx->pipe_ang_used[i] = false;   // the old queues were drained above
    return MMW_OK;
}

// Does the angle stage of this call run k_angle64's ZE variant (planes 0 and V-1 never loaded)?
static bool angle_fast_path(int V, long bins, int A, bool mag) {      // per launch of at most 65535 frames
    // odd bin counts: complex output only (k_angle64_rows_odd)
    return A == 64 && (bins % 2 == 0 || !mag) && !env_int("MMW_NO_FUSED_ANGLE", 0) && (V == 4 || V == 8 || V == 12 || V == 16);
}
static bool angle_skips_end_planes(const mmw_ctx *ctx, int V, long bins, int A, int flags) {
    return angle_fast_path(V, bins, A, (flags & MMW_ANGLE_MAGNITUDE) != 0) && V > 2 && !(flags & MMW_ANGLE_NO_WINDOW) &&
           np_window(TAB_HANN, 0, V) == 0.0 && np_window(TAB_HANN, V - 1, V) == 0.0 && opt_int(ctx, "MMW_ANGLE_ZE", 1) != 0;
}

// Schedule of one chain call (DESIGN.md "chain schedule"), also reported by mmw_diag_chain_plan.
//  serial    : RD then angle over chunks of frames on the context stream.
//  overlapped: RD(k+1) on a queue masked to rd_cus CUs runs beside angle(k) on the remaining CUs.  RD is
//              LDS/ALU/latency bound, angle is HBM-write bound (a CU sustains ~44 GB/s of stores, so it needs
//              about half the chip), and the ring of RD chunks stays resident in the 256 MB Infinity
//              Cache, so the RD->angle intermediate never goes to HBM.  Default for batches of the fused
//              shape; MMW_CHAIN_PIPELINE=0/1 forces a schedule.
struct ChainPlan {
    bool pipelined;
    int chunk, ring, rd_cus, vskip;
    bool sync;              // device-synchronised form: one RD launch + one angle launch per call (ChainSync)
    int ring_frames;        // sync: frames in the ring of RD cubes
};
static ChainPlan chain_plan(const mmw_ctx *ctx, bool keep_rd, bool raw, int n_frames, int V, int S, int C, int A, int flags,
                            bool allow_sync = true, bool i16 = false) {
    ChainPlan p{};
    // Planes nobody reads are not transformed: with the Hann(V) antenna window the end antennas have weight exactly 0
    // (np.hanning end points) and k_angle64's ZE variant never loads them, so when the RD cube is only an internal
    // intermediate (d_rd == NULL) their range-Doppler transform is skipped: 1/6 of the RD work at V = 12.
    // (Non-finite samples in an end antenna: the reference's 0 * inf gives NaN everywhere, this path stays finite.)
    p.vskip = (!keep_rd && angle_skips_end_planes(ctx, V, (long)S * C, A, flags) && opt_int(ctx, "MMW_CHAIN_SKIP_ENDS", 1)) ? V : 0;
    const int v_live = p.vskip > 2 ? V - 2 : V;
    // CU split: half the chip each.  The angle stage is bound by HBM writes and needs ~128 CUs to saturate them (a CU
    // sustains ~44 GB/s of stores); the range-Doppler stage keeps up from 112 CUs on (sweep in DESIGN.md)
    // Mixed-radix planes above 80 KB leave ONE range-Doppler workgroup per CU and are arithmetic bound: 5/8 of the chip
    // (tools/chain_modes.sh: 254 x 50, 100 x 100, 120 x 126 are 6-25 % faster at 160 than at 128 CUs; the angle stage
    // still saturates its store stream from the other 96).
    // Planes with a 127-point level are arithmetic bound wherever they run (the level is a dense real matrix product on
    // float32 MFMAs, which share the vector ALUs): 5/8 as well at 63 x 127 (+4 %), 3/4 at 254 x 50 (one 104 KB plane per
    // CU: 4.06 against 3.82 TB/s at 5/8; sweep of 128..208 CUs in profiles/r03_chain_rd_cus.log).
    const size_t plane_lds = (size_t)S * (C | 1) * sizeof(cplx<float>);
    const bool big_prime = S % 127 == 0 || C % 127 == 0;
    const bool heavy_rd = !fused_rd_ok(S, C) && (plane_lds > 80 * 1024 || big_prime);
    const int rd_cus_dflt = !heavy_rd ? ctx->num_cu / 2 : (big_prime && plane_lds > 100 * 1024) ? ctx->num_cu * 3 / 4 : ctx->num_cu * 5 / 8;
    p.rd_cus = opt_int(ctx, "MMW_RD_CUS", rd_cus_dflt);
    if (p.rd_cus < 0 || p.rd_cus >= ctx->num_cu) p.rd_cus = rd_cus_dflt;         // 0: unmasked queues
    p.ring = std::max(2, std::min(opt_int(ctx, "MMW_CHAIN_RING", 3), (int)PIPE_RING_MAX));
    // chunk: whole RD waves (rd_cus planes each) and `ring` chunks of live RD planes within ~250 MB of cache
    const size_t live_bytes = (size_t)v_live * S * C * sizeof(cplx<float>);
    int chunk_auto = (int)((250u << 20) / ((size_t)p.ring * live_bytes));
    {
        const int wave_cus = p.rd_cus > 0 ? p.rd_cus : ctx->num_cu;
        const int waves = (int)((long)chunk_auto * v_live / wave_cus);
        if (waves >= 1) chunk_auto = (int)((long)waves * wave_cus / v_live);
    }
    if (chunk_auto < 1) chunk_auto = 1;
    const bool fused_shape = (fused_rd_ok(S, C) || rd_lds_supported(S, C) || rd_mixed_supported(S, C)) &&
                             angle_fast_path(V, (long)S * C, A, (flags & MMW_ANGLE_MAGNITUDE) != 0);   // both stages have a single-pass kernel
    // (the split kernel of the 32768-cell planes is latency bound at 2 waves per SIMD: on half the chip it is slower than
    //  the serial schedule -- 5.6 vs 5.3 us/frame at 12 x 512 x 64 -- so those planes stay serial)
    const int want_pipe = opt_int(ctx, "MMW_CHAIN_PIPELINE", -1);
    p.pipelined = !keep_rd && !ctx->pipe_unavailable &&
                  (want_pipe == 1 || (want_pipe == -1 && fused_shape && n_frames >= 2 * chunk_auto));
    p.chunk = opt_int(ctx, "MMW_CHAIN_CHUNK", p.pipelined ? chunk_auto : 1024);
    p.chunk = std::max(1, std::min(p.chunk, std::min(n_frames, 65535)));          // 65535: grid.y of the angle kernel
    // Device-synchronised form (256 x 128 planes): needs the two disjoint CU sets, because persistent angle workgroups
    // that filled every CU would keep the range-Doppler workgroups they wait for from ever becoming resident.
    const char *mode = std::getenv("MMW_CHAIN_MODE");
    // (windowed chains only: the un-windowed angle variants of the persistent kernel exceed its register budget)
    // raw cubes: RAWIN variant of the 256 x 128 kernel, MODE 3 of the compile-time mixed-radix ones
    // (int16 raw cubes: the 256 x 128 producer only)
    const bool sync_shape = fused_rd_ok(S, C) || (!i16 && (raw ? rd_mixed_ct_raw_sync_supported(S, C) : rd_mixed_ct_supported(S, C)));
    p.sync = allow_sync && p.pipelined && sync_shape && p.rd_cus > 0 && p.vskip > 2 && !(mode && !std::strcmp(mode, "events"));
    // ring: ~120 MB of live planes (48 frames at 10 x 256 x 128): the ring and the streaming traffic around it share the
    // 256 MB Infinity Cache; 40-64 frames measured equal, 96 was 8 % slower
    p.ring_frames = (int)((120u << 20) / live_bytes);
    p.ring_frames = std::max(2, std::min(p.ring_frames, (int)CTL_RING_MAX));
    while ((size_t)p.ring_frames * V * S * C * sizeof(cplx<float>) >= ((size_t)1 << 31)) --p.ring_frames;   // 32-bit buffer offsets
    if (p.sync) p.chunk = n_frames;
    return p;
}

// The device-synchronised chain keeps two persistent kernels resident on disjoint halves of the chip.  Two contexts of
// ONE process doing that on the same device could starve each other (each one's consumer holding the slots the other's
// producer needs) until the bounded spins give up, so per device only one context at a time may have such work in
// flight; a second context takes the event schedule for that call.  (Two PROCESSES on one device cannot see each other:
// INTEGRATION.md.)
static std::mutex g_sync_mu;
static mmw_ctx *g_sync_owner[64] = {};
static bool g_sync_launching[64] = {};      // the owner is between acquire and the recording of its completion events
static bool sync_slot_acquire(mmw_ctx *ctx) {
    if (ctx->device < 0 || ctx->device >= 64) return true;
    std::lock_guard<std::mutex> lock(g_sync_mu);
    mmw_ctx *&owner = g_sync_owner[ctx->device];
    if (owner && owner != ctx) {
        // still launching, or still running?  (its last device-synchronised call recorded pipe_ang[0] / [1] behind the two
        // launches, under this mutex: sync_slot_launched)
        bool busy = g_sync_launching[ctx->device];
        for (int i = 0; i < 2 && !busy; ++i)
            if (owner->pipe_ang_used[i] && owner->pipe_ang[i] && hipEventQuery(owner->pipe_ang[i]) == hipErrorNotReady) busy = true;
        (void)hipGetLastError();
        if (busy) return false;
    }
    owner = ctx;
    g_sync_launching[ctx->device] = true;
    if (const int ms = opt_int(ctx, "MMW_SYNC_SLOT_TEST_SLEEP_MS", 0)) {      // test hook: widen the window between acquire and launch
        g_sync_mu.unlock();
        std::this_thread::sleep_for(std::chrono::milliseconds(ms));
        g_sync_mu.lock();
    }
    return true;
}
// the owner's launches are enqueued (or have failed): record the completion events under the mutex, end the launch window
static int sync_slot_launched(mmw_ctx *ctx, bool record) {
    std::lock_guard<std::mutex> lock(g_sync_mu);
    if (ctx->device >= 0 && ctx->device < 64 && g_sync_owner[ctx->device] == ctx) g_sync_launching[ctx->device] = false;
    if (record) {
        MMW_HIP(hipEventRecord(ctx->pipe_ang[0], ctx->q_ang));
        MMW_HIP(hipEventRecord(ctx->pipe_ang[1], ctx->q_rd));
        ctx->pipe_ang_used[0] = ctx->pipe_ang_used[1] = true;
        for (int i = 2; i < PIPE_RING_MAX; ++i) ctx->pipe_ang_used[i] = false;
    }
    return MMW_OK;
}
static void sync_slot_forget(mmw_ctx *ctx) {
    std::lock_guard<std::mutex> lock(g_sync_mu);
    for (int d = 0; d < 64; ++d)
        if (g_sync_owner[d] == ctx) {
            g_sync_owner[d] = nullptr;
            g_sync_launching[d] = false;
        }
}

static int chain3d_impl(mmw_ctx *ctx, const void *d_cubes, RawView rv, void *d_rd, void *d_out, int n_frames, int V, int S, int C,
                        int A, int flags, bool allow_sync = true);

// Have the device-synchronised chain calls enqueued since the last check completed?  Waits for them, reads the abort
// word and, if a bounded spin gave up (the two launches of a call did not run side by side: a tool that serialises kernel
// dispatches such as rocprofv3 --pmc, another process holding the CUs ...), resets the hand-off state and RE-RUNS the
// affected calls on the event schedule -- same kernels as the serial schedule, no device-side waiting -- so that the
// outputs the caller is about to read are complete.  MMW_CHAIN_NO_RERUN=1: report the timeout instead (MMW_ERR_HIP).
extern "C++" {
namespace mmw {
int chain_settle(mmw_ctx *ctx) {
    if (!ctx->chain_dirty) return MMW_OK;
    ctx->chain_settling = true;
    struct Done {
        mmw_ctx *c;
        ~Done() { c->chain_settling = false; }
    } done{ctx};
    if (ctx->q_rd) {
        MMW_HIP(hipStreamSynchronize(ctx->q_rd));
        MMW_HIP(hipStreamSynchronize(ctx->q_ang));
    }
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    unsigned aborted = 0;
    MMW_HIP(hipMemcpy(&aborted, ctx->chain_ctl + CTL_ABORT, sizeof(unsigned), hipMemcpyDeviceToHost));
    ctx->chain_dirty = false;
    std::vector<ChainCall> calls;
    calls.swap(ctx->chain_calls);
    if (!aborted) return MMW_OK;
    // counters are inconsistent (aborted workgroups leave without drawing their past-the-end ticket): fresh layout
    MMW_HIP(hipMemset(ctx->chain_ctl, 0, CTL_WORDS * sizeof(unsigned)));
    ctx->chain_layout[0] = 0;
    if (opt_int(ctx, "MMW_CHAIN_NO_RERUN", 0))
        return set_error(MMW_ERR_HIP, "chain hand-off timed out on the device (output incomplete): the range-Doppler and angle "
                         "launches of the device-synchronised chain must run concurrently -- a tool that serialises kernel "
                         "dispatches (e.g. rocprofv3 --pmc) needs MMW_CHAIN_MODE=events");
    for (const ChainCall &c : calls) {
        MMW_TRY(chain3d_impl(ctx, c.d_cubes, RawView{c.ntx, c.nrx, 0, c.i16}, nullptr, c.d_out, c.n_frames, c.V, c.S, c.C, c.A, c.flags, false));
        ++ctx->chain_fallbacks;
    }
    if (ctx->pipe_pending) {
        for (int i = 0; i < PIPE_RING_MAX; ++i)
            if (ctx->pipe_ang_used[i]) MMW_HIP(hipStreamWaitEvent(ctx->stream, ctx->pipe_ang[i], 0));
        ctx->pipe_pending = false;
    }
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    return MMW_OK;
}
}  // namespace mmw
}  // extern "C++"

// One RD launch + one angle launch for the whole call, synchronised through device counters (ChainSync).
static int chain3d_sync(mmw_ctx *ctx, const ChainPlan &plan, const void *d_cubes, RawView rv, void *d_out, int n_frames, int V,
                        int S, int C, int A_bins, int flags) {
    const size_t cube_bytes = (size_t)V * S * C * sizeof(cplx<float>);
    const long bins = (long)S * C;
    const bool mag_out = (flags & MMW_ANGLE_MAGNITUDE) != 0;
    const int tiles = angle_sync_tiles(bins, mag_out), v_live = plan.vskip > 2 ? V - 2 : V;
    MMW_REQUIRE((long)n_frames * std::max(tiles, v_live) < (1L << 30), "too many frames for one chain call");
    if (ensure_pipe_queues(ctx, plan.rd_cus) != MMW_OK) return MMW_ERR_UNSUPPORTED;
    MMW_TRY(ensure_scratch(ctx, (size_t)plan.ring_frames * cube_bytes));
    if (!ctx->chain_ctl) {
        MMW_HIP(hipMalloc((void **)&ctx->chain_ctl, CTL_WORDS * sizeof(unsigned)));
        ctx->chain_layout[0] = 0;       // forces the reset below
    }
    const long layout[6] = {V, bins, plan.vskip, plan.ring_frames, tiles, (long)(uintptr_t)ctx->scratch};
    if (std::memcmp(layout, ctx->chain_layout, sizeof(layout)) != 0) {
        // new ring layout: nothing of the old one may be in flight, counters restart from zero
        MMW_HIP(hipStreamSynchronize(ctx->q_rd));
        MMW_HIP(hipStreamSynchronize(ctx->q_ang));
        MMW_HIP(hipStreamSynchronize(ctx->q_ang2));
        MMW_HIP(hipMemsetAsync(ctx->chain_ctl, 0, CTL_WORDS * sizeof(unsigned), ctx->stream));
        MMW_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(ctx->chain_layout, layout, sizeof(layout));
        ctx->chain_g = 0;
        ctx->chain_rd_base = ctx->chain_ang_base = 0;
    }
    const bool window = !(flags & MMW_ANGLE_NO_WINDOW), shift = !(flags & MMW_ANGLE_NO_SHIFT), mag = flags & MMW_ANGLE_MAGNITUDE;
    float h[16];
    for (int i = 0; i < V; ++i) h[i] = window ? (float)np_window(TAB_HANN, i, V) : 1.f;
    ChainSync cs{};
    cs.ctl = ctx->chain_ctl;
    cs.rd_base = ctx->chain_rd_base;
    cs.ang_base = ctx->chain_ang_base;
    cs.s0 = (unsigned)(ctx->chain_g % (unsigned long long)plan.ring_frames);
    cs.u0 = (unsigned)(ctx->chain_g / (unsigned long long)plan.ring_frames);
    cs.ring = plan.ring_frames;
    cs.V = V;
    cs.vskip = plan.vskip;
    cs.v_live = v_live;
    cs.tiles = tiles;
    cs.n_frames = n_frames;
    cs.timeout = (unsigned long long)std::max(1, opt_int(ctx, "MMW_CHAIN_TIMEOUT_MS", 2000)) * 100000ull;      // 100 MHz ticks
    // order of the RD work items inside a frame: plain cubes by antenna; raw cubes rx-major, so that the ntx planes that
    // de-interleave the same raw rows are handed out back to back (their second and third read come from cache)
    {
        int n = 0;
        const bool raw = rv.ntx > 1;
        for (int i = 0; i < V && n < 16; ++i) {
            const int v = raw ? (i % rv.ntx) * rv.nrx + i / rv.ntx : i;       // i = rx * ntx + tx when raw
            if (plan.vskip > 2 && (v == 0 || v == V - 1)) continue;
            cs.vmap |= (unsigned long long)v << (4 * n++);
        }
    }
    cs.ntx = rv.ntx > 1 ? rv.ntx : 1;
    cs.nrx = rv.nrx;
    cs.i16 = rv.i16;
    cs.naps_rd = std::max(0, 32);
    cs.naps_ang = std::max(0, 4);
    const int n_rd_items = n_frames * v_live, n_ang_items = n_frames * tiles;
    const bool fused = fused_rd_ok(S, C);
    int rd_grid = std::min(plan.rd_cus, n_rd_items);
    if (!fused)     // compile-time mixed-radix producer: several workgroups per CU where they fit
        MMW_TRY(launch_rd_mixed_ct(ctx, nullptr, 0, nullptr, n_rd_items, S, C, RawView{rv.ntx > 1 ? rv.ntx : 1, rv.nrx}, nullptr, plan.rd_cus,
                                   &rd_grid, true));
    const int ang_grid = std::min((ctx->num_cu - plan.rd_cus) * 3, n_ang_items);
    hipStream_t main_stream = ctx->stream;
    MMW_HIP(hipEventRecord(ctx->pipe_begin, main_stream));
    MMW_HIP(hipStreamWaitEvent(ctx->q_rd, ctx->pipe_begin, 0));
    MMW_HIP(hipStreamWaitEvent(ctx->q_ang, ctx->pipe_begin, 0));
    // an event-mode call that is still running reads the same scratch: order behind it
    for (int i = 0; i < PIPE_RING_MAX; ++i)
        if (ctx->pipe_ang_used[i] && (ctx->pipe_ring != -1)) MMW_HIP(hipStreamWaitEvent(ctx->q_rd, ctx->pipe_ang[i], 0));
    ctx->pipe_ring = -1;            // marks "last chain call was device-synchronised" for the event-mode layout check
    ctx->pipe_slot_bytes = 0;
    ctx->pipe_pending = true;
    ctx->chain_dirty = true;
    // the counters advance by exactly these amounts whether or not the launches below succeed in full
    ctx->chain_g += (unsigned long long)n_frames;
    ctx->chain_rd_base += (unsigned)(n_rd_items + rd_grid);       // every workgroup draws one ticket past the end
    ctx->chain_ang_base += (unsigned)(n_ang_items + ang_grid);
    int rc;
    if (opt_int(ctx, "MMW_CHAIN_DIAG_SKIP_RD", 0)) {
        rc = MMW_OK;        // diagnostics: no producer -- the consumer's bounded spin must give up (tests of the abort path)
    } else {
        ctx->stream = ctx->q_rd;
        ProfScope ps(ctx, "rd");
        if (fused) rc = launch_rd_fused_sync(ctx, d_cubes, ctx->scratch, n_rd_items, cs, rd_grid);
        else rc = launch_rd_mixed_ct(ctx, d_cubes, (long)S * C, ctx->scratch, n_rd_items, S, C, RawView{1, 0}, &cs, plan.rd_cus, &rd_grid);
    }
    if (rc == MMW_OK) {
        ctx->stream = ctx->q_ang;
        ProfScope ps(ctx, "angle");
        switch (V) {
            case 4: rc = launch_angle64_sync<4>(ctx, ctx->scratch, d_out, bins, mag, h, shift, cs, ang_grid); break;
            case 8: rc = launch_angle64_sync<8>(ctx, ctx->scratch, d_out, bins, mag, h, shift, cs, ang_grid); break;
            case 12: rc = launch_angle64_sync<12>(ctx, ctx->scratch, d_out, bins, mag, h, shift, cs, ang_grid); break;
            default: rc = launch_angle64_sync<16>(ctx, ctx->scratch, d_out, bins, mag, h, shift, cs, ang_grid); break;
        }
    }
    ctx->stream = main_stream;
    if (rc != MMW_OK) {
        // a launch failed: the counters no longer match the host mirror; drain and force a fresh layout next time
        (void)sync_slot_launched(ctx, false);
        (void)hipStreamSynchronize(ctx->q_rd);
        (void)hipStreamSynchronize(ctx->q_ang);
        ctx->chain_layout[0] = 0;
        return rc;
    }
    MMW_TRY(sync_slot_launched(ctx, true));
    ctx->chain_calls.push_back(ChainCall{d_cubes, d_out, rv.ntx, rv.nrx, n_frames, V, S, C, A_bins, flags, rv.i16});
    return MMW_OK;
}

static int chain3d_impl(mmw_ctx *ctx, const void *d_cubes, RawView rv, void *d_rd, void *d_out, int n_frames, int V, int S, int C,
                        int A, int flags, bool allow_sync) {
    const int magnitude = flags & MMW_ANGLE_MAGNITUDE;
    MMW_REQUIRE(ctx && d_cubes && d_out, "null argument");
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && A >= V, "bad shape (need A >= V)");
    if (n_frames == 0) return MMW_OK;
    MMW_HIP(hipSetDevice(ctx->device));
    const size_t cube_bytes = (size_t)V * S * C * sizeof(cplx<float>);
    const size_t in_frame_bytes = rv.i16 ? cube_bytes / 2 : cube_bytes;      // int16 (I, Q) cells are 4 bytes
    const size_t out_frame_bytes = (size_t)A * S * C * (magnitude ? sizeof(float) : sizeof(cplx<float>));
    ChainPlan plan = chain_plan(ctx, d_rd != nullptr, rv.ntx > 1, n_frames, V, S, C, A, flags, allow_sync, rv.i16 != 0);
    if (plan.sync && ctx->chain_calls.size() >= 256) MMW_TRY(chain_settle(ctx));     // bound the unchecked backlog
    if (plan.sync && !sync_slot_acquire(ctx))       // another context of this process has synchronised work in flight here
        plan = chain_plan(ctx, d_rd != nullptr, rv.ntx > 1, n_frames, V, S, C, A, flags, false, rv.i16 != 0);
    rv.vskip = plan.vskip;
    const bool pipelined = plan.pipelined;
    const int ring = plan.ring, rd_cus = plan.rd_cus;
    int chunk = plan.chunk;
    if (plan.sync) {
        const int rc = chain3d_sync(ctx, plan, d_cubes, rv, d_out, n_frames, V, S, C, A, flags);
        if (rc != MMW_OK) (void)sync_slot_launched(ctx, false);         // (whatever the exit path: end the launch window)
        if (rc != MMW_ERR_UNSUPPORTED) return rc;          // queues unavailable: the serial schedule below
        chunk = std::min(n_frames, 1024);
    }
    const bool queues_ok = pipelined && ensure_pipe_queues(ctx, rd_cus) == MMW_OK;
    if (!queues_ok) {
        // serial schedule (also the fallback when the chain queues cannot be created: same results)
        MMW_JOIN(ctx);
        if (pipelined) chunk = std::min(n_frames, 1024);
        void *rd_scratch = nullptr;
        if (!d_rd) {
            MMW_TRY(ensure_scratch(ctx, (size_t)chunk * cube_bytes));
            rd_scratch = ctx->scratch;
        }
        for (int f0 = 0; f0 < n_frames; f0 += chunk) {
            const int nf = std::min(chunk, n_frames - f0);
            const char *in = (const char *)d_cubes + (size_t)f0 * in_frame_bytes;
            char *rd = d_rd ? (char *)d_rd + (size_t)f0 * cube_bytes : (char *)rd_scratch;
            MMW_TRY(range_doppler_impl(ctx, in, rd, nullptr, nf, V, S, C, rv));
            MMW_TRY(angle_fft_impl(ctx, rd, (char *)d_out + (size_t)f0 * out_frame_bytes, nf, V, S, C, A, flags));
        }
        return MMW_OK;
    }
    MMW_TRY(ensure_scratch(ctx, (size_t)ring * chunk * cube_bytes));
    char *rd_scratch = (char *)ctx->scratch;
    hipStream_t main_stream = ctx->stream;
    // both queues start after whatever the caller enqueued on the context stream
    MMW_HIP(hipEventRecord(ctx->pipe_begin, main_stream));
    MMW_HIP(hipStreamWaitEvent(ctx->q_rd, ctx->pipe_begin, 0));
    MMW_HIP(hipStreamWaitEvent(ctx->q_ang, ctx->pipe_begin, 0));
    MMW_HIP(hipStreamWaitEvent(ctx->q_ang2, ctx->pipe_begin, 0));
    const int n_angq = opt_int(ctx, "MMW_ANGLE_QUEUES", 2) >= 2 ? 2 : 1;
    // The ring slots of this call overlay those of the previous (possibly still running) chain call only if the
    // layout is the same; otherwise RD must not start before every earlier angle launch has read its slot.
    const size_t slot_bytes = (size_t)chunk * cube_bytes;
    if (ctx->pipe_slot_bytes != slot_bytes || ctx->pipe_ring != ring) {
        for (int i = 0; i < PIPE_RING_MAX; ++i)
            if (ctx->pipe_ang_used[i]) MMW_HIP(hipStreamWaitEvent(ctx->q_rd, ctx->pipe_ang[i], 0));
        ctx->pipe_slot_bytes = slot_bytes;
        ctx->pipe_ring = ring;
    }
    ctx->pipe_pending = true;       // from here on work may be in flight on the chain queues, whatever the exit path
    auto fail = [&](int rc) {           // leave nothing in flight that join_pipe's events do not cover
        ctx->stream = main_stream;
        ctx->active_cus = 0;
        (void)hipStreamSynchronize(ctx->q_rd);
        (void)hipStreamSynchronize(ctx->q_ang);
        (void)hipStreamSynchronize(ctx->q_ang2);
        return rc;
    };
#define MMW_PIPE_HIP(call)                                                                                        \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fail(set_error(MMW_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__)); \
    } while (0)
    int k = 0;
    for (int f0 = 0; f0 < n_frames; f0 += chunk, ++k) {
        const int nf = std::min(chunk, n_frames - f0), slot = k % ring;
        char *rd = rd_scratch + (size_t)slot * slot_bytes;
        // RD(k) may overwrite its slot only after angle(k - ring) has read it
        // (also across calls: the events persist, so a back-to-back chain keeps the pipeline full)
        if (ctx->pipe_ang_used[slot]) MMW_PIPE_HIP(hipStreamWaitEvent(ctx->q_rd, ctx->pipe_ang[slot], 0));
        ctx->stream = ctx->q_rd;
        ctx->active_cus = rd_cus > 0 ? rd_cus : ctx->num_cu;
        int rc = range_doppler_impl(ctx, (const char *)d_cubes + (size_t)f0 * in_frame_bytes, rd, nullptr, nf, V, S, C, rv);
        ctx->stream = main_stream;
        ctx->active_cus = 0;
        if (rc != MMW_OK) return fail(rc);
        MMW_PIPE_HIP(hipEventRecord(ctx->pipe_rd[slot], ctx->q_rd));
        hipStream_t q_a = (n_angq == 2 && (k & 1)) ? ctx->q_ang2 : ctx->q_ang;
        MMW_PIPE_HIP(hipStreamWaitEvent(q_a, ctx->pipe_rd[slot], 0));
        ctx->stream = q_a;
        rc = angle_fft_impl(ctx, rd, (char *)d_out + (size_t)f0 * out_frame_bytes, nf, V, S, C, A, flags);
        ctx->stream = main_stream;
        if (rc != MMW_OK) return fail(rc);
        MMW_PIPE_HIP(hipEventRecord(ctx->pipe_ang[slot], q_a));
        ctx->pipe_ang_used[slot] = true;
    }
#undef MMW_PIPE_HIP
    // The context stream joins lazily (join_pipe) at the next entry point that uses it.
    return MMW_OK;
}

int mmw_chain3d(mmw_ctx *ctx, const void *d_cubes, void *d_rd, void *d_out, int n_frames, int V, int S, int C,
                int A, int flags) {
    return chain3d_impl(ctx, d_cubes, RawView{1, 0}, d_rd, d_out, n_frames, V, S, C, A, flags);
}

static int raw_args_ok(int num_rx, int num_tx) {
    MMW_REQUIRE(num_rx > 0 && num_tx > 0 && (long)num_rx * num_tx <= 4096, "bad antenna counts %d x %d", num_rx, num_tx);
    return MMW_OK;
}

int mmw_range_doppler_raw(mmw_ctx *ctx, const void *d_raw, void *d_out, int n_frames, int num_rx, int num_tx, int S,
                          int loops) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    MMW_TRY(raw_args_ok(num_rx, num_tx));
    return range_doppler_impl(ctx, d_raw, d_out, nullptr, n_frames, num_rx * num_tx, S, loops, RawView{num_tx, num_rx});
}

int mmw_chain3d_raw(mmw_ctx *ctx, const void *d_raw, void *d_rd, void *d_out, int n_frames, int num_rx, int num_tx,
                    int S, int loops, int A, int flags) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_TRY(raw_args_ok(num_rx, num_tx));
    return chain3d_impl(ctx, d_raw, RawView{num_tx, num_rx}, d_rd, d_out, n_frames, num_rx * num_tx, S, loops, A, flags);
}

// int16 (I, Q) raw cubes [F][num_rx][S][num_tx * loops][2]: the conversion to float and the TDM de-interleave happen in the
// loads of the first kernel -- the 4-byte cells are read once, no complex64 cube is written in between.  NO UPSTREAM ORACLE
// for this sample layout (the reference gets complex cubes from the absent cpsl_datasets reader): results are defined as,
// and tested bit-identical to, mmw_virtual_array_reformat_i16 followed by the virtual-array entry point.
int mmw_range_doppler_raw_i16(mmw_ctx *ctx, const void *d_raw_i16, void *d_out, int n_frames, int num_rx, int num_tx, int S,
                              int loops) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    MMW_TRY(raw_args_ok(num_rx, num_tx));
    MMW_REQUIRE(num_tx > 1, "int16 cubes are raw TDM cubes (num_tx > 1); convert a single-tx cube with mmw_virtual_array_reformat_i16");
    return range_doppler_impl(ctx, d_raw_i16, d_out, nullptr, n_frames, num_rx * num_tx, S, loops, RawView{num_tx, num_rx, 0, 1});
}

int mmw_chain3d_raw_i16(mmw_ctx *ctx, const void *d_raw_i16, void *d_rd, void *d_out, int n_frames, int num_rx, int num_tx, int S,
                        int loops, int A, int flags) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_TRY(raw_args_ok(num_rx, num_tx));
    MMW_REQUIRE(num_tx > 1, "int16 cubes are raw TDM cubes (num_tx > 1)");
    return chain3d_impl(ctx, d_raw_i16, RawView{num_tx, num_rx, 0, 1}, d_rd, d_out, n_frames, num_rx * num_tx, S, loops, A, flags);
}

int mmw_range_profile(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C,
                      int chirp_idx) {
    return range_profile_impl<float>(ctx, d_cubes, d_out, n_frames, V, S, C, chirp_idx);
}

int mmw_range_profile_f64(mmw_ctx *ctx, const void *d_cubes, double *d_out, int n_frames, int V, int S, int C,
                          int chirp_idx) {
    return range_profile_impl<double>(ctx, d_cubes, d_out, n_frames, V, S, C, chirp_idx);
}

int mmw_range_zoom(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C, int chirp_idx,
                   int m, double f0_cycles_per_sample, double df_cycles_per_sample) {
    MMW_REQUIRE(ctx && d_cubes && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && m > 0, "bad shape");
    MMW_REQUIRE(chirp_idx >= -C && chirp_idx < C, "chirp_idx %d out of range", chirp_idx);
    MMW_REQUIRE((size_t)S * sizeof(float2) <= 64 * 1024, "too many samples for the zoom DFT row buffer");
    if (chirp_idx < 0) chirp_idx += C;
    if (n_frames == 0) return MMW_OK;
    const void *hann;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &hann));
    MMW_TRY(ensure_scratch(ctx, (size_t)n_frames * V * m * sizeof(float2)));
    // rows = (frame, antenna): x[row][i] = cube[frame][v][i][chirp]
    if (czt_length(S) > 0 && !env_int("MMW_ZOOM_DIRECT", 0)) {
        std::vector<double> freq(m);
        for (int k = 0; k < m; ++k) freq[k] = f0_cycles_per_sample + (double)k * df_cycles_per_sample;
        const CztPlan *plan = nullptr;
        MMW_TRY(czt_plan(ctx, freq.data(), m, S, &plan));
        CztArgs z{};
        z.x = (const float2 *)d_cubes + chirp_idx;
        z.outer_stride = (long)S * C;
        z.inner_stride = 0;
        z.elem_stride = C;
        z.s_lo = 0;
        z.s_keep = 1;
        z.win = (const float *)hann;
        z.M = m;
        z.rows = (long)n_frames * V;
        z.out = (float2 *)ctx->scratch;
        MMW_TRY(launch_czt_rows(ctx, *plan, z));
    } else {
        hipLaunchKernelGGL(k_zoom_dft, dim3(n_frames * V), dim3(256), (size_t)S * sizeof(float2), ctx->stream,
                           (const float2 *)d_cubes + chirp_idx, (long)S * C, (long)C, (const float *)hann,
                           (float2 *)ctx->scratch, S, m, f0_cycles_per_sample, df_cycles_per_sample);
        MMW_TRY(check_launch("zoom_dft"));
    }
    const long n = (long)n_frames * m;
    hipLaunchKernelGGL((k_mean_abs_over_v<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const cplx<float> *)ctx->scratch, d_out, n_frames, V, m);
    return check_launch("mean_abs_over_v");
}

int mmw_range_angle(mmw_ctx *ctx, const void *d_cubes, float *d_out, int n_frames, int V, int S, int C, int A,
                    int chirp_idx, const int *h_rx, int n_rx, int perform_windowing) {
    MMW_REQUIRE(ctx && d_cubes && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && A > 0, "bad shape");
    MMW_REQUIRE(chirp_idx >= -C && chirp_idx < C, "chirp_idx %d out of range", chirp_idx);
    MMW_REQUIRE(n_rx >= 0 && (n_rx == 0 || h_rx), "bad rx list");
    if (chirp_idx < 0) chirp_idx += C;
    std::vector<int> rx;
    if (n_rx == 0)
        for (int v = 0; v < V; ++v) rx.push_back(v);
    else
        for (int i = 0; i < n_rx; ++i) {
            int r = h_rx[i];
            if (r < 0) r += V;
            MMW_REQUIRE(r >= 0 && r < V, "rx index %d out of range", h_rx[i]);
            rx.push_back(r);
        }
    const int n_sel = (int)rx.size();
    MMW_REQUIRE(n_sel <= A, "more antennas (%d) than angle bins (%d)", n_sel, A);
    if (n_frames == 0) return MMW_OK;
    MMW_TRY(ensure_scratch(ctx, (size_t)n_frames * n_sel * S * sizeof(cplx<float>)));
    for (int i = 0; i < n_sel; ++i) {
        FftArgs a{};
        a.in = (const cplx<float> *)d_cubes + (long)rx[i] * S * C + chirp_idx;
        a.out = (cplx<float> *)ctx->scratch + (long)i * S;
        a.outer = n_frames;
        a.inner = 1;
        a.n_in = S;
        a.in_outer_stride = (long)V * S * C;
        a.in_axis_stride = C;
        a.in_inner_stride = 1;
        a.out_outer_stride = (long)n_sel * S;
        a.out_axis_stride = 1;
        a.out_inner_stride = 1;
        a.scale = 1.0;
        if (perform_windowing) {
            MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &a.win_axis));
            a.scale = np_window(TAB_HANN, rx[i], V);   // hann over ALL V antennas, then the subset
        }
        MMW_TRY((launch_fft_axis<float, float>(ctx, a, S, false)));
    }
    FftArgs b{};
    b.in = ctx->scratch;
    b.out = d_out;
    b.outer = n_frames;
    b.inner = S;
    b.n_in = n_sel;
    b.in_outer_stride = (long)n_sel * S;
    b.in_axis_stride = S;
    b.in_inner_stride = 1;
    b.out_outer_stride = (long)S * A;
    b.out_axis_stride = 1;
    b.out_inner_stride = A;
    b.scale = 1.0;
    b.shift = 1;
    b.magnitude = 1;
    return launch_fft_axis<float, float>(ctx, b, A, false);
}

// ------------------------------------------------------------------ CFAR
static int cfar2d_impl(mmw_ctx *ctx, const double *d_X, double *d_thr, double *d_noise, uint8_t *d_mask, int n_frames,
                       int R, int D, int kind, int train_r, int train_d, int guard_r, int guard_d, double scale,
                       int k_rank) {
    MMW_REQUIRE(ctx && d_X, "null argument");
    MMW_REQUIRE(n_frames >= 0 && R > 0 && D > 0, "bad shape");
    MMW_REQUIRE(train_r >= 0 && train_d >= 0 && guard_r >= 0 && guard_d >= 0, "negative window size");
    MMW_REQUIRE(kind == MMW_CFAR_CA || kind == MMW_CFAR_OS, "2-D CFAR kind must be CA or OS");
    MMW_REQUIRE(n_frames <= 65535, "at most 65535 frames per call");
    const int hr = train_r + guard_r, hd = train_d + guard_d;
    const long ntrain = (long)(2 * hr + 1) * (2 * hd + 1) - (long)(2 * guard_r + 1) * (2 * guard_d + 1);
    if (kind == MMW_CFAR_OS) MMW_REQUIRE(k_rank >= 1 && k_rank <= ntrain, "k_rank must be between 1 and %ld, got %d", ntrain, k_rank);
    if (kind == MMW_CFAR_OS && !d_thr && !d_noise && d_mask && scale > 0.0) {
        // mask only: one count per cell under test instead of an order-statistic selection (k_cfar2d_os_mask)
        // the GUI's window (gui_configs/processor_params.yaml: [5,5] / [3,2]) and the (4,4)/(2,2) of the tests: four cells per thread
        if (n_frames > 0 && opt_int(ctx, "MMW_OS_MASK_FORM", 1) == 1) {
            Cfar2dArgs a{d_X, nullptr, nullptr, d_mask, R, D, kind, train_r, train_d, guard_r, guard_d, scale, k_rank, 0};
            const dim3 grid((D + OSV_T - 1) / OSV_T, (R + OSV_T - 1) / OSV_T, n_frames);
#define MMW_OSV(tr, td, gr, gd) \
    if (train_r == tr && train_d == td && guard_r == gr && guard_d == gd) { \
        ProfScope ps(ctx, "cfar"); \
        hipLaunchKernelGGL((k_cfar2d_os_mask_v<tr, td, gr, gd>), grid, dim3(256), 0, ctx->stream, a); \
        return check_launch("cfar2d_os_mask_v"); \
    }
            MMW_OSV(5, 5, 3, 2)
            MMW_OSV(4, 4, 2, 2)
#undef MMW_OSV
        }
        const int TW = OSM_TC + 2 * hd, TH = OSM_TR + 2 * hr, TWp = ((TW + 15) / 32) * 32 + 16;
        const size_t lds_m = (size_t)TH * TWp * sizeof(double);
        if (lds_m <= 64 * 1024) {
            if (n_frames == 0) return MMW_OK;
            ProfScope ps(ctx, "cfar");
            Cfar2dArgs a{d_X, nullptr, nullptr, d_mask, R, D, kind, train_r, train_d, guard_r, guard_d, scale, k_rank, 0};
            dim3 grid((D + OSM_TC - 1) / OSM_TC, (R + OSM_TR - 1) / OSM_TR, n_frames);
            hipLaunchKernelGGL(k_cfar2d_os_mask, grid, dim3(OSM_TR * OSM_TC), lds_m, ctx->stream, a);
            return check_launch("cfar2d_os_mask");
        }
    }
    MMW_REQUIRE(2 * hd + 1 <= 512, "Doppler window too wide for the exact summation order");
    if (kind == MMW_CFAR_CA) {
        // 32 x 32 tile (k_cfar2d_ca) while tile + halo + row-sum tables fit the default LDS limit
        const size_t th = CA_TR + 2 * hr, tw = CA_TC + 2 * hd;
        const size_t lds_ca = (th * tw + 2 * th * CA_TC) * sizeof(double);
        if (lds_ca <= 64 * 1024) {
            if (n_frames == 0) return MMW_OK;
            ProfScope ps(ctx, "cfar");
            Cfar2dArgs a{d_X, d_thr, d_noise, d_mask, R, D, kind, train_r, train_d, guard_r, guard_d, scale, k_rank, 0};
            dim3 grid((D + CA_TC - 1) / CA_TC, (R + CA_TR - 1) / CA_TR, n_frames);
            hipLaunchKernelGGL(k_cfar2d_ca, grid, dim3(256), lds_ca, ctx->stream, a);
            return check_launch("cfar2d_ca");
        }
    }
    const size_t n_tile = (size_t)(CFAR_TR + 2 * hr) * (CFAR_TC + 2 * hd);
    size_t npad = 1;
    while (npad < n_tile) npad <<= 1;
    const size_t aux_ca = 2 * (size_t)(CFAR_TR + 2 * hr) * CFAR_TC * sizeof(double);     // row-sum tables
    size_t aux_os = npad * 8 + npad * 2 + ((n_tile + 1) & ~(size_t)1) * 2 + 16;         // keys, positions, ranks
    const size_t integ = (size_t)(OS_COARSE - 1) * (CFAR_TR + 2 * hr + 1) * (CFAR_TC + 2 * hd + 1) * 2;
    const bool os_fast = kind == MMW_CFAR_OS && npad <= 1024 && n_tile * sizeof(double) + aux_os + integ <= 64 * 1024;
    if (os_fast) aux_os += integ;
    const size_t lds = n_tile * sizeof(double) + (kind == MMW_CFAR_OS ? aux_os : aux_ca);
    MMW_REQUIRE(kind != MMW_CFAR_OS || npad <= 32768, "OS-CFAR window too large");
    if (lds > 64 * 1024) return set_error(MMW_ERR_UNSUPPORTED, "CFAR window %dx%d too large for the LDS tile", 2 * hr + 1, 2 * hd + 1);
    if (n_frames == 0) return MMW_OK;
    ProfScope ps(ctx, "cfar");
    Cfar2dArgs a{d_X, d_thr, d_noise, d_mask, R, D, kind, train_r, train_d, guard_r, guard_d, scale, k_rank, os_fast ? 1 : 0};
    dim3 grid((D + CFAR_TC - 1) / CFAR_TC, (R + CFAR_TR - 1) / CFAR_TR, n_frames);
    hipLaunchKernelGGL(k_cfar2d, grid, dim3(CFAR_TR * CFAR_TC), lds, ctx->stream, a);
    return check_launch("cfar2d");
}

int mmw_cfar2d(mmw_ctx *ctx, const double *d_X, double *d_thr, double *d_noise, uint8_t *d_mask, int n_frames,
               int R, int D, int kind, int train_r, int train_d, int guard_r, int guard_d, double scale,
               int k_rank) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    return cfar2d_impl(ctx, d_X, d_thr, d_noise, d_mask, n_frames, R, D, kind, train_r, train_d, guard_r, guard_d, scale,
                       k_rank);
}

int mmw_cfar1d(mmw_ctx *ctx, const double *d_x, double *d_thr, double *d_noise, uint8_t *d_mask, int n_rows,
               int L, int kind, int num_train, int num_guard, double scale, int k_rank) {
    MMW_REQUIRE(ctx && d_x, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_rows >= 0 && L > 0 && num_train >= 0 && num_guard >= 0, "bad shape");
    MMW_REQUIRE(kind >= MMW_CFAR_CA && kind <= MMW_CFAR_SO, "unknown CFAR kind %d", kind);
    MMW_REQUIRE(num_train <= 512, "num_train too large for the exact summation order");
    if (kind == MMW_CFAR_OS)
        MMW_REQUIRE(k_rank >= 1 && k_rank <= 2 * num_train, "k_rank must be between 1 and %d, got %d", 2 * num_train, k_rank);
    if (n_rows == 0) return MMW_OK;
    Cfar1dArgs a{d_x, d_thr, d_noise, d_mask, n_rows, L, kind, num_train, num_guard, scale, k_rank};
    const long total = (long)n_rows * L;
    hipLaunchKernelGGL(k_cfar1d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a);
    return check_launch("cfar1d");
}

static int compact2d_impl(mmw_ctx *ctx, const uint8_t *d_mask, int32_t *d_dets, int32_t *d_counts, int n_frames,
                          int R, int D, int cap) {
    MMW_REQUIRE(ctx && d_mask && d_dets && d_counts, "null argument");
    MMW_REQUIRE(n_frames >= 0 && R > 0 && D > 0 && cap >= 0, "bad shape");
    if (n_frames == 0) return MMW_OK;
    ProfScope ps(ctx, "compact");
    hipLaunchKernelGGL(k_compact2d, dim3(n_frames), dim3(1024), 0, ctx->stream, d_mask, d_dets, d_counts, R, D, cap);
    return check_launch("compact2d");
}

int mmw_compact2d(mmw_ctx *ctx, const uint8_t *d_mask, int32_t *d_dets, int32_t *d_counts, int n_frames,
                  int R, int D, int cap) {
    MMW_REQUIRE(ctx, "ctx is null");
    MMW_JOIN(ctx);
    return compact2d_impl(ctx, d_mask, d_dets, d_counts, n_frames, R, D, cap);
}

// One call for the per-frame detection pipeline of RangeDopplerDetector2D over a batch: RD cube (fp32, all
// antennas), float64 |RD| of antenna 0, 2-D CFAR, ordered compaction.  The stages run back to back on the context
// stream: running the RD kernel beside the detection kernels on CU-masked queues was measured 2x SLOWER, because the
// float64 FFT and CFAR kernels are latency/ALU bound and scale with the CUs they get (unlike the angle kernel).
int mmw_detect_batch(mmw_ctx *ctx, const void *d_cubes, void *d_rd, double *d_mag64, uint8_t *d_mask,
                     int32_t *d_dets, int32_t *d_counts, float *d_l1, int n_frames, int V, int S, int C, int cfar_kind,
                     int train_r, int train_d, int guard_r, int guard_d, double scale, int k_rank, int cap) {
    MMW_REQUIRE(ctx && d_cubes && d_rd && d_mag64 && d_mask && d_dets && d_counts, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && n_frames <= 65535 && V > 0 && S > 0 && C > 0 && cap >= 0, "bad shape");
    if (n_frames == 0) return MMW_OK;
    MMW_TRY(range_doppler_impl(ctx, d_cubes, d_rd, nullptr, n_frames, V, S, C, RawView{1, 0}, d_l1));
    MMW_TRY(range_doppler_mag64_impl(ctx, d_cubes, d_mag64, n_frames, V, S, C, 0));
    MMW_TRY(cfar2d_impl(ctx, d_mag64, nullptr, nullptr, d_mask, n_frames, S, C, cfar_kind, train_r, train_d, guard_r,
                        guard_d, scale, k_rank));
    return compact2d_impl(ctx, d_mask, d_dets, d_counts, n_frames, S, C, cap);
}

// ------------------------------------------------------------------ point cloud
static int fill_ant_list(const int *h_ant, int n_ant, int V, int A, AntList *ants) {
    MMW_REQUIRE(n_ant >= 1 && n_ant <= MAX_ANT && n_ant <= A && h_ant, "antenna list must have 1..%d entries (<= A)", MAX_ANT);
    ants->n = n_ant;
    for (int i = 0; i < n_ant; ++i) {
        int a = h_ant[i];
        if (a < 0) a += V;
        MMW_REQUIRE(a >= 0 && a < V, "antenna index %d out of range", h_ant[i]);
        ants->idx[i] = a;
    }
    return MMW_OK;
}

static void launch_angle_argmax(mmw_ctx *ctx, dim3 grid, const float2 *rd, const int32_t *dets, const int32_t *counts,
                                int32_t *idx, int V, int S, int C, int cap, const AntList &ants, int A, int shift,
                                const float2 *twA, const ArgmaxRefine &rf) {
#define MMW_ARGMAX(NA) \
    hipLaunchKernelGGL(k_angle_argmax<NA>, grid, dim3(256), 0, ctx->stream, rd, dets, counts, idx, V, S, C, cap, ants, A, shift, twA, rf)
    // lists of up to 8 antennas with the error bound: the lane-resident routine of the fused detection stage
    if (rf.l1 && ants.n <= DET_LATE_MAX_ANT && A == 64 && opt_int(ctx, "MMW_ARGMAX_FORM", 1) == 1) {
        // one lane per detection (64 bins as register FFTs): 256 detections of a frame per workgroup and pass
        const dim3 g2((unsigned)std::min((cap + 255) / 256, 4), grid.y);
#define MMW_ARGMAX_DETS(NA, SH) \
    hipLaunchKernelGGL((k_angle_argmax_dets<NA, SH>), g2, dim3(256), 0, ctx->stream, rd, dets, counts, idx, V, S, C, cap, ants, twA, rf)
        if (ants.n <= 4) {
            if (shift) MMW_ARGMAX_DETS(4, true); else MMW_ARGMAX_DETS(4, false);
        } else if (ants.n <= 8) {
            if (shift) MMW_ARGMAX_DETS(8, true); else MMW_ARGMAX_DETS(8, false);
        } else {
            if (shift) MMW_ARGMAX_DETS(16, true); else MMW_ARGMAX_DETS(16, false);
        }
#undef MMW_ARGMAX_DETS
        return;
    }
    if (rf.l1 && ants.n <= DET_MAX_ANT && A <= 1024) {
        if (ants.n <= 4)
            hipLaunchKernelGGL(k_angle_argmax_lanes<4>, grid, dim3(256), (size_t)A * 8, ctx->stream, rd, dets, counts, idx, V, S, C, cap, ants, A,
                               shift, twA, rf);
        else
            hipLaunchKernelGGL(k_angle_argmax_lanes<DET_MAX_ANT>, grid, dim3(256), (size_t)A * 8, ctx->stream, rd, dets, counts, idx, V, S, C,
                               cap, ants, A, shift, twA, rf);
        return;
    }
    if (ants.n <= 4) MMW_ARGMAX(4);
    else if (ants.n <= 8) MMW_ARGMAX(8);
    else if (ants.n <= 16) MMW_ARGMAX(16);
    else MMW_ARGMAX(32);
#undef MMW_ARGMAX
}

int mmw_angle_argmax(mmw_ctx *ctx, const void *d_rd, const int32_t *d_dets, const int32_t *d_counts,
                     int32_t *d_idx, int n_frames, int V, int S, int C, int cap, const int *h_ant, int n_ant,
                     int A, int shift) {
    MMW_REQUIRE(ctx && d_rd && d_dets && d_counts && d_idx, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && n_frames <= 65535 && V > 0 && S > 0 && C > 0 && cap >= 0 && A > 0, "bad shape");
    AntList ants{};
    MMW_TRY(fill_ant_list(h_ant, n_ant, V, A, &ants));
    if (n_frames == 0 || cap == 0) return MMW_OK;
    const void *twA = nullptr;
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, A, &twA));
    ProfScope ps(ctx, "argmax");
    dim3 grid(std::min((cap + 3) / 4, 32), n_frames);      // 128 detections per frame in one pass, more by looping
    launch_angle_argmax(ctx, grid, (const float2 *)d_rd, d_dets, d_counts, d_idx, V, S, C, cap, ants, A, shift,
                        (const float2 *)twA, ArgmaxRefine{});
    return check_launch("angle_argmax");
}

int mmw_plane_l1(mmw_ctx *ctx, const void *d_cubes, float *d_l1, int n_frames, int V, int S, int C) {
    MMW_REQUIRE(ctx && d_cubes && d_l1, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && (long)n_frames * V < (1L << 31), "bad shape");
    return plane_l1_impl(ctx, d_cubes, d_l1, n_frames, V, S, C);
}

static int plane_l1_impl(mmw_ctx *ctx, const void *d_cubes, float *d_l1, int n_frames, int V, int S, int C) {
    if (n_frames == 0) return MMW_OK;
    const void *ws, *wc;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &ws));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, C, &wc));
    ProfScope ps(ctx, "plane_l1");
    hipLaunchKernelGGL(k_plane_l1, dim3((unsigned)(n_frames * V)), dim3(256), 0, ctx->stream, (const float2 *)d_cubes, d_l1, S,
                       C, (const float *)ws, (const float *)wc);
    return check_launch("plane_l1");
}

// Rounding-error budget of the float32 range-Doppler cell values, in units of eps = 2^-24 of the plane's L1 norm
// (|rd32 - rd64| <= ulps * eps * sum |w x|), for whichever kernel range_doppler_impl runs on an S x C plane.
//   windows: two products with table values, <= 8.
//   structured transforms (k_rd_fused_256x128, k_rd_lds, k_rd_mixed_ct, k_rd_split2_ct: RegFFT / RegDFT levels): per
//     prime factor p of the axis length one twiddle / inter-level product (complex FMA product + table value, <= 4) plus
//     p == 2: the butterfly's add (inside the 4);  odd p: the real-symmetric form of RegDFT -- pair sums (1), a
//     (p - 1) / 2-term FMA chain with literal coefficients ((p + 1) / 2), the +- i combine (1): <= (p + 7) / 2 + 1.
//     The 127-point level of 63 x 127 / 127 x 32 / 254 x 50 is that form on MFMA tiles (64-term exact FMA chains).
//   run-time mixed-radix kernel: its levels may be direct R-term chains: R per level on top.
//   generic path: radix-2 levels (4 each) for powers of two, an N-term direct sum (N + 4) otherwise.
static int rd_error_ulps(int S, int C) {
    auto factors = [](int n) {
        int u = 0;
        for (int p = 2; n > 1; ++p)
            while (n % p == 0) {
                u += 4 + (p == 2 ? 0 : (p == 127 ? 88 : (p + 7) / 2 + 1));       // (127: the bfloat16 x 3 MFMA form, mmw_fft_mixed_ct.h)
                n /= p;
            }
        return u;
    };
    const int structured = 8 + factors(S) + factors(C);
    const bool no_fused = env_int("MMW_NO_FUSED_RD", 0), no_mixed = env_int("MMW_NO_MIXED_RD", 0);
    if (!no_fused && (fused_rd_ok(S, C) || rd_lds_supported(S, C))) return structured;
    if (!no_mixed && rd_mixed_ct_supported(S, C)) return structured;
    RdMixedPlan mp;
    if (!no_mixed && rd_mixed_plan(S, C, sizeof(cplx<float>), &mp))
        return structured + mp.s1 + mp.s2 + mp.c1 + mp.c2 + 2 * (mp.rad_s[0] + mp.rad_s[1] + mp.rad_c[0] + mp.rad_c[1]);
    if (rd_split_ct_supported(S, C) && !env_int("MMW_NO_SPLIT_RD", 0)) return structured;
    int ulps = 8;
    for (int axis = 0; axis < 2; ++axis) {
        const int n = axis ? C : S;
        if (is_pow2(n)) ulps += 4 * ilog2(n);
        else ulps += n + 4;
    }
    return ulps;
}

// float64 re-evaluation of the detections k_angle_argmax flagged (ra.n_flag / ra.list): fixed grids, the flagged count
// stays on the device, workgroups beyond it leave at once.  ra.partial must hold n_split * REFINE_PARTS * ants.n entries.
static int launch_argmax_refine(mmw_ctx *ctx, const RefineArgs &ra) {
    const size_t lds = refine_tabs_lds(ra.S, ra.C);
    MMW_REQUIRE(lds <= 150 * 1024, "plane %d x %d too large for the refinement tables", ra.S, ra.C);
    if (lds > 48 * 1024) {
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_argmax_refine_part), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_argmax_refine_whole), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (ra.n_split > 0) {
        const int units = std::min(ra.n_split, std::max(1, 512));
        hipLaunchKernelGGL(k_argmax_refine_part, dim3(ra.parts, units), dim3(256), lds, ctx->stream, ra);
        MMW_TRY(check_launch("argmax_refine_part"));
        hipLaunchKernelGGL(k_argmax_refine_finish, dim3(std::min((ra.n_split + 3) / 4, ctx->num_cu)), dim3(256), 0, ctx->stream, ra);
        MMW_TRY(check_launch("argmax_refine_finish"));
    }
    if (ra.list_cap > ra.n_split) {
        hipLaunchKernelGGL(k_argmax_refine_whole, dim3(ctx->num_cu), dim3(256), lds, ctx->stream, ra);
        MMW_TRY(check_launch("argmax_refine_whole"));
    }
    return MMW_OK;
}

// Slices per plane sum: a few flagged evaluations per hundred frames are expected, and ~1000 workgroups of the slice
// kernel are resident at once: 16 slices (latency of the single task), fewer and longer ones for very large batches.
static int refine_parts(int n_frames) {
    int p = REFINE_PARTS;
    while (p > 2 && (long)p * n_frames > 60000) p /= 2;
    return p;
}

static int fill_refine_args(mmw_ctx *ctx, RefineArgs *ra, int S, int C, int A) {
    const void *twA64, *twS64, *twC64, *ws64, *wc64;
    MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, A, &twA64));
    MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, S, &twS64));
    MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, C, &twC64));
    MMW_TRY(get_table<double>(ctx, TAB_HANN, S, &ws64));
    MMW_TRY(get_table<double>(ctx, TAB_HANN, C, &wc64));
    ra->S = S;
    ra->C = C;
    ra->A = A;
    ra->ws = (const double *)ws64;
    ra->wc = (const double *)wc64;
    ra->twS = (const cplx<double> *)twS64;
    ra->twC = (const cplx<double> *)twC64;
    ra->twA = (const cplx<double> *)twA64;
    return MMW_OK;
}

// The worst-case bound assumes every rounding error of every partial sum lines up; measured float32 errors stay below
// 1.2 % of it (tests/argmax_margin.py: 68 000 evaluations on the 256 x 128 and 63 x 100 planes).
//  * mmw_detect_points (CA-CFAR detections: strong cells) uses the FULL worst-case bound with the pairwise form of the test:
//    0.15 % of the evaluations are re-done in float64.
//  * the stand-alone mmw_angle_argmax_exact serves the other detectors.  With the GUI's OS-CFAR parameters (rho 0.7, alpha 2:
//    470 mostly noise-level detections per 256 x 128 frame, flat angle spectra) the full bound sends 12.8 % of the
//    evaluations to float64 -- 120 per frame, each a direct DFT sum over whole planes: 29-36 us/frame against 7.2 with an
//    eighth of the bound (~10x above anything observed; 1.6 % refined).  Default: 1/8 + the pairwise pass (lists of up to 8
//    antennas); MMW_ARGMAX_BOUND_DIV=1 selects the worst case, i.e. a proof, at that price.
static float argmax_bound_div(const mmw_ctx *ctx) { return (float)std::max(1, opt_int(ctx, "MMW_ARGMAX_BOUND_DIV", 1)); }


int mmw_angle_argmax_exact(mmw_ctx *ctx, const void *d_cubes, const float *d_l1, const void *d_rd, const int32_t *d_dets,
                           const int32_t *d_counts, int32_t *d_idx, int n_frames, int V, int S, int C, int cap,
                           const int *h_ant, int n_ant, int A, int shift, int *h_n_refined) {
    MMW_REQUIRE(ctx && d_cubes && d_l1 && d_rd && d_dets && d_counts && d_idx, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && n_frames <= 65535 && V > 0 && S > 0 && C > 0 && cap >= 0 && A > 0, "bad shape");
    MMW_REQUIRE((long)n_frames * cap < (1L << 31), "too many detection slots for one call");
    AntList ants{};
    MMW_TRY(fill_ant_list(h_ant, n_ant, V, A, &ants));
    if (h_n_refined) *h_n_refined = 0;
    if (n_frames == 0 || cap == 0) return MMW_OK;
    const void *twA = nullptr;
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, A, &twA));
    const int list_cap = n_frames * cap;
    const int n_split = std::min(list_cap, std::max(0, opt_int(ctx, "MMW_REFINE_SPLIT", 32768)));   // flagged detections whose
                                                                                          // plane sums are split over workgroups
    // Dense refinement (mmw_cells64.h): when many evaluations are flagged -- noise-level detections, e.g. the GUI's OS-CFAR --
    // the float64 cells of a whole frame come from one Doppler FFT per sample row + 256-term range sums instead of one
    // 32768-term sum per cell and antenna.  On when the call flags >= dense_min evaluations (8 per frame; MMW_ARGMAX_DENSE_MIN)
    // and the plane has a kernel (128 chirps); covers the first dense_cap entries of the list, the direct kernels the rest.
    const int max_cells = C == 128 ? cells64_max_cells(S, C) : 0;
    const int dense_min = max_cells > 0 ? std::max(1, opt_int(ctx, "MMW_ARGMAX_DENSE_MIN", 8 * n_frames)) : 0;
    const int dense_cap = dense_min > 0 ? (int)std::min<long>(list_cap, 256L * n_frames) : 0;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t list_bytes = up(256 + (size_t)list_cap * sizeof(int));
    const size_t part_bytes = up((size_t)n_split * REFINE_PARTS * n_ant * sizeof(cplx<double>));
    const size_t pos_bytes = dense_min > 0 ? up((size_t)list_cap * sizeof(int)) : 0;
    const size_t cell_bytes = up((size_t)dense_cap * n_ant * sizeof(cplx<double>));
    MMW_TRY(ensure_scratch(ctx, list_bytes + part_bytes + pos_bytes + cell_bytes));
    int *d_nflag = (int *)ctx->scratch, *d_list = (int *)((char *)ctx->scratch + 256);
    int *d_flagpos = dense_min > 0 ? (int *)((char *)ctx->scratch + list_bytes + part_bytes) : nullptr;
    cplx<double> *d_cells = (cplx<double> *)((char *)ctx->scratch + list_bytes + part_bytes + pos_bytes);
    MMW_HIP(hipMemsetAsync(d_nflag, 0, 256, ctx->stream));
    if (d_flagpos) MMW_HIP(hipMemsetAsync(d_flagpos, 0, (size_t)list_cap * sizeof(int), ctx->stream));
    const float eps = 5.9604645e-8f, div = argmax_bound_div(ctx);
    ArgmaxRefine rf{d_l1, d_nflag, d_list, list_cap, (float)rd_error_ulps(S, C) * eps / div, 4.f * (float)(n_ant + 4) * eps / div,
                    d_flagpos, dense_cap};
    ProfScope ps(ctx, "argmax");
    dim3 grid(std::min((cap + 3) / 4, 32), n_frames);      // 128 detections per frame in one pass, more by looping
    launch_angle_argmax(ctx, grid, (const float2 *)d_rd, d_dets, d_counts, d_idx, V, S, C, cap, ants, A, shift,
                        (const float2 *)twA, rf);
    MMW_TRY(check_launch("angle_argmax"));
    RefineArgs ra{};
    MMW_TRY(fill_refine_args(ctx, &ra, S, C, A));
    ra.cubes = (const float2 *)d_cubes;
    ra.dets = d_dets;
    ra.n_flag = d_nflag;
    ra.list = d_list;
    ra.list_cap = list_cap;
    ra.out_idx = d_idx;
    ra.V = V;
    ra.cap = cap;
    ra.ants = ants;
    ra.shift = shift;
    ra.partial = (cplx<double> *)((char *)ctx->scratch + list_bytes);
    ra.n_split = n_split;
    ra.parts = refine_parts(n_frames);
    ra.dense_min = dense_min;
    ra.dense_cap = dense_cap;
    if (dense_min > 0) {
        Cells64Args ca{};
        ca.cubes = (const float2 *)d_cubes;
        ca.dets = d_dets;
        ca.counts = d_counts;
        ca.flagpos = d_flagpos;
        ca.n_flag = d_nflag;
        ca.dense_min = dense_min;
        ca.out = d_cells;
        ca.V = V;
        ca.S = S;
        ca.cap = cap;
        ca.n_ant = n_ant;
        ca.max_cells = max_cells;
        ca.ants = ants;
        ca.ws = ra.ws;
        ca.wc = ra.wc;
        ca.twS = ra.twS;
        ca.twC = ra.twC;
        const size_t lds = cells64_lds(S, C, max_cells);
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cells64<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_cells64<128>, dim3((unsigned)n_ant, (unsigned)n_frames), dim3(C64_NT), lds, ctx->stream, ca);
        MMW_TRY(check_launch("cells64"));
        Argmax64ListArgs la{d_cells, d_nflag, d_list, dense_min, dense_cap, n_ant, A, shift, d_idx, ra.twA};
        hipLaunchKernelGGL(k_argmax64_list, dim3((unsigned)std::min(std::max(dense_cap / 4, 1), 4 * ctx->num_cu)), dim3(256), 0, ctx->stream, la);
        MMW_TRY(check_launch("argmax64_list"));
    }
    MMW_TRY(launch_argmax_refine(ctx, ra));
    if (h_n_refined) {
        MMW_HIP(hipMemcpyAsync(h_n_refined, d_nflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MMW_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MMW_OK;
}

// ------------------------------------------------------------------ fused detection + point-cloud indices (mmw_detect.h)
// The two queues of the overlapped detection schedule: the range-Doppler producer owns the first rd_cus CU-mask bits, the
// screening consumer the rest (mask bit i belongs to XCD i % 8, and to that XCD's shader engines in turn: multiples of 32
// keep every engine of every XCD equally populated -- an engine with fewer CUs than its neighbours paces the whole launch).
static int ensure_det_queues(mmw_ctx *ctx, int rd_cus) {
    if (ctx->det_unavailable) return set_error(MMW_ERR_UNSUPPORTED, "detection queues unavailable on this runtime");
    if (ctx->q_drd && ctx->q_drd_cus == rd_cus) return MMW_OK;
    if (ctx->q_drd) {
        MMW_HIP(hipStreamSynchronize(ctx->q_drd));
        MMW_HIP(hipStreamSynchronize(ctx->q_dscr));
        MMW_HIP(hipStreamDestroy(ctx->q_drd));
        MMW_HIP(hipStreamDestroy(ctx->q_dscr));
        ctx->q_drd = ctx->q_dscr = nullptr;
    }
    if (!ctx->det_begin) {
        MMW_HIP(hipEventCreateWithFlags(&ctx->det_begin, hipEventDisableTiming));
        MMW_HIP(hipEventCreateWithFlags(&ctx->det_rd_done, hipEventDisableTiming));
        MMW_HIP(hipEventCreateWithFlags(&ctx->det_scr_done, hipEventDisableTiming));
    }
    const int words = (ctx->num_cu + 31) / 32;
    std::vector<uint32_t> m_rd(words, 0u), m_scr(words, 0u);
    for (int i = 0; i < ctx->num_cu; ++i) ((i < rd_cus) ? m_rd : m_scr)[i / 32] |= 1u << (i % 32);
    hipError_t e1 = hipExtStreamCreateWithCUMask(&ctx->q_drd, (uint32_t)words, m_rd.data()), e2 = hipSuccess;
    if (e1 == hipSuccess) e2 = hipExtStreamCreateWithCUMask(&ctx->q_dscr, (uint32_t)words, m_scr.data());
    if (e1 != hipSuccess || e2 != hipSuccess) {
        if (ctx->q_drd) (void)hipStreamDestroy(ctx->q_drd);
        ctx->q_drd = ctx->q_dscr = nullptr;
        ctx->det_unavailable = true;
        (void)hipGetLastError();
        return set_error(MMW_ERR_UNSUPPORTED, "detection queue creation failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    ctx->q_drd_cus = rd_cus;
    return MMW_OK;
}

namespace {
struct DetectPlan {
    bool ok, ct_window;
    int band_rows, band_pitch, words;
    size_t lds_screen, lds_cell;
};
DetectPlan detect_plan(int S, int C, int kind, int tr, int td, int gr, int gd, int n_az, int n_el, int A = 64) {
    DetectPlan p{};
    const long n = (long)S * C;
    const int hr = tr + gr, hd = td + gd;
    // lists of up to 8 antennas: any angle FFT size; 9 to 16: the late argmax only, i.e. 64 angle bins
    const int n_max = A == 64 ? DET_LATE_MAX_ANT : DET_MAX_ANT;
    if (kind != MMW_CFAR_CA || n_az > n_max || n_el > n_max || n > (1L << 20) || A < 1 || A > 1024) return p;
    if (0) return p;
    p.words = (int)((n + 31) / 32);
    p.lds_cell = cell_exact_lds(S, C, 2 * hr + 1, 2 * hd + 1);
    if (p.lds_cell > 64 * 1024) return p;
    // compile-time windows (the launch below knows the same two): four rows / columns per thread, padded band rows
    p.ct_window = (tr == 4 && td == 4 && gr == 2 && gd == 2) || (tr == 5 && td == 5 && gr == 3 && gd == 2);
    p.band_pitch = p.ct_window ? det_band_pitch(C, hd) : C;
    const int unit = p.ct_window ? 4 : 1;
    // Band of rows under test: its cells and the 2 hr halo rows travel as at most DET_LOADS loads per thread (two cells per
    // load); 32 rows x 128 columns give each of the 1024 threads one item in either CFAR pass.
    const long cap_cells = (long)DET_LOADS * DET_NT * 2;
    int band = std::min<long>(std::max(1, tune_int("MMW_DETECT_BAND", DET_NT >= 1024 ? 32 : 24)), cap_cells / C - 2 * hr);
    band = std::min(band, std::max(unit, S - 2 * hr));
    band = band / unit * unit;
    if (band < unit) return p;                              // rows too long for the band loads
    p.band_rows = band;
    p.lds_screen = detect_screen_lds(band + 2 * hr, C, band, p.band_pitch, p.words, A);
    p.ok = p.lds_screen <= 160 * 1024 - 64;
    return p;
}
int fill_det_ant(const int *h_ant, int n_ant, int V, int A, DetAnt *out, AntList *full) {
    out->n = 0;
    full->n = 0;
    if (n_ant == 0) return MMW_OK;
    MMW_TRY(fill_ant_list(h_ant, n_ant, V, A, full));
    out->n = full->n;
    for (int i = 0; i < full->n && i < DET_LATE_MAX_ANT; ++i) out->idx[i] = full->idx[i];
    return MMW_OK;
}
}  // namespace

int mmw_detect_points_supported(int S, int C, int cfar_kind, int train_r, int train_d, int guard_r, int guard_d, int n_az,
                                int n_el, int A) {
    if (S <= 0 || C <= 0 || train_r < 0 || train_d < 0 || guard_r < 0 || guard_d < 0 || n_az < 0 || n_el < 0) return 0;
    return detect_plan(S, C, cfar_kind, train_r, train_d, guard_r, guard_d, n_az, n_el, A).ok ? 1 : 0;
}

int mmw_detect_points(mmw_ctx *ctx, const void *d_cubes, void *d_rd, float *d_l1, float *d_mag32, int32_t *d_dets,
                      int32_t *d_counts, int32_t *d_az_idx, int32_t *d_el_idx, int n_frames, int V, int S, int C, int cfar_kind,
                      int train_r, int train_d, int guard_r, int guard_d, double scale, int k_rank, int cap, const int *h_az,
                      int n_az, int shift_az, const int *h_el, int n_el, int shift_el, int A, int *h_stats) {
    MMW_REQUIRE(ctx && d_cubes && d_rd && d_l1 && d_dets && d_counts, "null argument");
    MMW_TRY(join_pipe(ctx, true));      // (the previous call's deferred tail: joined below, behind this call's range-Doppler kernel)
    MMW_REQUIRE(n_frames >= 0 && V > 0 && S > 0 && C > 0 && cap >= 0 && A > 0, "bad shape");
    MMW_REQUIRE(train_r >= 0 && train_d >= 0 && guard_r >= 0 && guard_d >= 0, "negative window size");
    MMW_REQUIRE(n_az >= 0 && n_el >= 0 && (n_az == 0 || d_az_idx) && (n_el == 0 || d_el_idx), "antenna list without an index buffer");
    MMW_REQUIRE((long)n_frames * std::max(cap, 1) < (1L << 31) && (long)n_frames * 64 < (1L << 31), "too many detection slots for one call");
    const DetectPlan plan = detect_plan(S, C, cfar_kind, train_r, train_d, guard_r, guard_d, n_az, n_el, A);
    if (!plan.ok)
        return set_error(MMW_ERR_UNSUPPORTED, "mmw_detect_points: no screening kernel for this request (CA-CFAR on planes whose "
                         "float32 magnitudes fit the LDS, <= %d antennas per list, <= %d with 64 angle bins): use mmw_detect_batch + "
                         "mmw_angle_argmax_exact", DET_MAX_ANT, DET_LATE_MAX_ANT);
    const int hr = train_r + guard_r, hd = train_d + guard_d;
    const long n_train = (long)(2 * hr + 1) * (2 * hd + 1) - (long)(2 * guard_r + 1) * (2 * guard_d + 1);
    MMW_REQUIRE(n_train >= 1 && n_train < (1L << 30), "empty training window");
    if (cfar_kind == MMW_CFAR_OS) MMW_REQUIRE(k_rank >= 1 && k_rank <= n_train, "k_rank must be between 1 and %ld, got %d", n_train, k_rank);
    DetectArgs a{};
    AntList az_full{}, el_full{};
    MMW_TRY(fill_det_ant(h_az, n_az, V, A, &a.az, &az_full));
    MMW_TRY(fill_det_ant(h_el, n_el, V, A, &a.el, &el_full));
    if (h_stats)
        for (int i = 0; i < 5; ++i) h_stats[i] = 0;
    if (n_frames == 0) return MMW_OK;
    const void *twA = nullptr, *ws64, *wc64, *twS64, *twC64;
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, A, &twA));
    MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, S, &twS64));
    MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, C, &twC64));
    MMW_TRY(get_table<double>(ctx, TAB_HANN, S, &ws64));
    MMW_TRY(get_table<double>(ctx, TAB_HANN, C, &wc64));
    // scratch: counters | flagged frames | undecided cells | bit masks | flagged argmax evaluations + their partial sums
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const int cell_cap = std::max(4096, 16 * n_frames);
    const int list_cap = n_frames * cap;
    const int n_split = std::min(list_cap, std::max(0, opt_int(ctx, "MMW_REFINE_SPLIT", 32768)));
    const size_t b_ctl = up(DCTL_WORDS * sizeof(int)), b_ff = up((size_t)n_frames * sizeof(int)),
                 b_cells = up((size_t)cell_cap * 2 * sizeof(int)), b_bits = up((size_t)n_frames * (2 * DET_SPEC + 1) * sizeof(int));      // (speculative slots of the frames with undecided cells)
    const int list_cap2 = (int)std::min<long>(2L * list_cap, 0x7fffffffL);      // both lists flag into one
    const int n_split2 = std::min(list_cap2, n_split);
    const size_t b_list2 = up((size_t)std::max(list_cap2, 1) * sizeof(int));
    const size_t b_part = up((size_t)n_split2 * REFINE_PARTS * std::max(std::max(n_az, n_el), 1) * sizeof(cplx<double>));
    // Overlapped schedule (256 x 128 planes): the range-Doppler producer and the screening consumer run side by side on disjoint
    // CU sets, frames handed over through counters in device memory.  Built, tested and MEASURED: it does not
    // beat the serial schedule on this chip -- the range-Doppler kernel keeps its 5.2 TB/s down to ~224 CUs only, and the
    // screening of 1250 frames needs ~70 CU-milliseconds, i.e. 2.2 ms on the 32 CUs that leaves -- so it is an option, not the
    // default: MMW_DETECT_OVERLAP=1 selects it; MMW_DETECT_SCR_CUS = CUs of the consumer (a multiple of 32);
    // MMW_DETECT_TAIL=0: no second consumer launch behind the producer on the producer's CUs.
    const int want_overlap = opt_int(ctx, "MMW_DETECT_OVERLAP", -1);
    int scr_cus = opt_int(ctx, "MMW_DETECT_SCR_CUS", 32);
    if (scr_cus < 1 || scr_cus >= ctx->num_cu) scr_cus = 32;
    const bool overlap = fused_rd_ok(S, C) && !env_int("MMW_NO_FUSED_RD", 0) &&
                         want_overlap == 1 &&
                         ensure_det_queues(ctx, ctx->num_cu - scr_cus) == MMW_OK;
    const size_t b_sync = overlap ? up((CTL_CNT + (size_t)n_frames) * sizeof(unsigned)) : 0;
    size_t total = b_ctl + b_ff + b_cells + b_bits + b_sync;
    if (n_az || n_el) total += b_list2 + b_part;
    // Late argmax (64 angle bins -- the reference's az_el_fft_size --, MMW_DETECT_LATE_ARGMAX=0: in the screening kernel): the
    // screening workgroup stops at the ordered detection list and copies each detection's range-Doppler cells into a flat
    // record list of the context; both angle estimates are launches of the lane-per-detection routine over those records
    // (k_angle_argmax_recs: the same float32 test with the same worst-case bound) on the side queue, in front of the float64
    // refinement -- part of the tail, so with the tail deferred they run beside the NEXT call's range-Doppler kernel, and they
    // read nothing that call overwrites.  The in-kernel form (one wave per detection, ~1.2 detections' worth of lanes busy)
    // cost the screening launch a third of its time: 0.27 -> 0.18 ms per 1250 frames.
    const int NV = n_az + n_el;
    const size_t b_rec = up((size_t)list_cap * NV * sizeof(float2)), b_rslot = up((size_t)list_cap * sizeof(int32_t)),
                 b_l1c = up((size_t)n_frames * V * sizeof(float));
    const bool late = cap > 0 && NV > 0 && A == 64 && b_rec + b_rslot + b_l1c <= ((size_t)1 << 30) &&
                      opt_int(ctx, "MMW_ARGMAX_FORM", 1) == 1 && opt_int(ctx, "MMW_DETECT_LATE_ARGMAX", 1) != 0;
    if ((n_az > DET_MAX_ANT || n_el > DET_MAX_ANT) && !late)
        return set_error(MMW_ERR_UNSUPPORTED, "mmw_detect_points: lists of more than %d antennas need the late argmax (64 angle bins, "
                         "MMW_DETECT_LATE_ARGMAX / MMW_ARGMAX_FORM at their defaults, a detection capacity whose records fit 1 GiB)", DET_MAX_ANT);
    const size_t off_rec = total;
    if (late) total += b_rec + b_rslot + b_l1c;
    MMW_TRY(ensure_scratch(ctx, total));
    char *base = (char *)ctx->scratch;
    a.ctl = (int *)base;
    a.flag_frames = (int *)(base + b_ctl);
    a.cells = (int *)(base + b_ctl + b_ff);
    a.spec = (int *)(base + b_ctl + b_ff + b_cells);
    unsigned *sync_words = (unsigned *)(base + b_ctl + b_ff + b_cells + b_bits);      // tickets, abort word | planes published per frame
    char *next = base + b_ctl + b_ff + b_cells + b_bits + b_sync;
    int *list = (n_az || n_el) ? (int *)next : nullptr;
    cplx<double> *part = (n_az || n_el) ? (cplx<double> *)(next + b_list2) : nullptr;
    // Back-to-back calls (a frame loop over resident chunks): the previous call's tail -- exact cells, list insertion, float64
    // refinement: ~0.15 ms of latency-bound launches that leave the chip almost idle -- is still running on its side queues.
    // This call's range-Doppler kernel touches nothing the tail uses (the fused 256 x 128 kernel needs no scratch; its output
    // buffers are checked against the tail's), so it goes first and the tail is joined in front of the screening stage.
    // (the same holds for the compile-time mixed-radix kernels -- every shipped cfg's plane --: tables only, no scratch)
    const bool rd_no_scratch = (fused_rd_ok(S, C) && !env_int("MMW_NO_FUSED_RD", 0)) ||
                               (!fused_rd_ok(S, C) && rd_mixed_ct_supported(S, C) && !env_int("MMW_NO_MIXED_RD", 0));
    bool rd_first = !overlap && ctx->tail_pending && rd_no_scratch;
    if (rd_first) {
        const std::pair<const char *, size_t> outs[2] = {{(const char *)d_rd, (size_t)n_frames * V * S * C * 8},
                                                         {(const char *)d_l1, (size_t)n_frames * V * sizeof(float)}};
        for (const auto &o : outs)
            for (const auto &t : ctx->tail_bufs)
                if (o.first < t.first + t.second && t.first < o.first + o.second) rd_first = false;
    }
    // Behind a pending tail the range-Doppler planes of a 256 x 128 batch are handed out by TICKETS (the producer kernel of the
    // overlapped schedule: it never waits for anybody; same arithmetic, sc1 stores), so that a second launch of the same kernel can
    // join in.  The counters are an allocation of their own: the scratch belongs to the pending tail.
    const int n_planes_all = n_frames * V;
    const bool tickets = !overlap && fused_rd_ok(S, C) && !env_int("MMW_NO_FUSED_RD", 0) && n_planes_all > 4 * ctx->num_cu;
    auto ticketed_rd = [&](int grid_main, int grid_help) -> int {
        const size_t words = CTL_CNT + (size_t)n_frames;
        MMW_TRY(ensure_help_sync(ctx, words));
        ChainSync cs{};
        cs.ctl = ctx->help_sync;
        cs.frame_cnt = ctx->help_sync + CTL_CNT;
        cs.V = cs.v_live = V;
        cs.n_frames = n_frames;
        cs.ntx = 1;
        hipStream_t main_q = ctx->stream;
        MMW_HIP(hipMemsetAsync(ctx->help_sync, 0, words * sizeof(unsigned), main_q));
        if (grid_help > 0) {
            MMW_HIP(hipEventRecord(ctx->help_begin, main_q));
            MMW_HIP(hipStreamWaitEvent(ctx->q_tail, ctx->help_begin, 0));
        }
        int rc;
        {
            ProfScope ps(ctx, "rd");
            rc = launch_rd_fused_det(ctx, d_cubes, d_rd, d_l1, n_planes_all, cs, grid_main);
        }
        if (grid_help > 0) {
            if (rc == MMW_OK) {
                ctx->stream = ctx->q_tail;
                ProfScope ps(ctx, "rd_help");
                rc = launch_rd_fused_det(ctx, d_cubes, d_rd, d_l1, n_planes_all, cs, grid_help);
                ctx->stream = main_q;
            }
            MMW_HIP(hipEventRecord(ctx->help_done, ctx->q_tail));       // (in any case: joined by the caller)
            if (rc != MMW_OK) MMW_HIP(hipStreamWaitEvent(main_q, ctx->help_done, 0));
        }
        return rc;
    };
    bool helper = false;
    if (rd_first) {
        // the persistent kernel holds one workgroup per CU for its whole run: it leaves a few CUs to the tail's short workgroups
        // (32 at least: with fewer some shader engine has no free CU, and the dispatcher -- which places a launch's workgroups
        // engine by engine -- holds the tail's launches back until the range-Doppler launch ends)
        const int leave = std::max(0, std::min(opt_int(ctx, "MMW_DETECT_TAIL_CUS", 40), ctx->num_cu / 4));
        helper = tickets && leave > 0 && opt_int(ctx, "MMW_DETECT_RD_HELPER", 1) != 0;
        if (helper) {
            // On num_cu - leave CUs the launch is bound by CU-time (15000 planes x 23 us on 216 CUs: 1.60 ms), and the tail needs
            // its CUs for the first ~0.7-1.0 ms only: a second launch, `leave` workgroups, sits in the tail's queue BEHIND the
            // tail -- when the tail is done its CUs draw tickets too.
            MMW_TRY(ticketed_rd(ctx->num_cu - leave, leave));
        } else {
            ctx->rd_leave_cus = leave;
            const int rc = range_doppler_impl(ctx, d_cubes, d_rd, nullptr, n_frames, V, S, C, RawView{1, 0}, d_l1);
            ctx->rd_leave_cus = 0;
            MMW_TRY(rc);
        }
    }
    MMW_TRY(join_tail(ctx));
    if (helper) MMW_HIP(hipStreamWaitEvent(ctx->stream, ctx->help_done, 0));
    MMW_HIP(hipMemsetAsync(a.ctl, 0, DCTL_WORDS * sizeof(int), ctx->stream));      // counters
    if (overlap) MMW_HIP(hipMemsetAsync(sync_words, 0, b_sync, ctx->stream));
    // range-Doppler of every antenna (float32) with the planes' L1 norms
    if (!overlap && !rd_first) MMW_TRY(range_doppler_impl(ctx, d_cubes, d_rd, nullptr, n_frames, V, S, C, RawView{1, 0}, d_l1));
    const float eps = 5.9604645e-8f, div = argmax_bound_div(ctx);
    const int ulps = rd_error_ulps(S, C);
    a.rd = (const float2 *)d_rd;
    a.l1 = d_l1;
    a.mag32 = d_mag32;
    a.dets = d_dets;
    a.counts = d_counts;
    a.az_idx = d_az_idx;
    a.el_idx = d_el_idx;
    a.cell_cap = cell_cap;
    a.V = V;
    a.S = S;
    a.C = C;
    a.cap = cap;
    a.words = plan.words;
    a.band_rows = plan.band_rows;
    a.band_pitch = plan.band_pitch;
    a.kind = cfar_kind;
    a.tr = train_r;
    a.td = train_d;
    a.gr = guard_r;
    a.gd = guard_d;
    a.n_train = (int)n_train;
    a.k_rank = k_rank;
    a.scale = scale;
    // the screening band always uses the full worst-case bound (MMW_DETECT_BAND_MULT widens it: test hook that sends more
    // cells through the float64 decision, or -- when huge -- whole frames back to the caller)
    a.k_fft = (float)ulps * eps * (float)std::max(1, opt_int(ctx, "MMW_DETECT_BAND_MULT", 1));
    a.A = A;
    a.shift_az = shift_az;
    a.shift_el = shift_el;
    a.twA = (const float2 *)twA;
    a.rf_az = ArgmaxRefine{d_l1, a.ctl + DCTL_ARGMAX, list, list_cap2, (float)ulps * eps / div, 4.f * (float)(n_az + 4) * eps / div};
    a.rf_el = ArgmaxRefine{d_l1, a.ctl + DCTL_ARGMAX, list, list_cap2, (float)ulps * eps / div, 4.f * (float)(n_el + 4) * eps / div};
    a.rf_el.tag = REFINE_SECOND;
    a.rf_el.n_tagged = a.ctl + DCTL_EL;
    a.n_frames = n_frames;
    if (late) {
        a.rec_cells = (float2 *)(base + off_rec);
        a.rec_slot = (int32_t *)(base + off_rec + b_rec);
        a.l1_copy = (float *)(base + off_rec + b_rec + b_rslot);
        a.rec_cap = list_cap;
        a.rf_az.l1 = a.rf_el.l1 = a.l1_copy;
    }
    if (overlap) {
        a.sy_ctl = sync_words;
        a.sy_frame_cnt = sync_words + CTL_CNT;
        a.sy_timeout = (unsigned long long)std::max(1, opt_int(ctx, "MMW_CHAIN_TIMEOUT_MS", 2000)) * 100000ull;       // 100 MHz ticks
        a.sy_naps = std::max(0, opt_int(ctx, "MMW_DETECT_NAPS", 4));
        ChainSync cs{};
        cs.ctl = sync_words;
        cs.frame_cnt = a.sy_frame_cnt;
        cs.V = cs.v_live = V;
        cs.n_frames = n_frames;
        cs.ntx = 1;
        const int n_planes = n_frames * V, rd_cus = ctx->num_cu - scr_cus;
        hipStream_t main_stream = ctx->stream;
        MMW_HIP(hipEventRecord(ctx->det_begin, main_stream));
        MMW_HIP(hipStreamWaitEvent(ctx->q_drd, ctx->det_begin, 0));
        MMW_HIP(hipStreamWaitEvent(ctx->q_dscr, ctx->det_begin, 0));
        auto screen = [&](int grid) -> int {
            auto go = [&](auto kern) -> int {
                MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_screen));
                hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(DET_NT), plan.lds_screen, ctx->stream, a);
                return check_launch("detect_screen_sync");
            };
            if (train_r == 4 && train_d == 4 && guard_r == 2 && guard_d == 2) return go(k_detect_screen_sync<4, 4, 2, 2>);
            if (train_r == 5 && train_d == 5 && guard_r == 3 && guard_d == 2) return go(k_detect_screen_sync<5, 5, 3, 2>);
            return go(k_detect_screen_sync<-1, -1, -1, -1>);
        };
        long long *d_clk = nullptr;
        if (opt_int(ctx, "MMW_PHASE_CLOCKS", 0)) {              // diagnostics: phase clocks of one consumer workgroup, summed over its frames
            MMW_HIP(hipMalloc((void **)&d_clk, 16 * sizeof(long long)));
            MMW_HIP(hipMemsetAsync(d_clk, 0, 16 * sizeof(long long), main_stream));
            MMW_HIP(hipStreamSynchronize(main_stream));
            a.clk = d_clk;
        }
        int rc = MMW_OK;
        if (!opt_int(ctx, "MMW_DETECT_DIAG_SKIP_RD", 0)) {      // (test hook: no producer -- the consumer's bounded wait must give up)
            ctx->stream = ctx->q_drd;
            ProfScope ps(ctx, "rd");
            rc = launch_rd_fused_det(ctx, d_cubes, d_rd, d_l1, n_planes, cs, std::min(rd_cus, n_planes));
        }
        if (rc == MMW_OK) {
            ctx->stream = ctx->q_dscr;
            ProfScope ps(ctx, "detect");
            rc = screen(std::min(scr_cus * (1024 / DET_NT), n_frames));       // (512-thread workgroups: two per CU)
        }
        if (rc == MMW_OK && opt_int(ctx, "MMW_DETECT_TAIL", 1)) {
            // frames the consumer's CUs have not reached when the producer drains: the same kernel (same ticket counter) on
            // the producer's CUs, behind it in its queue
            ctx->stream = ctx->q_drd;
            ProfScope ps(ctx, "detect_tail");
            rc = screen(std::min(rd_cus * (1024 / DET_NT), n_frames));
        }
        ctx->stream = main_stream;
        // join in any case: whatever was enqueued runs to its end (bounded waits) before the context's next work
        MMW_HIP(hipEventRecord(ctx->det_rd_done, ctx->q_drd));
        MMW_HIP(hipEventRecord(ctx->det_scr_done, ctx->q_dscr));
        MMW_HIP(hipStreamWaitEvent(main_stream, ctx->det_rd_done, 0));
        MMW_HIP(hipStreamWaitEvent(main_stream, ctx->det_scr_done, 0));
        if (d_clk) {
            long long h[16] = {0};
            MMW_HIP(hipStreamSynchronize(main_stream));
            MMW_HIP(hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
            MMW_HIP(hipFree(d_clk));
            a.clk = nullptr;
            std::fprintf(stderr, "detect_screen_sync %dx%d, workgroup 0 of the consumer: %lld frames; clocks per frame: wait %lld load %lld cfar %lld "
                         "compact %lld argmax %lld\n", S, C, h[7], h[5] / std::max(1LL, h[7]), h[0] / std::max(1LL, h[7]), h[1] / std::max(1LL, h[7]),
                         h[2] / std::max(1LL, h[7]), h[3] / std::max(1LL, h[7]));
            const long long nf = std::max(1LL, h[7]);
            std::fprintf(stderr, "  band loop per frame: top %lld land(+load wait) %lld issue+barrier %lld pass1 %lld pass2 %lld pass3 %lld; candidates %lld\n",
                         h[8] / nf, h[9] / nf, h[10] / nf, h[11] / nf, h[12] / nf, h[13] / nf, h[14] / nf);
        }
        MMW_TRY(rc);
    } else {
        ProfScope ps(ctx, "detect");
        auto go = [&](auto kern) -> int {
            MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_screen));
            if (tune_int("MMW_PHASE_CLOCKS", 0)) {
                // diagnostics: shader clocks of a mid-batch workgroup's phases (load + |.|, CFAR bands, compaction, argmax)
                long long *d = nullptr, h[5] = {0};
                MMW_HIP(hipMalloc((void **)&d, sizeof(h)));
                MMW_HIP(hipMemsetAsync(d, 0, sizeof(h), ctx->stream));
                DetectArgs b = a;
                b.clk = d;
                hipLaunchKernelGGL(kern, dim3((unsigned)n_frames), dim3(DET_NT), plan.lds_screen, ctx->stream, b);
                MMW_HIP(hipStreamSynchronize(ctx->stream));
                MMW_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                MMW_HIP(hipFree(d));
                std::fprintf(stderr, "detect_screen %dx%d clocks: load %lld cfar %lld compact %lld argmax %lld\n", S, C, h[1] - h[0],
                             h[2] - h[1], h[3] - h[2], h[4] - h[3]);
                return MMW_OK;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)n_frames), dim3(DET_NT), plan.lds_screen, ctx->stream, a);
            return MMW_OK;
        };
        // the windows of the reference's own configs as compile-time constants: (4,4)/(2,2) (tests/verify_processors.py:165,
        // SURVEY.md 8d) and the GUI's (5,5)/(3,2) (gui_configs/processor_params.yaml:44-45); anything else at run time
        if (train_r == 4 && train_d == 4 && guard_r == 2 && guard_d == 2) MMW_TRY(go(k_detect_screen<4, 4, 2, 2>));
        else if (train_r == 5 && train_d == 5 && guard_r == 3 && guard_d == 2) MMW_TRY(go(k_detect_screen<5, 5, 3, 2>));
        else MMW_TRY(go(k_detect_screen<-1, -1, -1, -1>));
        MMW_TRY(check_launch("detect_screen"));
    }
    // Behind the screening: the exact decision of the undecided cells (context stream) and the float64 refinement of the flagged
    // argmax evaluations (side stream) run SIDE BY SIDE -- neither needs the other: the frames with undecided cells carry them in
    // speculative slots --, then k_detect_insert puts the cells decided positive into their lists.
    const bool refine = cap > 0 && (n_az || n_el);
    hipStream_t main_stream = ctx->stream;
    if (!ctx->q_side) {
        MMW_HIP(hipStreamCreateWithFlags(&ctx->q_side, hipStreamNonBlocking));
        MMW_HIP(hipStreamCreateWithFlags(&ctx->q_tail, hipStreamNonBlocking));
        MMW_HIP(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
        MMW_HIP(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
        MMW_HIP(hipEventCreateWithFlags(&ctx->tail_done, hipEventDisableTiming));
        MMW_HIP(hipEventCreateWithFlags(&ctx->help_begin, hipEventDisableTiming));
        MMW_HIP(hipEventCreateWithFlags(&ctx->help_done, hipEventDisableTiming));
    }
    MMW_HIP(hipEventRecord(ctx->side_fork, main_stream));
    MMW_HIP(hipStreamWaitEvent(ctx->q_tail, ctx->side_fork, 0));
    // from here on the context stream is only the point the tail is joined to: on error, or when the caller wants the
    // statistics / the tail is not to be deferred (MMW_DETECT_DEFER_TAIL=0), before this call returns; else at the next entry point
    auto tail_end = [&](bool join_now) -> int {
        MMW_HIP(hipEventRecord(ctx->tail_done, ctx->q_tail));
        ctx->tail_pending = true;
        ctx->tail_bufs = {{(const char *)d_cubes, (size_t)n_frames * V * S * C * 8}, {(const char *)d_dets, (size_t)n_frames * cap * 8},
                          {(const char *)d_counts, (size_t)n_frames * 4}, {(const char *)d_az_idx, d_az_idx ? (size_t)n_frames * cap * 4 : 0},
                          {(const char *)d_el_idx, d_el_idx ? (size_t)n_frames * cap * 4 : 0}};
        return join_now ? join_tail(ctx) : MMW_OK;
    };
    if (refine) {
        MMW_HIP(hipStreamWaitEvent(ctx->q_side, ctx->side_fork, 0));
        ctx->stream = ctx->q_side;
        int rc = MMW_OK;
        if (late) {
            // angle estimates of every record (one lane each), then the float64 refinement of what they flag
            ProfScope ps(ctx, "argmax_tail");
            const unsigned grid = (unsigned)std::max(1, std::min((list_cap + 255) / 256, 2 * ctx->num_cu));
            auto recs = [&](auto kern, int off, int32_t *idx, const AntList &ants, const ArgmaxRefine &rf) {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, ctx->stream, (const float2 *)a.rec_cells, (const int32_t *)a.rec_slot,
                                   (const int *)(a.ctl + DCTL_RECS), a.rec_cap, NV, off, (const float *)a.l1_copy, idx, V, cap, ants,
                                   (const float2 *)twA, rf);
            };
            auto one_list = [&](int off, int32_t *idx, const AntList &ants, int shift, const ArgmaxRefine &rf) {
                if (ants.n == 0) return;
                if (ants.n <= 4) {
                    if (shift) recs(k_angle_argmax_recs<4, true>, off, idx, ants, rf); else recs(k_angle_argmax_recs<4, false>, off, idx, ants, rf);
                } else if (ants.n <= 8) {
                    if (shift) recs(k_angle_argmax_recs<8, true>, off, idx, ants, rf); else recs(k_angle_argmax_recs<8, false>, off, idx, ants, rf);
                } else {
                    if (shift) recs(k_angle_argmax_recs<16, true>, off, idx, ants, rf); else recs(k_angle_argmax_recs<16, false>, off, idx, ants, rf);
                }
            };
            one_list(0, d_az_idx, az_full, shift_az, a.rf_az);
            one_list(n_az, d_el_idx, el_full, shift_el, a.rf_el);
            rc = check_launch("angle_argmax_recs");
        }
        if (rc == MMW_OK) {
            ProfScope ps(ctx, "argmax_refine");
            RefineArgs ra{};
            rc = fill_refine_args(ctx, &ra, S, C, A);
            ra.cubes = (const float2 *)d_cubes;
            ra.dets = d_dets;
            ra.n_flag = a.ctl + DCTL_ARGMAX;
            ra.list = list;
            ra.list_cap = list_cap2;
            ra.out_idx = d_az_idx;
            ra.out_idx2 = d_el_idx;
            ra.V = V;
            ra.cap = cap;
            ra.ants = az_full;
            ra.ants2 = el_full;
            ra.shift = shift_az;
            ra.shift2 = shift_el;
            ra.partial = part;
            ra.n_split = n_split2;
            ra.parts = refine_parts(n_frames);
            if (rc == MMW_OK) rc = launch_argmax_refine(ctx, ra);
        }
        ctx->stream = main_stream;
        MMW_HIP(hipEventRecord(ctx->side_join, ctx->q_side));      // (joined below in any case: in front of the list insertion)
        if (rc != MMW_OK) {
            MMW_HIP(hipStreamWaitEvent(ctx->q_tail, ctx->side_join, 0));
            (void)tail_end(true);
            return rc;
        }
    }
    ctx->stream = ctx->q_tail;
    int rc_tail = MMW_OK;
    {
        ProfScope ps(ctx, "detect_exact");
        CellExactArgs ce{};
        ce.cubes = (const float2 *)d_cubes;
        ce.cells = a.cells;
        ce.n_cells = a.ctl + DCTL_CELLS;
        ce.cell_cap = cell_cap;
        ce.spec = a.spec;
        ce.V = V;
        ce.S = S;
        ce.C = C;
        ce.kind = cfar_kind;
        ce.tr = train_r;
        ce.td = train_d;
        ce.gr = guard_r;
        ce.gd = guard_d;
        ce.n_train = (int)n_train;
        ce.k_rank = k_rank;
        ce.scale = scale;
        ce.ws = (const double *)ws64;
        ce.wc = (const double *)wc64;
        ce.twS = (const cplx<double> *)twS64;
        ce.twC = (const cplx<double> *)twC64;
        long long *d_clk = nullptr;
        if (tune_int("MMW_PHASE_CLOCKS", 0)) {
            MMW_HIP(hipMalloc((void **)&d_clk, 5 * sizeof(long long)));
            MMW_HIP(hipMemsetAsync(d_clk, 0, 5 * sizeof(long long), ctx->stream));
            ce.clk = d_clk;
        }
        hipLaunchKernelGGL(k_cfar_cell_exact, dim3(std::min(cell_cap, 2 * ctx->num_cu)), dim3(CE_NT), plan.lds_cell, ctx->stream, ce);
        if (d_clk) {
            long long h[5] = {0};
            MMW_HIP(hipStreamSynchronize(ctx->stream));
            MMW_HIP(hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
            MMW_HIP(hipFree(d_clk));
            std::fprintf(stderr, "cfar_cell_exact clocks: tables %lld range sums %lld doppler sums %lld decision %lld\n", h[1] - h[0], h[2] - h[1],
                         h[3] - h[2], h[4] - h[3]);
        }
        rc_tail = check_launch("cfar_cell_exact");
        if (refine) MMW_HIP(hipStreamWaitEvent(ctx->q_tail, ctx->side_join, 0));       // the refinement is done with the speculative slots
        if (rc_tail == MMW_OK) {
            hipLaunchKernelGGL(k_detect_insert, dim3(std::min(n_frames, 4 * ctx->num_cu)), dim3(INS_NT), 0, ctx->stream, a);
            rc_tail = check_launch("detect_insert");
        }
    }
    ctx->stream = main_stream;
    {
        const int rc_end = tail_end(rc_tail != MMW_OK || h_stats != nullptr || !opt_int(ctx, "MMW_DETECT_DEFER_TAIL", 1));
        MMW_TRY(rc_tail);
        MMW_TRY(rc_end);
    }
    if (h_stats) {
        int h[DCTL_WORDS];
        MMW_HIP(hipMemcpyAsync(h, a.ctl, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        MMW_HIP(hipStreamSynchronize(ctx->stream));
        h_stats[0] = h[DCTL_FLAG_FRAMES];
        h_stats[1] = h[DCTL_CELLS];
        h_stats[2] = h[DCTL_FALLBACK];
        h_stats[3] = h[DCTL_ARGMAX] - h[DCTL_EL];
        h_stats[4] = h[DCTL_EL];
    }
    return MMW_OK;
}

int mmw_angle_argmax_cells64(mmw_ctx *ctx, const void *d_cells, int32_t *d_idx, int n_rows, int n_ant, int A, int shift) {
    MMW_REQUIRE(ctx && d_cells && d_idx, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_rows >= 0 && n_ant >= 1 && A >= n_ant, "bad shape (need 1 <= n_ant <= A)");
    if (n_rows == 0) return MMW_OK;
    const void *twA64;
    MMW_TRY(get_table<double>(ctx, TAB_TWIDDLE, A, &twA64));
    hipLaunchKernelGGL(k_argmax64_cells, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const cplx<double> *)d_cells, d_idx, n_rows, n_ant, A, shift, (const cplx<double> *)twA64);
    return check_launch("argmax64_cells");
}

int mmw_abs_c64(mmw_ctx *ctx, const void *d_in, float *d_out, size_t n) {
    MMW_REQUIRE(ctx && (n == 0 || (d_in && d_out)), "null argument");
    MMW_JOIN(ctx);
    return abs_c64(ctx, d_in, d_out, n);
}

int mmw_widen_f32_f64(mmw_ctx *ctx, const float *d_in, double *d_out, size_t n) {
    MMW_REQUIRE(ctx && (n == 0 || (d_in && d_out)), "null argument");
    MMW_JOIN(ctx);
    if (n == 0) return MMW_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx->num_cu * 16);
    hipLaunchKernelGGL(k_widen_f32_f64, dim3(grid), dim3(256), 0, ctx->stream, d_in, d_out, n);
    return check_launch("widen_f32_f64");
}

int mmw_diag_set_option(mmw_ctx *ctx, const char *name, int value) {
    MMW_REQUIRE(ctx && name && *name, "null argument");
    if (value == INT_MIN) ctx->opts.erase(name);
    else ctx->opts[name] = value;
    return MMW_OK;
}

int mmw_diag_rd_plan(int S, int C, int float64, int plan[8]) {
    MMW_REQUIRE(plan && S > 0 && C > 0, "bad argument");
    for (int i = 0; i < 8; ++i) plan[i] = 0;
    RdMixedPlan pl{};
    if (float64) {
        const bool pow2 = is_pow2(S) && is_pow2(C);
        plan[0] = (!pow2 && rd_mixed_plan(S, C, sizeof(cplx<double>), &pl)) ? 2 : 3;
    } else if (fused_rd_ok(S, C))
        plan[0] = 0;
    else if (rd_lds_supported(S, C))
        plan[0] = 1;
    else
        plan[0] = rd_mixed_plan(S, C, sizeof(cplx<float>), &pl) ? 2 : (rd_split_ct_supported(S, C) ? 4 : 3);
    if (plan[0] == 2) {
        plan[1] = pl.cls;
        plan[2] = pl.big;
        plan[3] = pl.s1;
        plan[4] = pl.s2;
        plan[5] = pl.c1;
        plan[6] = pl.c2;
        plan[7] = (int)pl.lds_bytes;
    }
    return MMW_OK;
}

int mmw_diag_chain_plan(mmw_ctx *ctx, int n_frames, int V, int S, int C, int A, int flags, int plan[8]) {
    MMW_REQUIRE(ctx && plan && n_frames > 0 && V > 0 && S > 0 && C > 0 && A >= V, "bad argument");
    const ChainPlan p = chain_plan(ctx, false, false, n_frames, V, S, C, A, flags);
    for (int i = 0; i < 8; ++i) plan[i] = 0;
    plan[0] = p.pipelined;
    plan[1] = p.chunk;
    plan[2] = p.ring;
    plan[3] = p.rd_cus;
    plan[4] = p.vskip > 2 ? V - 2 : V;      // range-Doppler planes transformed per frame
    plan[5] = ctx->chain_fallbacks;      // chain calls re-run on the event schedule after a hand-off timeout (mmw_chain_settle)
    plan[6] = p.sync;
    plan[7] = p.sync ? p.ring_frames : 0;
    return MMW_OK;
}

// Host-logic diagnostics that need no device (run under AddressSanitizer / UBSan by tests/cpp/host_sanitize.cpp):
// the chain schedule for a device with `num_cu` CUs, the tiling of the fused detection stage, the runs a chirp-z plan
// cuts a frequency list into.
int mmw_diag_chain_plan_nodev(int num_cu, int raw, int n_frames, int V, int S, int C, int A, int flags, int plan[8]) {
    MMW_REQUIRE(plan && num_cu > 0 && n_frames > 0 && V > 0 && S > 0 && C > 0 && A >= V, "bad argument");
    mmw_ctx fake;
    fake.num_cu = num_cu;
    const ChainPlan p = chain_plan(&fake, false, raw != 0, n_frames, V, S, C, A, flags);
    plan[0] = p.pipelined;
    plan[1] = p.chunk;
    plan[2] = p.ring;
    plan[3] = p.rd_cus;
    plan[4] = p.vskip > 2 ? V - 2 : V;
    plan[5] = 0;
    plan[6] = p.sync;
    plan[7] = p.sync ? p.ring_frames : 0;
    return MMW_OK;
}

int mmw_diag_detect_plan(int S, int C, int cfar_kind, int train_r, int train_d, int guard_r, int guard_d, int n_az, int n_el, int A,
                         int plan[8]) {
    MMW_REQUIRE(plan && S > 0 && C > 0 && train_r >= 0 && train_d >= 0 && guard_r >= 0 && guard_d >= 0 && n_az >= 0 && n_el >= 0,
                "bad argument");
    const DetectPlan p = detect_plan(S, C, cfar_kind, train_r, train_d, guard_r, guard_d, n_az, n_el, A);
    plan[0] = p.ok;
    plan[1] = p.ok ? 1 : 0;                                 // workgroups per frame (row tiles went with the band-streamed kernel)
    plan[2] = p.ok ? (int)(((long)DET_LOADS * DET_NT * 2) / C) : 0;      // rows one band's loads can carry (band + halo)
    plan[3] = p.band_rows;
    plan[4] = p.band_pitch;
    plan[5] = (int)p.lds_screen;
    plan[6] = p.ct_window;
    plan[7] = rd_error_ulps(S, C);
    return MMW_OK;
}

int mmw_diag_czt_runs(const double *h_freq, int M, int n_used, int *h_runs, int cap, int *n_runs) {
    MMW_REQUIRE(h_freq && n_runs && M > 0 && n_used > 0 && cap >= 0 && (cap == 0 || h_runs), "bad argument");
    const int L = czt_length(n_used);
    *n_runs = 0;
    if (L <= 0) return MMW_OK;              // no chirp-z length for this many input points
    const auto runs = czt_runs(h_freq, M, L - n_used + 1);
    *n_runs = (int)runs.size();
    for (int i = 0; i < *n_runs && i < cap; ++i) {          // (offset, length, filled with zeros) per run
        h_runs[3 * i] = std::get<0>(runs[i]);
        h_runs[3 * i + 1] = std::get<1>(runs[i]);
        h_runs[3 * i + 2] = std::get<2>(runs[i]) ? 1 : 0;
    }
    return MMW_OK;
}

int mmw_diag_mfma_peak(mmw_ctx *ctx, int kind, double *tflops) {
    MMW_REQUIRE(ctx && tflops && kind >= 0 && kind <= 7, "bad argument");
    MMW_JOIN(ctx);
    MMW_TRY(ensure_scratch(ctx, 256));
    // kinds 4 / 5: kinds 2 / 3 (vector FMAs between the MFMAs) with ONE wave per SIMD
    // kinds 6 / 7: v_mfma_f32_32x32x16_bf16 alone / with 8 float32 FMAs after every MFMA (two waves per SIMD)
    const int iters = 1 << 15, wgs = ctx->num_cu * (kind == 4 || kind == 5 ? 1 : 2);            // 8 waves per CU = 2 per SIMD
    if (kind == 4 || kind == 5) kind -= 2;
    hipLaunchKernelGGL(k_diag_mfma, dim3(wgs), dim3(256), 0, ctx->stream, (float *)ctx->scratch, 256, kind);    // warm-up
    MMW_HIP(hipEventRecord(ctx->t0, ctx->stream));
    hipLaunchKernelGGL(k_diag_mfma, dim3(wgs), dim3(256), 0, ctx->stream, (float *)ctx->scratch, iters, kind);
    MMW_TRY(check_launch("diag_mfma"));
    MMW_HIP(hipEventRecord(ctx->t1, ctx->stream));
    MMW_HIP(hipEventSynchronize(ctx->t1));
    float ms = 0.f;
    MMW_HIP(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    const double flops_per_mfma = kind >= 6 ? 2.0 * 32 * 32 * 16 : kind != 1 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
    *tflops = (double)wgs * 4 * iters * 4 * flops_per_mfma / (ms * 1e-3) / 1e12;
    return MMW_OK;
}

int mmw_diag_membw(mmw_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, int mode, int blocks) {
    MMW_REQUIRE(ctx && d_src && d_dst && bytes % 16 == 0 && mode >= 0 && mode <= 7, "bad argument");
    MMW_JOIN(ctx);
    if (blocks <= 0) blocks = ctx->num_cu * 8;
    hipLaunchKernelGGL(k_diag_membw, dim3(blocks), dim3(256), 0, ctx->stream, (const diag_f4 *)d_src,
                       (diag_f4 *)d_dst, bytes / 16, mode);
    return check_launch("diag_membw");
}

// ------------------------------------------------------------------ beamformers
int mmw_bartlett(mmw_ctx *ctx, const void *d_X, const double *d_P, const double *d_dirs, void *d_out, int n_frames, int S,
                 int E, int T, double lambda_m) {
    MMW_REQUIRE(ctx && d_X && d_P && d_dirs && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && n_frames <= 65535 && S > 0 && E > 0 && T > 0 && lambda_m > 0, "bad shape");
    if (n_frames == 0) return MMW_OK;
    return bartlett(ctx, d_X, d_P, d_dirs, d_out, n_frames, S, E, T, lambda_m);
}

int mmw_capon(mmw_ctx *ctx, const void *d_X, const double *h_thetas, float *d_out, int n_frames, int V, int R, int K,
              int T, double delta) {
    MMW_REQUIRE(ctx && d_X && h_thetas && d_out, "null argument");
    MMW_JOIN(ctx);
    MMW_REQUIRE(n_frames >= 0 && V > 0 && V <= 16 && R > 0 && K > 0 && T > 0, "bad shape (V <= 16)");
    if (n_frames == 0) return MMW_OK;
    return capon(ctx, d_X, h_thetas, d_out, n_frames, V, R, K, T, delta);
}

}  // extern "C"
