// Chirp-z (Bluestein) zoom transform: X[k] = sum_{n < n_used} x[n] exp(-j 2 pi n (f0 + k df)), k = 0 .. m-1, for rows of a
// cube -- what scipy.signal.ZoomFFT evaluates for DopplerAzimuthProcessor.zoom_fft (processors/doppler_azimuth_resp.py:
// 130-163, two calls, one per velocity sign) and RangeProcessor.zoom_fft (processors/range_resp.py:59-102).
//
// With n k = (n^2 + k^2 - (k - n)^2) / 2:
//     X[k] = b[k] * sum_n (x[n] a[n]) v[k - n],   a[n] = exp(-j 2 pi (f0 n + df n^2 / 2)),  b[k] = exp(-j pi df k^2),
//                                                 v[i] = exp(+j pi df i^2),  i = -(n_used - 1) .. m - 1
// i.e. one circular convolution of length L >= n_used + m - 1, done as FFT_L -> pointwise product with FFT_L(v) ->
// inverse FFT_L.  L = R * R (256 or 1024): R threads own one (row, segment) item, each holds R points; an L-point FFT is
// two R-point register FFTs around one LDS exchange, and because the output of that scheme (thread k1 holds X[k1 + R k2])
// is exactly the input layout of the next one (thread j holds x[j + R n1]) the pointwise product and the inverse
// transform (as conj FFT conj) follow without another exchange.  Round 1/2's direct form (k_zoom_rows: an n_used x m
// table, n_used complex MACs per output) cost 8 n_used m flops per row: 262 kflop at 128 chirps x 256 bins against
// ~50 kflop here.
//
// The frequency list of a call is split on the host into uniform runs ("segments", each <= L - n_used + 1 bins; NaN
// entries = bins the reference fills with zeros); the a / FFT(v) / b tables of a list are computed in long double and
// cached in the context, since a processor calls with the same list every frame.
#pragma once
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

#include "mmw_ctx.h"
#include "mmw_fft.h"

namespace mmw {

struct CztSegment {
    int out_off, m, zero, pad;      // output bins [out_off, out_off + m); zero: the reference writes zeros there
};

// host: split freq[0 .. M) into uniform runs of at most m_max bins
inline std::vector<std::tuple<int, int, bool, double, double>> czt_runs(const double *freq, int M, int m_max) {
    std::vector<std::tuple<int, int, bool, double, double>> runs;       // (offset, length, zero, f0, df)
    int k = 0;
    while (k < M) {
        if (freq[k] != freq[k]) {       // NaN run
            int e = k;
            while (e < M && freq[e] != freq[e] && e - k < m_max) ++e;
            runs.emplace_back(k, e - k, true, 0.0, 0.0);
            k = e;
            continue;
        }
        int e = k + 1;
        double df = 0.0;
        if (e < M && freq[e] == freq[e]) {
            df = freq[e] - freq[k];
            // the run continues while bin i sits on the line f0 + (i - k) df to within a few ulps of the list's values
            while (e < M && e - k < m_max && freq[e] == freq[e] &&
                   std::fabs(freq[e] - (freq[k] + (double)(e - k) * df)) <= 1e-13 * std::fmax(1.0, std::fabs(freq[e])))
                ++e;
        }
        runs.emplace_back(k, e - k, false, freq[k], df);
        k = e;
    }
    return runs;
}

// host: in-place radix-2 FFT, forward (e^{-j}), length a power of two
inline void host_fft(std::vector<std::complex<long double>> &a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const long double ang = -2.0L * M_PIl * (long double)k / (long double)len;
                const std::complex<long double> w(cosl(ang), sinl(ang));
                const auto u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

inline std::complex<long double> unit_turns(long double turns) {       // exp(j 2 pi turns), phase reduced first
    turns -= rintl(turns);
    return {cosl(2.0L * M_PIl * turns), sinl(2.0L * M_PIl * turns)};
}

// L for a list: the smallest of 256 / 1024 that leaves segments of at least 32 bins; 0 = use the direct kernels
inline int czt_length(int n_used) {
    if (n_used + 32 - 1 <= 256) return 256;
    if (n_used + 32 - 1 <= 1024) return 1024;
    return 0;
}

// Find or build the plan of (freq list, n_used).  Plans live in the context (at most 8; the oldest is dropped).
inline int czt_plan(mmw_ctx *ctx, const double *freq, int M, int n_used, const CztPlan **out) {
    const int L = czt_length(n_used);
    if (!L) return set_error(MMW_ERR_UNSUPPORTED, "no chirp-z length for %d input points", n_used);
    for (auto &p : ctx->czt_plans)
        if (p.n_used == n_used && (int)p.freq.size() == M && std::memcmp(p.freq.data(), freq, sizeof(double) * M) == 0) {
            *out = &p;
            return MMW_OK;
        }
    const auto runs = czt_runs(freq, M, L - n_used + 1);
    const int n_seg = (int)runs.size();
    std::vector<CztSegment> segs(n_seg);
    std::vector<float> tabs((size_t)n_seg * 3 * L * 2, 0.f);
    for (int s = 0; s < n_seg; ++s) {
        const auto [off, m, zero, f0, df] = runs[s];
        segs[s] = CztSegment{off, m, zero ? 1 : 0, 0};
        if (zero) continue;
        float *A = tabs.data() + (size_t)s * 3 * L * 2, *Vt = A + 2 * L, *B = Vt + 2 * L;
        for (int n = 0; n < n_used; ++n) {
            const long double nn = (long double)n;
            const auto a = unit_turns(-((long double)f0 * nn + (long double)df * nn * nn / 2.0L));
            A[2 * n] = (float)a.real();
            A[2 * n + 1] = (float)a.imag();
        }
        std::vector<std::complex<long double>> v(L, {0.0L, 0.0L});
        for (int i = -(n_used - 1); i < m; ++i) {
            const long double ii = (long double)i;
            v[(i + L) % L] = unit_turns((long double)df * ii * ii / 2.0L);
        }
        host_fft(v);
        for (int k = 0; k < L; ++k) {
            Vt[2 * k] = (float)v[k].real();
            Vt[2 * k + 1] = (float)v[k].imag();
        }
        for (int k = 0; k < m; ++k) {
            const long double kk = (long double)k;
            const auto b = unit_turns(-(long double)df * kk * kk / 2.0L) / (long double)L;     // 1 / L of the inverse FFT
            B[2 * k] = (float)b.real();
            B[2 * k + 1] = (float)b.imag();
        }
    }
    CztPlan p;
    p.freq.assign(freq, freq + M);
    p.n_used = n_used;
    p.L = L;
    p.n_seg = n_seg;
    const size_t seg_bytes = segs.size() * sizeof(CztSegment), tab_bytes = tabs.size() * sizeof(float);
    if (hipMalloc(&p.d_segs, seg_bytes) != hipSuccess || hipMalloc(&p.d_tabs, tab_bytes) != hipSuccess) {
        if (p.d_segs) (void)hipFree(p.d_segs);
        return set_error(MMW_ERR_NOMEM, "hipMalloc for chirp-z tables failed");
    }
    hipError_t e = hipMemcpyAsync(p.d_segs, segs.data(), seg_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p.d_tabs, tabs.data(), tab_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);        // the staging vectors die with this scope
    if (e != hipSuccess) {                                              // nothing of a half-built plan stays allocated
        (void)hipFree(p.d_segs);
        (void)hipFree(p.d_tabs);
        return set_error(MMW_ERR_HIP, "upload of the chirp-z tables failed: %s", hipGetErrorString(e));
    }
    if (ctx->czt_plans.size() >= 8) {
        (void)hipFree(ctx->czt_plans.front().d_segs);   // the stream was just drained: nothing uses the oldest plan
        (void)hipFree(ctx->czt_plans.front().d_tabs);
        ctx->czt_plans.pop_front();
    }
    ctx->czt_plans.push_back(std::move(p));             // (a deque: plans handed out earlier stay where they are)
    *out = &ctx->czt_plans.back();
    return MMW_OK;
}

struct CztArgs {
    const float2 *x;
    long outer_stride, inner_stride, elem_stride;   // element i of row r: x[(r / s_keep) outer + (s_lo + r % s_keep) inner + i elem]
    int s_lo, s_keep;
    const float *win;           // [n_used] or nullptr
    int n_used, n_seg, M;
    long rows;
    const CztSegment *segs;
    const cplx<float> *tabs;    // per segment: a[L], FFT(v)[L], b[L]
    const cplx<float> *tw;      // W_L^i
    float2 *out;                // [rows][M]
};

// grid = (row blocks, segments): a workgroup keeps ONE segment's a / FFT(v) / b tables in the LDS and each thread its 16
// (32) inter-pass twiddles in registers, and walks blocks of 256 / R rows -- per item only the row itself comes from memory
// (the first version fetched ~90 table words per thread per item through L1/L2 and ran at 7 TFLOP/s).
template <int R>
__global__ __launch_bounds__(256, R == 16 ? 4 : 1) void k_czt_rows(CztArgs p) {
    constexpr int L = R * R, IPW = 256 / R, P = R + 1;
    extern __shared__ __attribute__((aligned(16))) char czt_smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(czt_smem);       // [IPW][R][P] exchange, a, FFT(v), b, twiddles
    cplx<float> *A0 = lds + IPW * R * P;
    const int t = threadIdx.x, j = t % R, it = t / R;
    const int seg = blockIdx.y;
    const CztSegment sd = p.segs[seg];
    {
        const cplx<float> *g = p.tabs + (size_t)seg * 3 * L;
        for (int i = t; i < 3 * L; i += 256) A0[i] = g[i];
    }
    cplx<float> *twT = A0 + 3 * L;          // [k1][j] = W_L^(j k1): lanes read consecutive words
    for (int i = t; i < L; i += 256) twT[i] = p.tw[(i / R) * (i % R)];
    __syncthreads();
    cplx<float> *ex = lds + it * R * P;
    for (long row0 = (long)blockIdx.x * IPW; row0 < p.rows; row0 += (long)gridDim.x * IPW) {
        // keep the table reads inside the loop (an opaque copy of the lane index): hoisted, a thread's loop-invariant
        // table entries would sit in ~130 VGPRs and halve the occupancy
        int jo = j;
        asm volatile("" : "+v"(jo));
        const cplx<float> *A = A0 + (jo - j), *Vt = A + L, *B = Vt + L;
        const long row = row0 + it;
        const bool valid = row < p.rows;
        float2 *dst = p.out + row * p.M + sd.out_off;
        if (sd.zero) {                  // bins the reference fills with zeros (block-uniform branch)
            if (valid)
                for (int k = j; k < sd.m; k += R) dst[k] = make_float2(0.f, 0.f);
            continue;
        }
        const long fv = (valid ? row : 0) / p.s_keep;
        const float2 *src = p.x + fv * p.outer_stride + (p.s_lo + ((valid ? row : 0) - fv * p.s_keep)) * p.inner_stride;
        cplx<float> a[R], c[R];
        // y[n] = x[n] win[n] a[n], n = R n1 + j (zero beyond n_used)
#pragma unroll
        for (int n1 = 0; n1 < R; ++n1) {
            const int n = R * n1 + j;
            cplx<float> v = cplx<float>{0.f, 0.f};
            if (valid && n < p.n_used) {
                const float2 xv = src[(long)n * p.elem_stride];
                const float w = p.win ? p.win[n] : 1.f;
                v = cmul(cplx<float>{xv.x * w, xv.y * w}, A[n]);
            }
            a[n1] = v;
        }
        RegFFT<R, float>::run(a);
        static_for<R>([&](auto K1) {
            constexpr int k1 = decltype(K1)::value;
            ex[k1 * P + j] = cmul(a[bitrev<R>(k1)], twT[k1 * R + jo]);
        });
        __syncthreads();
#pragma unroll
        for (int n2 = 0; n2 < R; ++n2) c[n2] = ex[j * P + n2];
        RegFFT<R, float>::run(c);           // Y[j + R k2] = c[bitrev(k2)]
        // pointwise product, conjugated: the inverse transform is conj(FFT(conj(.)))
        static_for<R>([&](auto K2) {
            constexpr int k2 = decltype(K2)::value;
            const cplx<float> y = cmul(c[bitrev<R>(k2)], Vt[j + R * k2]);
            a[k2] = cplx<float>{y.x, -y.y};
        });
        RegFFT<R, float>::run(a);
        __syncthreads();                    // every thread has read its row of the first exchange
        static_for<R>([&](auto K1) {
            constexpr int k1 = decltype(K1)::value;
            ex[k1 * P + j] = cmul(a[bitrev<R>(k1)], twT[k1 * R + jo]);
        });
        __syncthreads();
#pragma unroll
        for (int n2 = 0; n2 < R; ++n2) c[n2] = ex[j * P + n2];
        RegFFT<R, float>::run(c);
        if (valid) {
            static_for<R>([&](auto K2) {
                constexpr int k2 = decltype(K2)::value;
                const int k = j + R * k2;
                if (k < sd.m) {
                    const cplx<float> z = c[bitrev<R>(k2)];
                    const cplx<float> o = cmul(cplx<float>{z.x, -z.y}, B[k]);
                    dst[k] = make_float2(o.x, o.y);
                }
            });
        }
        __syncthreads();                    // the exchange buffer is rewritten by the next block of rows
    }
}

inline int launch_czt_rows(mmw_ctx *ctx, const CztPlan &plan, CztArgs a) {
    a.n_used = plan.n_used;
    a.n_seg = plan.n_seg;
    a.segs = (const CztSegment *)plan.d_segs;
    a.tabs = (const cplx<float> *)plan.d_tabs;
    const void *tw;
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, plan.L, &tw));
    a.tw = (const cplx<float> *)tw;
    if (a.rows == 0) return MMW_OK;
    // a few resident workgroups per CU and segment, each walking blocks of rows
    auto grid_for = [&](int ipw) {
        const long blocks = (a.rows + ipw - 1) / ipw;
        const long cap = std::max<long>(1, (long)ctx->num_cu * 8 / plan.n_seg);
        return dim3((unsigned)std::min(blocks, cap), (unsigned)plan.n_seg);
    };
    if (plan.L == 256) {
        constexpr size_t lds_bytes = (16 * 16 * 17 + 4 * 256) * sizeof(cplx<float>);
        hipLaunchKernelGGL(k_czt_rows<16>, grid_for(16), dim3(256), lds_bytes, ctx->stream, a);
    } else {
        constexpr size_t lds_bytes = (8 * 32 * 33 + 4 * 1024) * sizeof(cplx<float>);     // 98 KB: above the default dynamic limit
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_czt_rows<32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes));       // per device, so on every call
        hipLaunchKernelGGL(k_czt_rows<32>, grid_for(8), dim3(256), lds_bytes, ctx->stream, a);
    }
    return check_launch("czt_rows");
}

}  // namespace mmw
