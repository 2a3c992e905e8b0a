// LDS-resident range-Doppler kernel for ANY plane shape that fits the LDS (S * C up to ~20k cells).
//
// The TI configs the reference ships (configs/*.cfg) produce 63/70/90/100/120/127/130/200/254 samples and
// 30/32/40/50/70/80/100/115/126/127 loops -- none of them a power of two -- so these are the
// shapes a user of the reference actually runs.  One workgroup owns one [S][C] plane (frame x antenna):
//   load (Hann(S) x Hann(C) folded in) -> range DFT down the columns -> Doppler DFT along the rows ->
//   store with the Doppler fftshift folded into the index                  (range_doppler_resp.py:94-103)
// Each axis of length N = N1 * N2 runs as two Cooley-Tukey levels of small DIRECT DFTs (n = N2 n1 + n2,
// k = k1 + N1 k2):  level A does the N1-point DFTs over n1 and multiplies by W_N^(n2 k1), level B the N2-point DFTs
// over n2.  A thread takes one group of R points into registers, so a level is in place in the LDS without
// hazards, and evaluates the R outputs one after the other from the R x R DFT matrix.  The matrix row is the same
// for every lane of the wave (the output loop is uniform), so it arrives through the scalar cache and the inner
// loop is pure FMA on VGPR x SGPR operands.  Cost per cell: (S1 + S2 + C1 + C2) complex MACs instead of the
// S + C of a one-level direct DFT: 36 instead of 163 for the 63 x 100 plane.
// Radix classes keep the register budget tight: class 0 = radices <= 16, class 1 = radices <= 32 (20, 23, 25),
// class 2 = radices <= 16 plus the prime 127 (63 x 127, 127 x 32, 254 x 50 planes).
#pragma once
#include "mmw_fft_generic.h"

namespace mmw {

struct RdMixedArgs {
    const void *in;          // cplx<float> planes
    void *out;               // cplx<T> planes, or T |.| planes when MAG
    long in_plane_stride;    // complex elements between consecutive input planes
    int S, C, Cp;            // plane shape and LDS row pitch (odd, so row-strided accesses spread over the banks)
    int s1, s2, c1, c2;      // S = s1 * s2, C = c1 * c2 (second factor 1 = single level)
    const void *win_s, *win_c;          // T[S], T[C]
    const void *tw_s, *tw_c;            // cplx<T>[S], cplx<T>[C]: W_N^m
    const void *m_s1, *m_s2, *m_c1, *m_c2;   // cplx<T>[R][R] DFT matrices
};

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// One level: groups (o, i), i fastest over the lanes; element j of a group sits at lds[i*inner_stride +
// o*outer_stride + j*estride].  Output k replaces element k, times tw[(o * k) mod N] when tw != nullptr.
template <int R, typename T>
__device__ __forceinline__ void dft_level(cplx<T> *lds, int tid, int n_inner, int inner_stride, int n_outer,
                                          int outer_stride, int estride, const cplx<T> *__restrict__ Wm,
                                          const cplx<T> *__restrict__ tw, int N) {
    const int n_groups = n_inner * n_outer;
    for (int g = tid; g < n_groups; g += 256) {
        const int o = g / n_inner, i = g - o * n_inner;
        cplx<T> *p = lds + i * inner_stride + o * outer_stride;
        cplx<T> x[R];
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = p[j * estride];
        int idx = 0;
#pragma unroll 1
        for (int k = 0; k < R; ++k) {
            const cplx<T> *w = Wm + k * R;      // uniform address: scalar loads
            // acc += x * w as two packed FMAs: (x.re, x.re) * (w.re, w.im) + (-x.im, x.im) * (w.im, w.re);
            // two independent chains (even / odd j); W^0 = 1 for j = 0
            cplx<T> a0 = x[0], a1 = cplx<T>{(T)0, (T)0};
#pragma unroll
            for (int j = 1; j < R; ++j) {
                const cplx<T> wj = w[j];
                const cplx<T> xr = cplx<T>{x[j].x, x[j].x}, xi = cplx<T>{-x[j].y, x[j].y}, ws = cplx<T>{wj.y, wj.x};
                if (j & 1) {
                    a1 = __builtin_elementwise_fma(xr, wj, a1);
                    a1 = __builtin_elementwise_fma(xi, ws, a1);
                } else {
                    a0 = __builtin_elementwise_fma(xr, wj, a0);
                    a0 = __builtin_elementwise_fma(xi, ws, a0);
                }
            }
            cplx<T> acc = a0 + a1;
            if (tw) {
                acc = cmul(acc, tw[idx]);
                idx += o;
                if (idx >= N) idx -= N;
            }
            p[k * estride] = acc;
        }
    }
}

template <int CLS, typename T, typename... A> __device__ __forceinline__ void dft_level_rt(int R, A... a) {
#define MMW_R(r) case r: dft_level<r, T>(a...); break;
    if (R <= 16) {
        switch (R) {
            MMW_R(2) MMW_R(3) MMW_R(4) MMW_R(5) MMW_R(6) MMW_R(7) MMW_R(8) MMW_R(9) MMW_R(10) MMW_R(11) MMW_R(12)
            MMW_R(13) MMW_R(14) MMW_R(15) MMW_R(16)
            default: break;                     // R == 1: identity
        }
    } else if constexpr (CLS == 1) {
        switch (R) {
            MMW_R(17) MMW_R(18) MMW_R(19) MMW_R(20) MMW_R(21) MMW_R(22) MMW_R(23) MMW_R(24) MMW_R(25) MMW_R(26)
            MMW_R(27) MMW_R(28) MMW_R(29) MMW_R(30) MMW_R(31) MMW_R(32)
            default: break;
        }
    } else if constexpr (CLS == 2) {
        if (R == 127) dft_level<127, T>(a...);
    }
#undef MMW_R
}

template <typename T, int CLS, bool MAG>
__global__ __launch_bounds__(256) void k_rd_mixed(RdMixedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T> *lds = reinterpret_cast<cplx<T> *>(smem);          // [S][Cp]
    const int tid = threadIdx.x;
    const int S = a.S, C = a.C, Cp = a.Cp, cells = S * C;
    const cplx<float> *in = reinterpret_cast<const cplx<float> *>(a.in) + (long)blockIdx.x * a.in_plane_stride;
    const T *ws = reinterpret_cast<const T *>(a.win_s), *wc = reinterpret_cast<const T *>(a.win_c);
    for (int e = tid; e < cells; e += 256) {
        const int s = e / C, c = e - s * C;
        const cplx<float> v = __builtin_nontemporal_load(in + e);
        const T w = ws[s] * wc[c];
        lds[s * Cp + c] = cplx<T>{(T)v.x * w, (T)v.y * w};
    }
    __syncthreads();
    typedef const cplx<T> *CP;
    // range axis: element s = s2 * n1 + n2 lives in row s
    dft_level_rt<CLS, T>(a.s1, lds, tid, C, 1, a.s2, Cp, a.s2 * Cp, (CP)a.m_s1, a.s2 > 1 ? (CP)a.tw_s : (CP) nullptr, S);
    __syncthreads();
    if (a.s2 > 1) {
        dft_level_rt<CLS, T>(a.s2, lds, tid, C, 1, a.s1, a.s2 * Cp, Cp, (CP)a.m_s2, (CP) nullptr, S);
        __syncthreads();
    }
    // Doppler axis: element c = c2 * m1 + m2 lives in column c; lanes walk the rows (pitch Cp is odd)
    dft_level_rt<CLS, T>(a.c1, lds, tid, S, Cp, a.c2, 1, a.c2, (CP)a.m_c1, a.c2 > 1 ? (CP)a.tw_c : (CP) nullptr, C);
    __syncthreads();
    if (a.c2 > 1) {
        dft_level_rt<CLS, T>(a.c2, lds, tid, S, Cp, a.c1, a.c2, 1, (CP)a.m_c2, (CP) nullptr, C);
        __syncthreads();
    }
    // bin k = k1 + s1 k2 sits in row s2 k1 + k2 (same along the Doppler axis); fftshift: out[(d + C/2) % C] = X[d]
    const long obase = (long)blockIdx.x * cells;
    const int half = C / 2;
    for (int e = tid; e < cells; e += 256) {
        const int k = e / C, dd = e - k * C;
        int d = dd - half;
        if (d < 0) d += C;
        const int k2 = k / a.s1, k1 = k - k2 * a.s1;
        const int d2 = d / a.c1, d1 = d - d2 * a.c1;
        const cplx<T> v = lds[(a.s2 * k1 + k2) * Cp + a.c2 * d1 + d2];
        if constexpr (MAG)
            reinterpret_cast<T *>(a.out)[obase + e] = mag<T>(v);
        else
            __builtin_nontemporal_store(v, reinterpret_cast<cplx<T> *>(a.out) + obase + e);
    }
}

// ------------------------------------------------------------------ host side: factorisation and launch
inline bool mixed_radix_ok(int r, int cls) {
    if (r <= 16) return true;
    if (cls == 1) return r <= 32;
    if (cls == 2) return r == 127;
    return false;
}

// N = n1 * n2 with both radices in the class, least (n1 + n2) plus a per-level overhead; n2 == 1: one level.
inline bool mixed_axis(int N, int cls, int *n1, int *n2) {
    int best = -1;
    for (int a = 1; a <= N; ++a) {
        if (N % a) continue;
        const int b = N / a;
        if (a < b || !mixed_radix_ok(a, cls) || !mixed_radix_ok(b, cls)) continue;
        const int cost = (b == 1) ? a + 3 : a + b + 6;
        if (best < 0 || cost < best) {
            best = cost;
            *n1 = a;
            *n2 = b;
        }
    }
    return best >= 0;
}

struct RdMixedPlan {
    int cls, s1, s2, c1, c2, Cp;
    size_t lds_bytes;
};

inline bool rd_mixed_plan(int S, int C, size_t elem_bytes, RdMixedPlan *pl) {
    if (S < 1 || C < 1 || (long)S * C > (1 << 20)) return false;
    pl->Cp = C | 1;
    pl->lds_bytes = (size_t)S * pl->Cp * elem_bytes;
    if (pl->lds_bytes > 160 * 1024) return false;
    const int n_cls = (elem_bytes > 8 || !tune_int("MMW_MIXED_CLS2", 0)) ? 2 : 3;   // radix 127 spills: off by default
    for (int cls = 0; cls < n_cls; ++cls)
        if (mixed_axis(S, cls, &pl->s1, &pl->s2) && mixed_axis(C, cls, &pl->c1, &pl->c2)) {
            pl->cls = cls;
            return true;
        }
    return false;
}

inline bool rd_mixed_supported(int S, int C) {
    RdMixedPlan pl;
    return rd_mixed_plan(S, C, sizeof(cplx<float>), &pl);
}

// planes x [S][C] complex64 at d_in (plane pitch in_plane_stride elements) -> d_out planes, contiguous:
// T = float: complex64 spectrum; T = double, MAG: float64 magnitude (the CFAR plane).
template <typename T, bool MAG>
int launch_rd_mixed(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C) {
    RdMixedPlan pl;
    if (!rd_mixed_plan(S, C, sizeof(cplx<T>), &pl))
        return set_error(MMW_ERR_UNSUPPORTED, "no mixed-radix RD plan for %dx%d", S, C);
    RdMixedArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.in_plane_stride = in_plane_stride;
    a.S = S;
    a.C = C;
    a.Cp = pl.Cp;
    a.s1 = pl.s1;
    a.s2 = pl.s2;
    a.c1 = pl.c1;
    a.c2 = pl.c2;
    MMW_TRY(get_table<T>(ctx, TAB_HANN, S, &a.win_s));
    MMW_TRY(get_table<T>(ctx, TAB_HANN, C, &a.win_c));
    MMW_TRY(get_table<T>(ctx, TAB_TWIDDLE, S, &a.tw_s));
    MMW_TRY(get_table<T>(ctx, TAB_TWIDDLE, C, &a.tw_c));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.s1, &a.m_s1));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.s2, &a.m_s2));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.c1, &a.m_c1));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.c2, &a.m_c2));
    auto go = [&](auto kern) -> int {
        if (pl.lds_bytes > 64 * 1024)
            MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)pl.lds_bytes));
        hipLaunchKernelGGL(kern, dim3(planes), dim3(256), pl.lds_bytes, ctx->stream, a);
        return check_launch("rd_mixed");
    };
    switch (pl.cls) {
        case 0: return go(k_rd_mixed<T, 0, MAG>);
        case 1: return go(k_rd_mixed<T, 1, MAG>);
        default:
            if constexpr (sizeof(T) == 4) return go(k_rd_mixed<T, 2, MAG>);
            return set_error(MMW_ERR_UNSUPPORTED, "radix class 2 is float32 only");
    }
}

}  // namespace mmw
