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
// Radix classes keep the register budget tight: class 0 = radices <= 16, class 1 = radices <= 32 (20, 23, 25);
// the planner (rd_mixed_plan) prices every factorisation and picks the cheapest kernel variant.
// A factor above 32 (the prime 127 of the 63 x 127, 127 x 32 and 254 x 50 planes) cannot sit in registers; such a
// level (dft_level_big) keeps its inputs in the LDS, gives every wave a block of 8 outputs whose matrix entries are
// again wave-uniform scalars, and goes through the spare LDS behind the plane to stay hazard-free.
#pragma once
#include <algorithm>
#include "mmw_fft_generic.h"
#include "mmw_fft_fused.h"
#include "mmw_dft_small.h"

namespace mmw {

// Rader's algorithm for a prime radix P > 32 (the 127 of the 63x127, 127x32 and 254x50 cfg planes): with a primitive
// root g, X[g^-q] = x[0] + sum_m x[g^m] W_P^(g^(m-q)) is a cyclic convolution of length L = P - 1 = r1 * r2, evaluated
// as inverse-DFT( DFT_L(a) * DFT_L(b) ) with the same small-radix levels as everything else:
// 2 (r1 + r2) + 3 complex MACs per point instead of P.
struct RaderTab {
    int P, r1, r2;               // P == 0: no Rader plan for this level (dft_level_big handles it)
    const int *idx;              // idx[n] = m with g^m = n (n = 1..P-1), idx[P + n] = q with g^-q = n
    const void *B;               // cplx<T>[L]: DFT_L(b) / L in slot order (slot r2 k1 + k2 = bin k1 + r1 k2)
    const void *twL;             // cplx<T>[L]: W_L^m
    const void *m_r1, *m_r2;     // DFT matrices of the two convolution radices
};

struct RdMixedArgs {
    const void *in;          // cplx<float> planes
    void *out;               // cplx<T> planes, or T |.| planes when MAG
    long in_plane_stride;    // complex elements between consecutive input planes
    int S, C, Cp;            // plane shape and LDS row pitch (odd, so row-strided accesses spread over the banks)
    int s1, s2, c1, c2;      // S = s1 * s2, C = c1 * c2 (second factor 1 = single level)
    const void *win_s, *win_c;          // T[S], T[C]
    const void *tw_s, *tw_c;            // cplx<T>[S], cplx<T>[C]: W_N^m
    const void *m_s1, *m_s2, *m_c1, *m_c2;   // cplx<T>[R][R] DFT matrices
    RaderTab rad_s, rad_c;   // Rader plans of the first range / Doppler level (radix s1 / c1 > 32)
    int tmp_cells;           // cells of spare LDS behind the plane
    RawView raw;             // ntx > 1: `in` is the raw [F][nrx][S][ntx * C] cube (in_plane_stride unused)
    long planes;             // raw only: planes in the launch (the grid is padded, see raw_block_plane)
    unsigned mg_C, mg_S, mg_s1, mg_c1;        // ceil(2^32 / d): e / d == umulhi(e, mg) for e < 2^32 / d (d > 1)
};

__device__ __forceinline__ int fast_div(int e, unsigned magic, int d) {
    return d == 1 ? e : (int)__umulhi((unsigned)e, magic);
}
__host__ __device__ inline unsigned div_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// One level: groups (o, i), i fastest over the lanes; element j of a group sits at lds[i*inner_stride +
// o*outer_stride + j*estride].  Output k replaces element k, times tw[(o * k) mod N] when tw != nullptr.
// The R points go through RegDFT<R> (mmw_dft_small.h): radix-2 networks, real-symmetric prime DFTs, prime-factor /
// Cooley-Tukey splits -- all at compile time, 3-4x fewer multiply-adds than the R x R matrix product of round 1
// (Wm, the DFT matrix, is only used by the run-time-radix levels now).
template <int R, int NT, typename T>
__device__ __forceinline__ void dft_level(cplx<T> *lds, int tid, int n_inner, unsigned mg_inner, int inner_stride, int n_outer,
                                          int outer_stride, int estride, const cplx<T> *__restrict__ Wm,
                                          const cplx<T> *__restrict__ tw, int N) {
    (void)Wm;
    const int n_groups = n_inner * n_outer;
    for (int g = tid; g < n_groups; g += NT) {
        const int o = fast_div(g, mg_inner, n_inner), i = g - o * n_inner;
        cplx<T> *p = lds + i * inner_stride + o * outer_stride;
        cplx<T> x[R];
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = p[j * estride];
        RegDFT<R, T>::run(x);
        if (tw) {
            int idx = o;                            // (o * k) mod N, k = 1 .. R-1 (o < N)
#pragma unroll
            for (int k = 1; k < R; ++k) {
                x[k] = cmul(x[k], tw[idx]);
                idx += o;
                if (idx >= N) idx -= N;
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) p[k * estride] = x[k];
    }
}

// Same contract as dft_level for a run-time radix R > 32.  Outputs of up to tmp_cells / R groups at a time are
// collected in tmp[group][k] and copied back over the inputs after a barrier.  Wm is symmetric, so row j holds
// W^(jk) for consecutive k; rows are read up to 7 entries past R (the table is padded).
constexpr int BIG_KB = 8;
template <int NT, typename T>
__device__ __forceinline__ void dft_level_big(int R, cplx<T> *lds, cplx<T> *tmp, int tmp_cells, int tid, int n_inner,
                                              unsigned mg_inner, int inner_stride, int n_outer, int outer_stride,
                                              int estride, const cplx<T> *__restrict__ Wm,
                                              const cplx<T> *__restrict__ tw, int N) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n_groups = n_inner * n_outer;
    int chunk = tmp_cells / R;
    if (chunk > n_groups) chunk = n_groups;
    if (chunk < 1) return;      // the planner reserves at least 32 groups
    for (int g0 = 0; g0 < n_groups; g0 += chunk) {
        const int ng = (n_groups - g0 < chunk) ? n_groups - g0 : chunk;
        for (int k0 = wave * BIG_KB; k0 < R; k0 += (NT / 64) * BIG_KB) {
            for (int gl = lane; gl < ng; gl += 64) {
                const int g = g0 + gl;
                const int o = fast_div(g, mg_inner, n_inner), i = g - o * n_inner;
                const cplx<T> *p = lds + i * inner_stride + o * outer_stride;
                cplx<T> acc[BIG_KB];
#pragma unroll
                for (int kb = 0; kb < BIG_KB; ++kb) acc[kb] = cplx<T>{(T)0, (T)0};
#pragma unroll 2
                for (int j = 0; j < R; ++j) {
                    const cplx<T> xj = p[j * estride];
                    const cplx<T> xr = cplx<T>{xj.x, xj.x}, xi = cplx<T>{-xj.y, xj.y};
                    const cplx<T> *w = Wm + (long)j * R + k0;       // wave-uniform: scalar loads
#pragma unroll
                    for (int kb = 0; kb < BIG_KB; ++kb) {
                        const cplx<T> wk = w[kb];
                        acc[kb] = __builtin_elementwise_fma(xr, wk, acc[kb]);
                        acc[kb] = __builtin_elementwise_fma(xi, cplx<T>{wk.y, wk.x}, acc[kb]);
                    }
                }
#pragma unroll
                for (int kb = 0; kb < BIG_KB; ++kb) {
                    const int k = k0 + kb;
                    if (k < R) {
                        cplx<T> v = acc[kb];
                        if (tw) v = cmul(v, tw[(o * k) % N]);
                        tmp[gl * R + k] = v;
                    }
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < ng * R; e += NT) {
            const int gl = e / R, k = e - gl * R;
            const int g = g0 + gl;
            const int o = fast_div(g, mg_inner, n_inner), i = g - o * n_inner;
            lds[i * inner_stride + o * outer_stride + k * estride] = tmp[e];
        }
        __syncthreads();
    }
}

template <int CLS, int NT, typename T, typename... A> __device__ __forceinline__ void dft_level_rt(int R, A... a) {
#define MMW_R(r) case r: dft_level<r, NT, T>(a...); break;
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
    }
#undef MMW_R
}

// Same contract as dft_level for a prime radix with a Rader plan.  Works on up to tmp_cells / (P + 1) groups at a
// time: gather the group in generator order into tmp[m][group] (+ x[0] and, later, the sum), run the forward levels
// there, multiply by DFT(b)/L and conjugate, run the levels again in decimation-in-time order (an inverse DFT up to
// the conjugations), and scatter x[0] + conj(.) back over the inputs in output order.
template <int CLS, int NT, typename T>
__device__ __forceinline__ void dft_level_rader(const RaderTab &rt, cplx<T> *lds, cplx<T> *tmp, int tmp_cells, int tid,
                                                int n_inner, unsigned mg_inner, int inner_stride, int n_outer,
                                                int outer_stride, int estride, const cplx<T> *__restrict__ tw, int N) {
    const int P = rt.P, L = P - 1, r1 = rt.r1, r2 = rt.r2;
    const int n_groups = n_inner * n_outer;
    int chunk = tmp_cells / (P + 1);
    if (chunk > n_groups) chunk = n_groups;
    if (chunk < 1) return;
    const cplx<T> *B = reinterpret_cast<const cplx<T> *>(rt.B), *twL = reinterpret_cast<const cplx<T> *>(rt.twL);
    for (int g0 = 0; g0 < n_groups; g0 += chunk) {
        const int ng = (n_groups - g0 < chunk) ? n_groups - g0 : chunk;
        const unsigned mg = div_magic(ng);
        cplx<T> *x0 = tmp + L * ng, *sum = x0 + ng;        // [ng] each, behind the [L][ng] work array
        for (int e = tid; e < ng * P; e += NT) {           // gather, lanes along the groups
            const int n = fast_div(e, mg, ng), gl = e - n * ng;
            const int g = g0 + gl;
            const int o = fast_div(g, mg_inner, n_inner), i = g - o * n_inner;
            const cplx<T> v = lds[i * inner_stride + o * outer_stride + n * estride];
            if (n == 0) x0[gl] = v;
            else tmp[rt.idx[n] * ng + gl] = v;
        }
        __syncthreads();
        // forward: level A over m1 (radix r1, slots r2 m1 + m2), level B over m2; then pointwise; then the same two
        // radices in the opposite order (slots r2 k1 + k2 -> natural q = r2 n1 + n2)
        for (int step = 0; step < 4; ++step) {
            const bool first = (step == 0 || step == 3);   // the radix-r1 levels
            const int R = first ? r1 : r2;
            if (R > 1) {
                const int n_out = first ? r2 : r1, o_stride = first ? ng : r2 * ng, e_stride = first ? r2 * ng : ng;
                const bool twiddle = r2 > 1 && (step == 0 || step == 2);
                dft_level_rt<CLS, NT, T>(R, tmp, tid, ng, mg, 1, n_out, o_stride, e_stride,
                                         reinterpret_cast<const cplx<T> *>(first ? rt.m_r1 : rt.m_r2),
                                         twiddle ? twL : (const cplx<T> *)nullptr, L);
                __syncthreads();
            }
            if (step == 1) {
                for (int e = tid; e < ng * L; e += NT) {
                    const int sl = fast_div(e, mg, ng), gl = e - sl * ng;
                    const cplx<T> a = tmp[e];
                    if (sl == 0) sum[gl] = x0[gl] + a;      // bin 0 of DFT(a) is the sum of x[1..P-1]
                    const cplx<T> v = cmul(a, B[sl]);
                    tmp[e] = cplx<T>{v.x, -v.y};
                }
                __syncthreads();
            }
        }
        for (int e = tid; e < ng * P; e += NT) {           // scatter in output order, inter-level twiddle folded in
            const int n = fast_div(e, mg, ng), gl = e - n * ng;
            const int g = g0 + gl;
            const int o = fast_div(g, mg_inner, n_inner), i = g - o * n_inner;
            cplx<T> v;
            if (n == 0) v = sum[gl];
            else {
                const cplx<T> c = tmp[rt.idx[P + n] * ng + gl];
                v = x0[gl] + cplx<T>{c.x, -c.y};
            }
            if (tw) v = cmul(v, tw[(o * n) % N]);
            lds[i * inner_stride + o * outer_stride + n * estride] = v;
        }
        __syncthreads();
    }
}

template <int CLS, bool BIG, int NT, typename T>
__device__ __forceinline__ void dft_level_any(int R, const RaderTab &rt, cplx<T> *lds, cplx<T> *tmp, int tmp_cells, int tid,
                                              int n_inner, unsigned mg_inner, int inner_stride, int n_outer,
                                              int outer_stride, int estride, const cplx<T> *Wm, const cplx<T> *tw, int N) {
    if constexpr (BIG) {
        if (R > 32) {       // both end with their own barrier
            if (rt.P == R) {
                dft_level_rader<CLS, NT, T>(rt, lds, tmp, tmp_cells, tid, n_inner, mg_inner, inner_stride, n_outer, outer_stride,
                                            estride, tw, N);
                return;
            }
            dft_level_big<NT, T>(R, lds, tmp, tmp_cells, tid, n_inner, mg_inner, inner_stride, n_outer, outer_stride, estride, Wm, tw, N);
            return;
        }
    }
    dft_level_rt<CLS, NT, T>(R, lds, tid, n_inner, mg_inner, inner_stride, n_outer, outer_stride, estride, Wm, tw, N);
    __syncthreads();
}

template <typename T, int CLS, bool BIG, bool MAG, int NT>
__global__ __launch_bounds__(NT) void k_rd_mixed(RdMixedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T> *lds = reinterpret_cast<cplx<T> *>(smem);          // [S][Cp]
    const int tid = threadIdx.x;
    const int S = a.S, C = a.C, Cp = a.Cp, cells = S * C;
    const int ntx = a.raw.ntx > 1 ? a.raw.ntx : 1;       // element (s, c) at in[(s * C + c) * ntx]
    long plane = blockIdx.x;
    if (a.raw.ntx > 1) {
        plane = raw_block_plane(blockIdx.x, a.planes, a.raw);
        if (plane < 0 || skip_raw_plane(plane, a.raw)) return;
    } else plane = skip_block_plane(blockIdx.x, a.raw);
    const cplx<float> *in = a.raw.ntx > 1
        ? raw_plane(reinterpret_cast<const cplx<float> *>(a.in), plane, S, C, a.raw)
        : reinterpret_cast<const cplx<float> *>(a.in) + plane * a.in_plane_stride;
    const T *ws = reinterpret_cast<const T *>(a.win_s), *wc = reinterpret_cast<const T *>(a.win_c);
    for (int e = tid; e < cells; e += NT) {
        const int s = fast_div(e, a.mg_C, C), c = e - s * C;
        const cplx<float> v = __builtin_nontemporal_load(in + (long)e * ntx);
        const T w = ws[s] * wc[c];
        lds[s * Cp + c] = cplx<T>{(T)v.x * w, (T)v.y * w};
    }
    // inter-level twiddles W_S^m, W_C^m next to the plane: an LDS read per output instead of a global one
    cplx<T> *tw_s = lds + S * Cp, *tw_c = tw_s + S;
    for (int i = tid; i < S; i += NT) tw_s[i] = reinterpret_cast<const cplx<T> *>(a.tw_s)[i];
    for (int i = tid; i < C; i += NT) tw_c[i] = reinterpret_cast<const cplx<T> *>(a.tw_c)[i];
    __syncthreads();
    typedef const cplx<T> *CP;
    cplx<T> *tmp = tw_c + C;                     // spare LDS behind them (levels with a radix > 32 only)
    // range axis: element s = s2 * n1 + n2 lives in row s: groups (column, n2) radix s1, then (column, k1) radix s2;
    // Doppler axis: element c = c2 * m1 + m2 lives in column c; lanes walk the rows (pitch Cp is odd).
    if constexpr (!BIG) {
        // four call sites: the strides that are 1 fold into the address arithmetic
        const RaderTab none{};
        dft_level_any<CLS, BIG, NT, T>(a.s1, none, lds, tmp, 0, tid, C, a.mg_C, 1, a.s2, Cp, a.s2 * Cp, (CP)a.m_s1,
                                       a.s2 > 1 ? (CP)tw_s : (CP) nullptr, S);
        if (a.s2 > 1)
            dft_level_any<CLS, BIG, NT, T>(a.s2, none, lds, tmp, 0, tid, C, a.mg_C, 1, a.s1, a.s2 * Cp, Cp, (CP)a.m_s2, (CP) nullptr, S);
        dft_level_any<CLS, BIG, NT, T>(a.c1, none, lds, tmp, 0, tid, S, a.mg_S, Cp, a.c2, 1, a.c2, (CP)a.m_c1,
                                       a.c2 > 1 ? (CP)tw_c : (CP) nullptr, C);
        if (a.c2 > 1)
            dft_level_any<CLS, BIG, NT, T>(a.c2, none, lds, tmp, 0, tid, S, a.mg_S, Cp, a.c1, a.c2, 1, (CP)a.m_c2, (CP) nullptr, C);
    } else {
    // kernels with a big level: one call site in a loop over the four levels, so the Rader / big-radix code and the
    // radix switch behind it are inlined once
    for (int lvl = 0; lvl < 4; ++lvl) {
        const bool rng = lvl < 2, lead = (lvl & 1) == 0;
        const int ra = rng ? a.s1 : a.c1, rb = rng ? a.s2 : a.c2;
        if (!lead && rb == 1) continue;
        const int R = lead ? ra : rb;
        const int n_inner = rng ? C : S, inner_stride = rng ? 1 : Cp, unit = rng ? Cp : 1;
        const int n_outer = lead ? rb : ra, o_stride = lead ? unit : rb * unit, e_stride = lead ? rb * unit : unit;
        const void *Wm = rng ? (lead ? a.m_s1 : a.m_s2) : (lead ? a.m_c1 : a.m_c2);
        const cplx<T> *tw = (lead && rb > 1) ? (rng ? tw_s : tw_c) : (const cplx<T> *)nullptr;
        // field-wise selects keep the plan in scalar registers (a run-time choice between the two structs would make
        // the compiler copy the whole kernel argument block to scratch)
        RaderTab rt;
        rt.P = lead ? (rng ? a.rad_s.P : a.rad_c.P) : 0;
        rt.r1 = rng ? a.rad_s.r1 : a.rad_c.r1;
        rt.r2 = rng ? a.rad_s.r2 : a.rad_c.r2;
        rt.idx = rng ? a.rad_s.idx : a.rad_c.idx;
        rt.B = rng ? a.rad_s.B : a.rad_c.B;
        rt.twL = rng ? a.rad_s.twL : a.rad_c.twL;
        rt.m_r1 = rng ? a.rad_s.m_r1 : a.rad_c.m_r1;
        rt.m_r2 = rng ? a.rad_s.m_r2 : a.rad_c.m_r2;
        dft_level_any<CLS, BIG, NT, T>(R, rt, lds, tmp, a.tmp_cells, tid, n_inner,
                                       rng ? a.mg_C : a.mg_S, inner_stride, n_outer, o_stride, e_stride, (CP)Wm, tw, rng ? S : C);
    }
    }
    // bin k = k1 + s1 k2 sits in row s2 k1 + k2 (same along the Doppler axis); fftshift: out[(d + C/2) % C] = X[d]
    const long obase = plane * cells;
    const int half = C / 2;
    for (int e = tid; e < cells; e += NT) {
        const int k = fast_div(e, a.mg_C, C), dd = e - k * C;
        int d = dd - half;
        if (d < 0) d += C;
        const int k2 = fast_div(k, a.mg_s1, a.s1), k1 = k - k2 * a.s1;
        const int d2 = fast_div(d, a.mg_c1, a.c1), d1 = d - d2 * a.c1;
        const cplx<T> v = lds[(a.s2 * k1 + k2) * Cp + a.c2 * d1 + d2];
        if constexpr (MAG)
            reinterpret_cast<T *>(a.out)[obase + e] = mag<T>(v);
        else
            __builtin_nontemporal_store(v, reinterpret_cast<cplx<T> *>(a.out) + obase + e);
    }
}

// ------------------------------------------------------------------ host side: factorisation and launch
// N = n1 * n2 (n1 >= n2) with the least n1 + n2 plus a per-level overhead; n2 == 1: one level.  Radices up to
// max_small run in registers, larger ones through dft_level_big when big_ok.
inline bool is_prime(int n) {
    if (n < 2) return false;
    for (int d = 2; (long)d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}

inline int mixed_axis(int N, int max_small, bool big_ok, int *n1, int *n2);

// Rader plan for a prime radix P > 32: P - 1 = r1 * r2 with both convolution radices in the register class.
inline bool rader_factors(int P, int max_small, int *r1, int *r2) {
    return P > 32 && is_prime(P) && mixed_axis(P - 1, max_small, false, r1, r2) >= 0;
}

inline int mixed_axis(int N, int max_small, bool big_ok, int *n1, int *n2) {
    int best = -1;
    for (int a = 1; a <= N; ++a) {
        if (N % a) continue;
        const int b = N / a;
        if (a < b || b > max_small) continue;
        if (a > max_small && (a <= 32 || !big_ok)) continue;
        int ca = a;
        if (a > 32) {                           // out of the LDS: Rader (~2 (r1 + r2) MACs + 6 LDS passes) or direct (~2 a)
            int r1, r2;
            ca = rader_factors(a, max_small, &r1, &r2) ? 2 * (r1 + r2) + 16 : 2 * a;
        }
        const int cost = (b == 1) ? ca + 3 : ca + b + 6;
        if (best < 0 || cost < best) {
            best = cost;
            *n1 = a;
            *n2 = b;
        }
    }
    return best;     // -1: no factorisation in this class
}

struct RdMixedPlan {
    int cls, big, s1, s2, c1, c2, Cp, tmp_cells;
    int rad_s[2], rad_c[2];      // Rader convolution radices of a prime s1 / c1 > 32 ({0, 0}: none)
    size_t lds_bytes;
};

constexpr size_t MIXED_LDS_MAX = 160 * 1024;

// Cheapest of the four kernel variants (radices <= 16 or <= 32 in registers, with or without a big level).
inline bool rd_mixed_plan(int S, int C, size_t elem_bytes, RdMixedPlan *out) {
    if (S < 1 || C < 1 || (long)S * C > (1 << 20)) return false;
    const int Cp = C | 1;
    const size_t plane = ((size_t)S * Cp + S + C) * elem_bytes;      // plane + the two inter-level twiddle tables
    if (plane > MIXED_LDS_MAX) return false;
    const int spare = (int)((MIXED_LDS_MAX - plane) / elem_bytes);
    int best = -1;
    for (int cls = 0; cls < 2; ++cls)
        for (int big = 0; big < 2; ++big) {
            RdMixedPlan pl{};
            const int max_small = cls == 0 ? 16 : 32;
            const int cs = mixed_axis(S, max_small, big, &pl.s1, &pl.s2), cc = mixed_axis(C, max_small, big, &pl.c1, &pl.c2);
            if (cs < 0 || cc < 0) continue;
            const int r_big = std::max(pl.s1 > 32 ? pl.s1 : 0, pl.c1 > 32 ? pl.c1 : 0);
            if (big && !r_big) continue;                        // same plan as the variant without the big level
            if (r_big && spare < 32 * (r_big + 1)) continue;    // a big level wants >= half a wave of groups in the spare LDS
            const int cost = (cs + cc) * 8 + cls;               // ties: the leaner register class
            if (best >= 0 && cost >= best) continue;
            best = cost;
            pl.cls = cls;
            pl.big = r_big ? 1 : 0;
            pl.Cp = Cp;
            if (!rader_factors(pl.s1, max_small, &pl.rad_s[0], &pl.rad_s[1])) pl.rad_s[0] = pl.rad_s[1] = 0;
            if (!rader_factors(pl.c1, max_small, &pl.rad_c[0], &pl.rad_c[1])) pl.rad_c[0] = pl.rad_c[1] = 0;
            // spare LDS for the big levels: all their groups if they fit, at most two waves of groups
            int g_big = 0;
            if (pl.s1 > 32) g_big = std::max(g_big, C * pl.s2);
            if (pl.c1 > 32) g_big = std::max(g_big, S * pl.c2);
            pl.tmp_cells = r_big ? std::min(spare, std::min(128, g_big) * (r_big + 1)) : 0;
            pl.lds_bytes = plane + (size_t)pl.tmp_cells * elem_bytes;
            *out = pl;
        }
    return best >= 0;
}

inline bool rd_mixed_supported(int S, int C) {
    RdMixedPlan pl;
    return rd_mixed_plan(S, C, sizeof(cplx<float>), &pl);
}

// Device tables of a Rader plan (cached in the context): index maps for the smallest primitive root g of P and
// DFT_L(b) / L, b[m] = W_P^(g^-m), in the slot order of the two-level DFT (long double on the host, rounded once).
template <typename T>
int get_rader_tables(mmw_ctx *ctx, int P, int r1, int r2, RaderTab *rt) {
    const int L = P - 1;
    rt->P = P;
    rt->r1 = r1;
    rt->r2 = r2;
    MMW_TRY(get_table<T>(ctx, TAB_TWIDDLE, L, &rt->twL));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, r1, &rt->m_r1));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, r2, &rt->m_r2));
    const auto key_idx = std::make_tuple(100, P, 0);
    const auto key_b = std::make_tuple(101, P * 4096 + r1 * 64 + r2, (int)(sizeof(T) == 8));
    auto it_i = ctx->tables.find(key_idx), it_b = ctx->tables.find(key_b);
    if (it_i != ctx->tables.end() && it_b != ctx->tables.end()) {
        rt->idx = (const int *)it_i->second;
        rt->B = it_b->second;
        return MMW_OK;
    }
    auto powmod = [&](long b, long e) {
        long r = 1;
        for (b %= P; e > 0; e >>= 1, b = b * b % P)
            if (e & 1) r = r * b % P;
        return r;
    };
    int g = 2;
    for (;; ++g) {                                      // smallest primitive root
        bool ok = true;
        int rest = L;
        for (int q = 2; q <= rest && ok; ++q)
            if (rest % q == 0) {
                if (powmod(g, L / q) == 1) ok = false;
                while (rest % q == 0) rest /= q;
            }
        if (ok) break;
    }
    std::vector<int> perm(L), idx(2 * P, 0);
    for (int m = 0; m < L; ++m) perm[m] = (int)powmod(g, m);
    for (int m = 0; m < L; ++m) idx[perm[m]] = m;
    for (int n = 1; n < P; ++n) idx[P + n] = (L - idx[n]) % L;         // g^-q = n  <=>  q = (L - m) mod L
    std::vector<T> hb(2 * (size_t)L);
    for (int k1 = 0; k1 < r1; ++k1)
        for (int k2 = 0; k2 < r2; ++k2) {
            const int k = k1 + r1 * k2;
            long double re = 0, im = 0;
            for (int m = 0; m < L; ++m) {
                const long e1 = perm[(L - m) % L];                       // b[m] = W_P^(g^-m)
                const long double ang = -2.0L * M_PIl * ((long double)e1 / P + (long double)(((long)m * k) % L) / L);
                re += cosl(ang);
                im += sinl(ang);
            }
            hb[2 * (size_t)(r2 * k1 + k2)] = (T)(re / L);
            hb[2 * (size_t)(r2 * k1 + k2) + 1] = (T)(im / L);
        }
    void *d_idx = nullptr, *d_b = nullptr;
    if (it_i == ctx->tables.end()) {
        if (hipMalloc(&d_idx, idx.size() * sizeof(int)) != hipSuccess) return set_error(MMW_ERR_NOMEM, "hipMalloc for Rader index table failed");
        MMW_HIP(hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        ctx->tables[key_idx] = d_idx;
    } else
        d_idx = it_i->second;
    if (hipMalloc(&d_b, hb.size() * sizeof(T)) != hipSuccess) return set_error(MMW_ERR_NOMEM, "hipMalloc for Rader spectrum table failed");
    MMW_HIP(hipMemcpyAsync(d_b, hb.data(), hb.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));         // host vectors go out of scope
    ctx->tables[key_b] = d_b;
    rt->idx = (const int *)d_idx;
    rt->B = d_b;
    return MMW_OK;
}

// compile-time specialised kernels for the shipped cfg shapes (mmw_fft_mixed_ct.h, own translation units)
bool rd_split_ct_supported(int S, int C);       // mmw_fft_split_ct.h: planes of 2 x 16384 cells, single pass
int launch_rd_split_ct(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, int S, int C, RawView rv);
bool rd_mixed_ct_supported(int S, int C);
bool rd_mixed_ct_raw_sync_supported(int S, int C);
int launch_rd_mixed_ct(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv,
                       const ChainSync *cs = nullptr, int sync_cus = 0, int *sync_grid = nullptr, bool query_only = false,
                       float *d_l1 = nullptr);

// planes x [S][C] complex64 at d_in (plane pitch in_plane_stride elements) -> d_out planes, contiguous:
// T = float: complex64 spectrum; T = double, MAG: float64 magnitude (the CFAR plane).
template <typename T, bool MAG>
int launch_rd_mixed(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C,
                    RawView rv = RawView{1, 0}) {
    if constexpr (sizeof(T) == 4 && !MAG)
        if (rd_mixed_ct_supported(S, C))
            return launch_rd_mixed_ct(ctx, d_in, in_plane_stride, d_out, planes, S, C, rv);
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
    a.tmp_cells = pl.tmp_cells;
    if (pl.rad_s[0]) MMW_TRY(get_rader_tables<T>(ctx, pl.s1, pl.rad_s[0], pl.rad_s[1], &a.rad_s));
    if (pl.rad_c[0]) MMW_TRY(get_rader_tables<T>(ctx, pl.c1, pl.rad_c[0], pl.rad_c[1], &a.rad_c));
    a.raw = rv;
    a.planes = planes;
    a.mg_C = div_magic(C);
    a.mg_S = div_magic(S);
    a.mg_s1 = div_magic(pl.s1);
    a.mg_c1 = div_magic(pl.c1);
    MMW_TRY(get_table<T>(ctx, TAB_HANN, S, &a.win_s));
    MMW_TRY(get_table<T>(ctx, TAB_HANN, C, &a.win_c));
    MMW_TRY(get_table<T>(ctx, TAB_TWIDDLE, S, &a.tw_s));
    MMW_TRY(get_table<T>(ctx, TAB_TWIDDLE, C, &a.tw_c));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.s1, &a.m_s1));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.s2, &a.m_s2));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.c1, &a.m_c1));
    MMW_TRY(get_table<T>(ctx, TAB_DFTMAT, pl.c2, &a.m_c2));
    auto go = [&](auto kern, int nt) -> int {
        if (pl.lds_bytes > 64 * 1024)
            MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)pl.lds_bytes));
        const unsigned grid = rv.ntx > 1 ? (unsigned)raw_grid(planes, rv) : (unsigned)skip_planes(planes, rv);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), pl.lds_bytes, ctx->stream, a);
        return check_launch("rd_mixed");
    };
    // Workgroup size: the register budget allows 24 waves per CU in class 0 (78 VGPRs) and 12 in class 1; the LDS
    // footprint decides how many planes share a CU, so the threads per plane are chosen to fill those waves
    // (measured on 63x100: 256 threads 0.68, 512 threads 0.53 us/frame; on 120x126: 256 -> 1024 threads 4.1 -> 1.7).
    const int wgs = (int)(MIXED_LDS_MAX / pl.lds_bytes);
    if constexpr (sizeof(T) == 4) {
        const bool wide = pl.big || wgs < 3;
        switch (pl.cls * 2 + pl.big) {
            case 0:
                if (wide) return go(k_rd_mixed<T, 0, false, MAG, 1024>, 1024);
                if (wgs < 6) return go(k_rd_mixed<T, 0, false, MAG, 512>, 512);
                return go(k_rd_mixed<T, 0, false, MAG, 256>, 256);
            case 1: return go(k_rd_mixed<T, 0, true, MAG, 1024>, 1024);
            case 2: return wide ? go(k_rd_mixed<T, 1, false, MAG, 512>, 512) : go(k_rd_mixed<T, 1, false, MAG, 256>, 256);
            default: return go(k_rd_mixed<T, 1, true, MAG, 512>, 512);
        }
    } else {
        switch (pl.cls * 2 + pl.big) {
            case 0: return go(k_rd_mixed<T, 0, false, MAG, 512>, 512);
            case 1: return go(k_rd_mixed<T, 0, true, MAG, 512>, 512);
            case 2: return go(k_rd_mixed<T, 1, false, MAG, 256>, 256);
            default: return go(k_rd_mixed<T, 1, true, MAG, 256>, 256);
        }
    }
}

}  // namespace mmw
