// Compile-time specialised mixed-radix range-Doppler kernels for the plane shapes of the cfg files the reference ships.
//
// k_rd_mixed (mmw_fft_mixed.h) takes any (S, C) at run time: radices through a switch, strides and trip counts in
// registers, one integer multiply-add per LDS element address, (o * k) mod N twiddle indices kept by compare-and-subtract.
// rocprof (--pmc SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU, 12 x 63 x 100) showed the kernel VALU-bound at ~160 vector
// instructions per cell of which only ~70 were transform arithmetic.  Here S, C and the factorisation S = S1 S2,
// C = C1 C2 are template parameters: every LDS address is a base register plus an immediate offset, divisions are by
// constants, loops unroll to their exact trip counts, the inter-level twiddles W_N^(n2 k1) come from a 2-D table
// tw2[n2][k1] in LDS at immediate offsets, and the R-point transforms are RegDFT<R> (mmw_dft_small.h).
// Same data flow as k_rd_mixed: one workgroup owns one [S][C] plane in LDS (row pitch C | 1),
//   load (Hann(S) x Hann(C)) -> range level A (radix S1) -> B (S2) -> Doppler level A (C1) -> B (C2) -> store with the
//   Doppler fftshift and the digit reversal folded into the index          (processors/range_doppler_resp.py:94-103).
#pragma once
#include "mmw_fft_mixed.h"

namespace mmw {

namespace mixct {
// vector-instruction counts of RegDFT<R> on gfx950 (hipcc -O3, measured from the ISA), R = 0..32
constexpr int DFT_VALU[33] = {0,   0,   1,   11,  9,   26,  28,  45,  36,  76,  62,  95,  74,  126, 104, 133, 110,
                              200, 170, 243, 154, 212, 212, 341, 199, 296, 278, 361, 250, 518, 296, 585, 298};
constexpr int MAX_RADIX = 20;       // registers: RegDFT<20> holds ~70 VGPRs
// N = N1 * N2 (N1 >= N2): cheapest pair by instructions per point; a second level costs a twiddle product and one more
// trip through the LDS (~8 instructions per point)
constexpr int split_cost(int n1, int n2) {
    return (DFT_VALU[n1] * 100) / n1 + (n2 > 1 ? (DFT_VALU[n2] * 100) / n2 + 800 : 0);
}
constexpr int BIG_PRIME = 127;      // handled by dft_level_bigprime_ct (63 x 127, 127 x 32, 254 x 50 cfgs)
constexpr int best_n1(int N) {
    if (N % BIG_PRIME == 0 && N / BIG_PRIME <= MAX_RADIX) return BIG_PRIME;
    int best = 0, best_cost = 1 << 30;
    for (int a = 1; a <= N && a <= MAX_RADIX; ++a) {
        if (N % a) continue;
        const int b = N / a;
        if (b > a || b > MAX_RADIX) continue;
        const int c = split_cost(a, b);
        if (c < best_cost) {
            best_cost = c;
            best = a;
        }
    }
    return best;        // 0: no split within the radix limit
}
constexpr bool supported(int S, int C) { return best_n1(S) > 0 && best_n1(C) > 0; }
constexpr int threads_for(int S, int C) {
    // as launch_rd_mixed: fill ~24 waves per CU given how many planes share its LDS
    const long lds = ((long)S * (C | 1) + S + C) * 8;
    const int wgs = (int)(160 * 1024 / lds);
    return wgs < 3 ? 1024 : (wgs < 6 ? 512 : 256);
}
}  // namespace mixct

struct RdMixedCtArgs {
    const void *in;             // complex64 planes
    void *out;                  // complex64 planes
    long in_plane_stride;       // complex elements between consecutive input planes
    const float *win_s, *win_c;
    const cplx<float> *tw2_s, *tw2_c;       // [S2][S1] and [C2][C1]: W_S^(n2 k1), W_C^(m2 k1)
    const float *cs_big;        // coefficient table of the big-prime level (nullptr when no axis has one)
    RawView raw;
    long planes;
};

// One level over the LDS plane: N_GROUPS groups g = o * N_INNER + i (i fastest over the lanes); element j of a group sits
// at base + i * INNER_STRIDE + o * OUTER_STRIDE + j * ESTRIDE.  tw2 (when TW): output k of outer index o times tw2[o * R + k].
template <int R, int NT, int N_INNER, int INNER_STRIDE, int N_OUTER, int OUTER_STRIDE, int ESTRIDE, bool TW>
__device__ __forceinline__ void dft_level_ct(cplx<float> *lds, const cplx<float> *tw2, int tid) {
    constexpr int N_GROUPS = N_INNER * N_OUTER, ROUNDS = (N_GROUPS + NT - 1) / NT;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int g = tid + r * NT;
        if (N_GROUPS % NT == 0 || r + 1 < ROUNDS || g < N_GROUPS) {
            const int o = g / N_INNER, i = g - o * N_INNER;
            cplx<float> *p = lds + i * INNER_STRIDE + o * OUTER_STRIDE;
            cplx<float> x[R];
#pragma unroll
            for (int j = 0; j < R; ++j) x[j] = p[j * ESTRIDE];
            RegDFT<R, float>::run(x);
            if constexpr (TW) {
                const cplx<float> *t = tw2 + o * R;
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], t[k]);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) p[k * ESTRIDE] = x[k];
        }
    }
}

// A level whose radix is a prime P too large for registers (the 127 of three shipped cfgs), same contract as dft_level_ct.
// Real-symmetric direct form, in place without scratch:
//   1. every group's x_j, x_{P-j} (j = 1 .. H = (P-1)/2) become s_j = x_j + x_{P-j} (stored at j) and d_j = x_j - x_{P-j}
//      (stored at P-j);
//   2. a work item = (block of 4 output indices k, group): t_k = x_0 + sum_j cos(2 pi jk/P) s_j, u_k = sum_j sin(..) d_j kept
//      in registers -- lanes of a wave share the block, so the 8 coefficients of every j arrive by ONE scalar load and
//      the inner loop is 8 packed FMAs per two LDS reads (H complex-by-real MACs per output, against 2 (r1 + r2) + 3
//      complex ones plus six LDS passes for the Rader form this replaces, whose few groups per pass left most lanes idle);
//   3. after a barrier X_k = t_k - j u_k and X_{P-k} = t_k + j u_k overwrite the group (times the inter-level twiddle).
// cs: [H][(H + 1) / 4 blocks][8] floats = cos(2 pi j k / P) for the block's four k, then the four sines; k = 0 .. H.
// A work item takes 4 or 8 consecutive k.
template <int P, int NT, int N_INNER, int INNER_STRIDE, int N_OUTER, int OUTER_STRIDE, int ESTRIDE, bool TW>
__device__ __forceinline__ void dft_level_bigprime_ct(cplx<float> *lds, const cplx<float> *tw2, const float *__restrict__ cs, int tid) {
    constexpr int H = (P - 1) / 2, N_GROUPS = N_INNER * N_OUTER, G64 = (N_GROUPS + 63) / 64 * 64;
    // outputs per work item: 8 when that still gives every thread an item (half the LDS reads per output), else 4
    constexpr int KB = ((H + 1) / 8) * G64 >= NT ? 8 : 4, NBLK = (H + 1) / KB;
    static_assert((H + 1) % 8 == 0, "output blocks of four or eight");
    constexpr int ITEMS = NBLK * G64, ROUNDS = (ITEMS + NT - 1) / NT;
    // 1. s / d in place
    for (int e = tid; e < N_GROUPS * H; e += NT) {
        const int j = e / N_GROUPS + 1, g = e - (j - 1) * N_GROUPS;
        const int o = g / N_INNER, i = g - o * N_INNER;
        cplx<float> *p = lds + i * INNER_STRIDE + o * OUTER_STRIDE;
        const cplx<float> a = p[j * ESTRIDE], b = p[(P - j) * ESTRIDE];
        p[j * ESTRIDE] = a + b;
        p[(P - j) * ESTRIDE] = a - b;
    }
    __syncthreads();
    // 2. outputs in registers
    cplx<float> t[ROUNDS][KB], u[ROUNDS][KB];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int w = tid + r * NT;
        const int blk = __builtin_amdgcn_readfirstlane(w / G64), g = w - (w / G64) * G64;    // the whole wave shares blk
        const bool live = w < ITEMS && g < N_GROUPS;
        const int gg = live ? g : 0, o = gg / N_INNER, i = gg - o * N_INNER;
        const cplx<float> *p = lds + i * INNER_STRIDE + o * OUTER_STRIDE;
        const cplx<float> x0 = p[0];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            t[r][q] = x0;
            u[r][q] = cplx<float>{0.f, 0.f};
        }
        // table rows are [j][block of 4][cos x4, sin x4]; a block of 8 outputs uses two consecutive 4-blocks
        const float *row = cs + (size_t)(blk < NBLK ? blk : 0) * (2 * KB);
#pragma unroll 3
        for (int j = 1; j <= H; ++j) {
            const cplx<float> sj = p[j * ESTRIDE], dj = p[(P - j) * ESTRIDE];
            const float *c = row + (size_t)(j - 1) * ((H + 1) / 4) * 8;     // wave-uniform address: scalar loads
#pragma unroll
            for (int q = 0; q < KB; ++q) {
                t[r][q] = t[r][q] + sj * c[(q >> 2) * 8 + (q & 3)];
                u[r][q] = u[r][q] + dj * c[(q >> 2) * 8 + 4 + (q & 3)];
            }
        }
    }
    __syncthreads();
    // 3. write X_k, X_{P-k}
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int w = tid + r * NT;
        const int blk = w / G64, g = w - blk * G64;
        if (w < ITEMS && g < N_GROUPS) {
            const int o = g / N_INNER, i = g - o * N_INNER;
            cplx<float> *p = lds + i * INNER_STRIDE + o * OUTER_STRIDE;
#pragma unroll
            for (int q = 0; q < KB; ++q) {
                const int k = blk * KB + q;
                cplx<float> lo = cplx<float>{t[r][q].x + u[r][q].y, t[r][q].y - u[r][q].x};      // t - j u
                cplx<float> hi = cplx<float>{t[r][q].x - u[r][q].y, t[r][q].y + u[r][q].x};      // t + j u
                if constexpr (TW) {
                    lo = cmul(lo, tw2[o * P + k]);
                    if (k > 0) hi = cmul(hi, tw2[o * P + P - k]);
                }
                p[k * ESTRIDE] = lo;
                if (k > 0) p[(P - k) * ESTRIDE] = hi;
            }
        }
    }
}

// level dispatch: register-resident radix or the big prime
template <int R, int NT, int N_INNER, int INNER_STRIDE, int N_OUTER, int OUTER_STRIDE, int ESTRIDE, bool TW>
__device__ __forceinline__ void dft_level_any_ct(cplx<float> *lds, const cplx<float> *tw2, const float *cs, int tid) {
    if constexpr (R == mixct::BIG_PRIME)
        dft_level_bigprime_ct<R, NT, N_INNER, INNER_STRIDE, N_OUTER, OUTER_STRIDE, ESTRIDE, TW>(lds, tw2, cs, tid);
    else
        dft_level_ct<R, NT, N_INNER, INNER_STRIDE, N_OUTER, OUTER_STRIDE, ESTRIDE, TW>(lds, tw2, tid);
}

template <int S, int C, int NT>
__global__ __launch_bounds__(NT) void k_rd_mixed_ct(RdMixedCtArgs a) {
    constexpr int S1 = mixct::best_n1(S), S2 = S / S1, C1 = mixct::best_n1(C), C2 = C / C1;
    constexpr int Cp = C | 1, CELLS = S * C;
    static_assert(S1 > 0 && C1 > 0, "no factorisation within the register-resident radices");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(smem);          // [S][Cp]
    cplx<float> *tw_s = lds + S * Cp, *tw_c = tw_s + S;
    const int tid = threadIdx.x;
    long plane = blockIdx.x;
    const bool raw = a.raw.ntx > 1;
    if (raw) {
        plane = raw_block_plane(blockIdx.x, a.planes, a.raw);
        if (plane < 0 || skip_raw_plane(plane, a.raw)) return;
    } else plane = skip_block_plane(blockIdx.x, a.raw);
    const int ntx = raw ? a.raw.ntx : 1;
    const cplx<float> *in = raw ? raw_plane(reinterpret_cast<const cplx<float> *>(a.in), plane, S, C, a.raw)
                                : reinterpret_cast<const cplx<float> *>(a.in) + plane * a.in_plane_stride;
    // ---- load + windows
    if ((C % 2 == 0) && !raw) {                 // two adjacent chirps per 16-B load
        const f32x4 *in4 = reinterpret_cast<const f32x4 *>(in);
        constexpr int PAIRS = CELLS / 2, ROUNDS = (PAIRS + NT - 1) / NT;
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) {
            const int e = tid + q * NT;
            if (PAIRS % NT == 0 || q + 1 < ROUNDS || e < PAIRS) {
                const int s = (2 * e) / C, c = (2 * e) - s * C;
                const f32x4 v = __builtin_nontemporal_load(in4 + e);
                const float ws = a.win_s[s], w0 = ws * a.win_c[c], w1 = ws * a.win_c[c + 1];
                lds[s * Cp + c] = cplx<float>{v.x * w0, v.y * w0};
                lds[s * Cp + c + 1] = cplx<float>{v.z * w1, v.w * w1};
            }
        }
    } else {
        constexpr int ROUNDS = (CELLS + NT - 1) / NT;
#pragma unroll 4
        for (int q = 0; q < ROUNDS; ++q) {
            const int e = tid + q * NT;
            if (CELLS % NT == 0 || q + 1 < ROUNDS || e < CELLS) {
                const int s = e / C, c = e - s * C;
                const cplx<float> v = __builtin_nontemporal_load(in + (long)e * ntx);
                lds[s * Cp + c] = v * (a.win_s[s] * a.win_c[c]);
            }
        }
    }
    if constexpr (S2 > 1)
        for (int i = tid; i < S; i += NT) tw_s[i] = a.tw2_s[i];
    if constexpr (C2 > 1)
        for (int i = tid; i < C; i += NT) tw_c[i] = a.tw2_c[i];
    __syncthreads();
    // ---- range axis: sample s = S2 n1 + n2 lives in row s.  A: groups (column, n2), radix S1; B: groups (column, k1), radix S2
    dft_level_any_ct<S1, NT, C, 1, S2, Cp, S2 * Cp, (S2 > 1)>(lds, tw_s, a.cs_big, tid);
    __syncthreads();
    if constexpr (S2 > 1) {
        dft_level_ct<S2, NT, C, 1, S1, S2 * Cp, Cp, false>(lds, nullptr, tid);
        __syncthreads();
    }
    // ---- Doppler axis: chirp c = C2 m1 + m2 lives in column c; lanes walk the rows (odd pitch: conflict free)
    dft_level_any_ct<C1, NT, S, Cp, C2, 1, C2, (C2 > 1)>(lds, tw_c, a.cs_big, tid);
    __syncthreads();
    if constexpr (C2 > 1) {
        dft_level_ct<C2, NT, S, Cp, C1, C2, 1, false>(lds, nullptr, tid);
        __syncthreads();
    }
    // ---- store: range bin k = k1 + S1 k2 sits in row S2 k1 + k2, Doppler bin d = d1 + C1 d2 in column C2 d1 + d2;
    //      fftshift: out[(d + C/2) % C] = X[d]
    cplx<float> *out = reinterpret_cast<cplx<float> *>(a.out) + plane * CELLS;
    constexpr int HALF = C / 2, ROUNDS = (CELLS + NT - 1) / NT;
#pragma unroll 4
    for (int q = 0; q < ROUNDS; ++q) {
        const int e = tid + q * NT;
        if (CELLS % NT == 0 || q + 1 < ROUNDS || e < CELLS) {
            const int k = e / C, dd = e - k * C;
            int d = dd - HALF;
            if (d < 0) d += C;
            const int k2 = k / S1, k1 = k - k2 * S1, d2 = d / C1, d1 = d - d2 * C1;
            __builtin_nontemporal_store(lds[(S2 * k1 + k2) * Cp + C2 * d1 + d2], out + e);
        }
    }
}

// the non-power-of-two planes of the shipped cfgs (tests/golden/cfg_scalars.json) that need no prime radix above 20
// (63 x 115 keeps the run-time kernel: 115 = 23 * 5)
#define MMW_MIXED_CT_SHAPES_A(X) X(63, 70) X(63, 100) X(64, 40) X(70, 40) X(90, 80) X(100, 30) X(254, 50) X(127, 32)
#define MMW_MIXED_CT_SHAPES_B(X) X(90, 100) X(100, 100) X(120, 126) X(130, 50) X(200, 40) X(63, 127)

// [N2][N1] table W_N^(n2 k1), cached per context
inline int get_tw2_table(mmw_ctx *ctx, int N, int N1, const void **out) {
    const auto key = std::make_tuple(200, N * 256 + N1, 0);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return MMW_OK;
    }
    const int N2 = N / N1;
    std::vector<float> h(2 * (size_t)N);
    for (int n2 = 0; n2 < N2; ++n2)
        for (int k1 = 0; k1 < N1; ++k1) {
            const long double ang = -2.0L * M_PIl * (long double)(((long)n2 * k1) % N) / (long double)N;
            h[2 * ((size_t)n2 * N1 + k1)] = (float)cosl(ang);
            h[2 * ((size_t)n2 * N1 + k1) + 1] = (float)sinl(ang);
        }
    void *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return set_error(MMW_ERR_NOMEM, "hipMalloc for twiddle table failed");
    MMW_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tables[key] = d;
    *out = d;
    return MMW_OK;
}

// [H][NBLK][8] coefficient table of dft_level_bigprime_ct, cached per context
inline int get_bigprime_table(mmw_ctx *ctx, int P, const void **out) {
    const auto key = std::make_tuple(201, P, 0);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return MMW_OK;
    }
    const int H = (P - 1) / 2, NBLK = (H + 1) / 4;
    std::vector<float> h((size_t)H * NBLK * 8);
    for (int j = 1; j <= H; ++j)
        for (int b = 0; b < NBLK; ++b)
            for (int q = 0; q < 4; ++q) {
                const long double ang = 2.0L * M_PIl * (long double)(((long)j * (4 * b + q)) % P) / (long double)P;
                h[((size_t)(j - 1) * NBLK + b) * 8 + q] = (float)cosl(ang);
                h[((size_t)(j - 1) * NBLK + b) * 8 + 4 + q] = (float)sinl(ang);
            }
    void *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return set_error(MMW_ERR_NOMEM, "hipMalloc for DFT coefficient table failed");
    MMW_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tables[key] = d;
    *out = d;
    return MMW_OK;
}

template <int S, int C>
int launch_rd_mixed_ct_sc(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, RawView rv) {
    constexpr int NT = mixct::threads_for(S, C), S1 = mixct::best_n1(S), C1 = mixct::best_n1(C);
    RdMixedCtArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.in_plane_stride = in_plane_stride;
    a.raw = rv;
    a.planes = planes;
    const void *p;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &p));
    a.win_s = (const float *)p;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, C, &p));
    a.win_c = (const float *)p;
    MMW_TRY(get_tw2_table(ctx, S, S1, &p));
    a.tw2_s = (const cplx<float> *)p;
    MMW_TRY(get_tw2_table(ctx, C, C1, &p));
    a.tw2_c = (const cplx<float> *)p;
    if constexpr (S1 == mixct::BIG_PRIME || C1 == mixct::BIG_PRIME) {
        MMW_TRY(get_bigprime_table(ctx, mixct::BIG_PRIME, &p));
        a.cs_big = (const float *)p;
    }
    constexpr size_t lds_bytes = ((size_t)S * (C | 1) + S + C) * sizeof(cplx<float>);
    auto kern = k_rd_mixed_ct<S, C, NT>;
    if (lds_bytes > 64 * 1024)
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const unsigned grid = rv.ntx > 1 ? (unsigned)raw_grid(planes, rv) : (unsigned)skip_planes(planes, rv);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_bytes, ctx->stream, a);
    return check_launch("rd_mixed_ct");
}

// one translation unit per half of the shape list (mmw_tu_mixed_ct_a/b.hip), so the library still builds in parallel
#if defined(MMW_TU_MIXED_CT_A) || defined(MMW_TU_MIXED_CT_B)
#ifdef MMW_TU_MIXED_CT_A
#define MMW_MIXED_CT_LIST MMW_MIXED_CT_SHAPES_A
#define MMW_MIXED_CT_FN launch_rd_mixed_ct_a
#else
#define MMW_MIXED_CT_LIST MMW_MIXED_CT_SHAPES_B
#define MMW_MIXED_CT_FN launch_rd_mixed_ct_b
#endif
int MMW_MIXED_CT_FN(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv) {
#define X(s, c) if (S == s && C == c) return launch_rd_mixed_ct_sc<s, c>(ctx, d_in, in_plane_stride, d_out, planes, rv);
    MMW_MIXED_CT_LIST(X)
#undef X
    return MMW_ERR_UNSUPPORTED;
}
#endif

#ifdef MMW_TU_MIXED_CT_A
int launch_rd_mixed_ct_b(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv);
bool rd_mixed_ct_supported(int S, int C) {
#define X(s, c) if (S == s && C == c) return true;
    MMW_MIXED_CT_SHAPES_A(X)
    MMW_MIXED_CT_SHAPES_B(X)
#undef X
    return false;
}
int launch_rd_mixed_ct(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv) {
    int rc = launch_rd_mixed_ct_a(ctx, d_in, in_plane_stride, d_out, planes, S, C, rv);
    if (rc == MMW_ERR_UNSUPPORTED) rc = launch_rd_mixed_ct_b(ctx, d_in, in_plane_stride, d_out, planes, S, C, rv);
    if (rc == MMW_ERR_UNSUPPORTED) return set_error(MMW_ERR_UNSUPPORTED, "no compile-time mixed-radix kernel for %dx%d", S, C);
    return rc;
}
#endif

}  // namespace mmw
