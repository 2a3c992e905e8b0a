// Single-pass range-Doppler for planes of up to 2 x 16384 cells that do NOT fit the LDS whole (512 x 64, 128 x 256,
// 1024 x 32: 256 KB against 160 KB): the outermost radix-2 step of the Doppler FFT runs through REGISTERS.
//
//   X[d], X[d + C/2] = E[d] +/- W_C^d O[d],   E / O = DFT_{C/2} of the even / odd chirps (decimation in time).
//
// One workgroup per plane.  It loads the whole plane once (16-B loads: chirps 2i, 2i + 1 of a sample row), puts the windowed
// EVEN chirps into the LDS half-plane [S][C/2] and keeps the windowed ODD chirps in registers; runs the levels of
// k_rd_mixed_ct (range S1 x S2, Doppler C1 x C2 of the half length) to get E, moves E into registers (the odd inputs
// go to the LDS at the same moment), runs the same levels again for O, and the final pass combines E (registers) with O (LDS)
// and stores both halves of every output row with the Doppler fftshift folded in.  HBM traffic = the algorithmic bytes;
// the two-kernel path these planes took before moved them twice (1.6-1.9 TB/s).
// 512 threads: 32 cells per thread and half = 64 VGPRs held across a half, beside the ~70 of a 32-point register FFT.
// (processors/range_doppler_resp.py:94-103)
#pragma once
#include "mmw_fft_mixed_ct.h"

namespace mmw {

struct RdSplitArgs {
    const void *in;
    void *out;
    const float *win_s, *win_c;             // Hann(S), Hann(C)
    const cplx<float> *tw2_s, *tw2_c;       // [S2][S1], [CH2][CH1] inter-level twiddles
    const cplx<float> *tw_c;                // W_C^d, d < C (the radix-2 combine uses d < C/2)
    RawView raw;                            // vskip only (virtual-array cubes)
    long planes;
};

template <int S, int C> constexpr size_t split_lds_bytes() {
    constexpr int CH = C / 2;
    return ((size_t)S * (CH | 1) + S + CH + CH) * sizeof(cplx<float>) + (size_t)(S + C) * sizeof(float);
}

template <int S, int C, int NT>
__global__ __launch_bounds__(NT) void k_rd_split2_ct(RdSplitArgs a) {
    constexpr int CH = C / 2, CHp = CH | 1;
    constexpr int S1 = mixct::best_n1(S), S2 = S / S1, C1 = mixct::best_n1(CH), C2 = CH / C1;
    static_assert(C % 2 == 0 && S1 > 0 && C1 > 0 && S1 != mixct::BIG_PRIME && C1 != mixct::BIG_PRIME, "split kernel shape");
    constexpr int HALF_CELLS = S * CH, ROUNDS = HALF_CELLS / NT;
    static_assert(HALF_CELLS % NT == 0, "whole rounds");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(smem);          // [S][CHp]
    cplx<float> *tw_s = lds + S * CHp, *tw_c = tw_s + S, *tw_comb = tw_c + CH;
    float *win_s = reinterpret_cast<float *>(tw_comb + CH), *win_c = win_s + S;
    const int tid = threadIdx.x;
    const long plane = skip_block_plane(blockIdx.x, a.raw);
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(a.in) + plane * (long)HALF_CELLS;
    // the whole plane: one 16-B load per (sample row, chirp pair)
    f32x4 raw[ROUNDS];
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) raw[q] = __builtin_nontemporal_load(in4 + tid + q * NT);
    if constexpr (S2 > 1)
        for (int i = tid; i < S; i += NT) tw_s[i] = a.tw2_s[i];
    if constexpr (C2 > 1)
        for (int i = tid; i < CH; i += NT) tw_c[i] = a.tw2_c[i];
    for (int i = tid; i < CH; i += NT) tw_comb[i] = a.tw_c[i];
    for (int i = tid; i < S; i += NT) win_s[i] = a.win_s[i];
    for (int i = tid; i < C; i += NT) win_c[i] = a.win_c[i];
    __syncthreads();
    cplx<float> held[ROUNDS];           // half 0: the windowed odd chirps; half 1: E
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
        const int e = tid + q * NT, s = e / CH, i = e - s * CH;
        const float ws = win_s[s], w0 = ws * win_c[2 * i], w1 = ws * win_c[2 * i + 1];
        lds[s * CHp + i] = cplx<float>{raw[q].x * w0, raw[q].y * w0};
        held[q] = cplx<float>{raw[q].z * w1, raw[q].w * w1};
    }
    cplx<float> *out = reinterpret_cast<cplx<float> *>(a.out) + plane * (long)(S * C);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // an opaque copy of the thread index per half keeps the two copies of the level code from sharing hoisted
        // address registers (the same device as in k_rd_mixed_ct's plane loop)
        int t = tid;
        asm volatile("" : "+v"(t));
        __syncthreads();
        dft_level_ct<S1, NT, CH, 1, S2, CHp, S2 * CHp, (S2 > 1)>(lds, tw_s, t);
        __syncthreads();
        if constexpr (S2 > 1) {
            dft_level_ct<S2, NT, CH, 1, S1, S2 * CHp, CHp, false>(lds, nullptr, t);
            __syncthreads();
        }
        dft_level_ct<C1, NT, S, CHp, C2, 1, C2, (C2 > 1)>(lds, tw_c, t);
        __syncthreads();
        if constexpr (C2 > 1) {
            dft_level_ct<C2, NT, S, CHp, C1, C2, 1, false>(lds, nullptr, t);
            __syncthreads();
        }
        // range bin k = k1 + S1 k2 sits in row S2 k1 + k2, half-length Doppler bin d = d1 + C1 d2 in column C2 d1 + d2
        if (half == 0) {
            cplx<float> ev[ROUNDS];
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) {
                const int e = t + q * NT, k = e / CH, d = e - k * CH;
                const int k2 = k / S1, k1 = k - k2 * S1, d2 = d / C1, d1 = d - d2 * C1;
                ev[q] = lds[(S2 * k1 + k2) * CHp + C2 * d1 + d2];
            }
            __syncthreads();            // every E is in registers: the odd chirps may overwrite the half-plane
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) {
                const int e = t + q * NT, s = e / CH, i = e - s * CH;
                lds[s * CHp + i] = held[q];
                held[q] = ev[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) {
                const int e = t + q * NT, k = e / CH, d = e - k * CH;
                const int k2 = k / S1, k1 = k - k2 * S1, d2 = d / C1, d1 = d - d2 * C1;
                const cplx<float> o = cmul(lds[(S2 * k1 + k2) * CHp + C2 * d1 + d2], tw_comb[d]);
                // fftshift over C: X[d] -> column d + C/2, X[d + C/2] -> column d
                __builtin_nontemporal_store(held[q] + o, out + (long)k * C + d + CH);
                __builtin_nontemporal_store(held[q] - o, out + (long)k * C + d);
            }
        }
    }
}

// planes of 2 x (at most 16384) cells with an even chirp count whose halves factor within the register radices
#define MMW_SPLIT_CT_SHAPES(X) X(512, 64) X(128, 256) X(1024, 32)

template <int S, int C>
int launch_rd_split_ct_sc(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, RawView rv) {
    // 512 threads; 128 x 256 (16-point radices) also fits the 128-VGPR budget of 1024 threads and is 6 % faster there
    constexpr int NT = (S == 128 && C == 256) ? 1024 : 512, CH = C / 2, S1 = mixct::best_n1(S), C1 = mixct::best_n1(CH);
    RdSplitArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.raw = rv;
    a.planes = planes;
    const void *p;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &p));
    a.win_s = (const float *)p;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, C, &p));
    a.win_c = (const float *)p;
    MMW_TRY(get_tw2_table(ctx, S, S1, &p));
    a.tw2_s = (const cplx<float> *)p;
    MMW_TRY(get_tw2_table(ctx, CH, C1, &p));
    a.tw2_c = (const cplx<float> *)p;
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, C, &p));
    a.tw_c = (const cplx<float> *)p;
    constexpr size_t lds_bytes = split_lds_bytes<S, C>();
    static_assert(lds_bytes <= 160 * 1024, "half plane must fit the LDS");
    auto kern = k_rd_split2_ct<S, C, NT>;
    MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)skip_planes(planes, rv)), dim3(NT), lds_bytes, ctx->stream, a);
    return check_launch("rd_split2_ct");
}

// ------------------------------------------------------------------ planes of 4 x 16384 cells (256 x 256, 512 x 128)
// The same idea one step further, along the RANGE axis (whole rows stay contiguous in memory): sample row s = 4 m + j
// belongs to quarter j, and with k = k' + (S/4) q
//     X[k][d] = sum_j W_4^(j q) ( W_S^(j k') G_j[k'][d] ),   G_j = range-Doppler transform of the quarter plane [S/4][C]
//                                                             under the windows hann(S)[4 m + j] x hann(C)[c].
// One workgroup per plane walks the four quarters through the LDS quarter plane [S/4][C | 1] (128 KB): load the quarter's
// rows (16-B loads), window, the levels of k_rd_mixed_ct, then quarters 0..2 move into REGISTERS (3 x 32 cells per thread
// at 512 threads = 192 VGPRs) and quarter 3 stays in the LDS for the final radix-4 combine, which stores four output rows
// per cell with the Doppler fftshift folded in.  HBM traffic = the algorithmic bytes; the two-kernel path these planes
// took before moved them twice (2.0 / 1.6 TB/s of algorithmic bytes).
struct RdSplit4Args {
    const void *in;
    void *out;
    const float *win_s, *win_c;             // Hann(S) (full length), Hann(C)
    const cplx<float> *tw2_s, *tw2_c;       // [S2q][S1q], [C2][C1] inter-level twiddles of the quarter / Doppler transforms
    const cplx<float> *tw_s;                // W_S^m, m < S: the combine uses W_S^(j k')
    RawView raw;                            // vskip only
    long planes;
};

template <int S, int C> constexpr size_t split4_lds_bytes() {
    return ((size_t)(S / 4) * (C | 1) + S / 4 + C) * sizeof(cplx<float>) + (size_t)(S + C) * sizeof(float);
}

template <int S, int C, int NT>
__global__ __launch_bounds__(NT) void k_rd_split4_ct(RdSplit4Args a) {
    constexpr int SQ = S / 4, Cp = C | 1;
    constexpr int S1 = mixct::best_n1(SQ), S2 = SQ / S1, C1 = mixct::best_n1(C), C2 = C / C1;
    static_assert(S % 4 == 0 && C % 2 == 0 && S1 > 0 && C1 > 0 && S1 != mixct::BIG_PRIME && C1 != mixct::BIG_PRIME, "split4 kernel shape");
    constexpr int QCELLS = SQ * C, ROUNDS = QCELLS / NT, PROUNDS = QCELLS / 2 / NT;
    static_assert(QCELLS % (2 * NT) == 0, "whole rounds");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(smem);          // [SQ][Cp]
    cplx<float> *tw_s = lds + SQ * Cp, *tw_c = tw_s + SQ;
    float *win_s = reinterpret_cast<float *>(tw_c + C), *win_c = win_s + S;
    const int tid = threadIdx.x;
    const long plane = skip_block_plane(blockIdx.x, a.raw);
    const f32x4 *in4 = reinterpret_cast<const f32x4 *>(a.in) + plane * (long)(S * C / 2);
    if constexpr (S2 > 1)
        for (int i = tid; i < SQ; i += NT) tw_s[i] = a.tw2_s[i];
    if constexpr (C2 > 1)
        for (int i = tid; i < C; i += NT) tw_c[i] = a.tw2_c[i];
    for (int i = tid; i < S; i += NT) win_s[i] = a.win_s[i];
    for (int i = tid; i < C; i += NT) win_c[i] = a.win_c[i];
    cplx<float> held[3][ROUNDS];        // G_0, G_1, G_2 of this thread's cells (k', dd)
    cplx<float> *out = reinterpret_cast<cplx<float> *>(a.out) + plane * (long)(S * C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // an opaque copy of the thread index per quarter keeps the four copies of the level code from sharing hoisted
        // address registers
        int t = tid;
        asm volatile("" : "+v"(t));
        __syncthreads();                // the previous quarter's readers are done (and the tables are in place)
        // ---- load + windows: quarter row m = global row 4 m + j; two batches keep the transient registers at 32
        constexpr int HB = PROUNDS / 2 > 0 ? PROUNDS / 2 : 1;
#pragma unroll
        for (int b0 = 0; b0 < PROUNDS; b0 += HB) {
            f32x4 raw[HB];
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int e = t + (b0 + q) * NT, m = (2 * e) / C, c = 2 * e - m * C;
                raw[q] = __builtin_nontemporal_load(in4 + ((long)(4 * m + j) * C + c) / 2);
            }
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int e = t + (b0 + q) * NT, m = (2 * e) / C, c = 2 * e - m * C;
                const float ws = win_s[4 * m + j], w0 = ws * win_c[c], w1 = ws * win_c[c + 1];
                lds[m * Cp + c] = cplx<float>{raw[q].x * w0, raw[q].y * w0};
                lds[m * Cp + c + 1] = cplx<float>{raw[q].z * w1, raw[q].w * w1};
            }
        }
        __syncthreads();
        // ---- quarter range transform (length SQ), Doppler transform (length C): the levels of k_rd_mixed_ct
        dft_level_ct<S1, NT, C, 1, S2, Cp, S2 * Cp, (S2 > 1)>(lds, tw_s, t);
        __syncthreads();
        if constexpr (S2 > 1) {
            dft_level_ct<S2, NT, C, 1, S1, S2 * Cp, Cp, false>(lds, nullptr, t);
            __syncthreads();
        }
        dft_level_ct<C1, NT, SQ, Cp, C2, 1, C2, (C2 > 1)>(lds, tw_c, t);
        __syncthreads();
        if constexpr (C2 > 1) {
            dft_level_ct<C2, NT, SQ, Cp, C1, C2, 1, false>(lds, nullptr, t);
            __syncthreads();
        }
        // range bin k' = k1 + S1 k2 sits in row S2 k1 + k2, Doppler bin d = d1 + C1 d2 in column C2 d1 + d2;
        // this thread's cells are (k', dd), dd the OUTPUT column: d = (dd - C/2) mod C
        if (j < 3) {
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) {
                const int e = t + q * NT, k = e / C, dd = e - k * C, d = (dd + C / 2) % C;
                const int k2 = k / S1, k1 = k - k2 * S1, d2 = d / C1, d1 = d - d2 * C1;
                held[j < 3 ? j : 0][q] = lds[(S2 * k1 + k2) * Cp + C2 * d1 + d2];
            }
        } else {
#pragma unroll
            for (int q = 0; q < ROUNDS; ++q) {
                const int e = t + q * NT, k = e / C, dd = e - k * C, d = (dd + C / 2) % C;
                const int k2 = k / S1, k1 = k - k2 * S1, d2 = d / C1, d1 = d - d2 * C1;
                const cplx<float> g3 = lds[(S2 * k1 + k2) * Cp + C2 * d1 + d2];
                // W_S^(j k'), j = 1, 2, 3 (table of the full length S)
                const cplx<float> t1 = a.tw_s[k], t2 = a.tw_s[2 * k], t3 = a.tw_s[3 * k];
                const cplx<float> A = held[0][q], B = cmul(held[1][q], t1), Cc = cmul(held[2][q], t2), D = cmul(g3, t3);
                const cplx<float> apc = A + Cc, amc = A - Cc, bpd = B + D, bmd = B - D;
                // W_4 = -i:  q = 1: A - i B - C + i D = (A - C) - i (B - D);   q = 3: (A - C) + i (B - D)
                const cplx<float> ibmd = cplx<float>{-bmd.y, bmd.x};         // i (B - D)
                __builtin_nontemporal_store(apc + bpd, out + (long)k * C + dd);
                __builtin_nontemporal_store(amc - ibmd, out + (long)(k + SQ) * C + dd);
                __builtin_nontemporal_store(apc - bpd, out + (long)(k + 2 * SQ) * C + dd);
                __builtin_nontemporal_store(amc + ibmd, out + (long)(k + 3 * SQ) * C + dd);
            }
        }
    }
}

#define MMW_SPLIT4_CT_SHAPES(X) X(256, 256) X(512, 128)

template <int S, int C>
int launch_rd_split4_ct_sc(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, RawView rv) {
    constexpr int NT = 512, SQ = S / 4, S1 = mixct::best_n1(SQ), C1 = mixct::best_n1(C);
    RdSplit4Args a{};
    a.in = d_in;
    a.out = d_out;
    a.raw = rv;
    a.planes = planes;
    const void *p;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &p));
    a.win_s = (const float *)p;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, C, &p));
    a.win_c = (const float *)p;
    MMW_TRY(get_tw2_table(ctx, SQ, S1, &p));
    a.tw2_s = (const cplx<float> *)p;
    MMW_TRY(get_tw2_table(ctx, C, C1, &p));
    a.tw2_c = (const cplx<float> *)p;
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, S, &p));
    a.tw_s = (const cplx<float> *)p;
    constexpr size_t lds_bytes = split4_lds_bytes<S, C>();
    static_assert(lds_bytes <= 160 * 1024, "quarter plane must fit the LDS");
    auto kern = k_rd_split4_ct<S, C, NT>;
    MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)skip_planes(planes, rv)), dim3(NT), lds_bytes, ctx->stream, a);
    return check_launch("rd_split4_ct");
}

#ifdef MMW_TU_MIXED_CT_C
bool rd_split_ct_supported(int S, int C) {
#define X(s, c) if (S == s && C == c) return true;
    MMW_SPLIT_CT_SHAPES(X)
    MMW_SPLIT4_CT_SHAPES(X)
#undef X
    return false;
}
int launch_rd_split_ct(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, int S, int C, RawView rv) {
#define X(s, c) if (S == s && C == c) return launch_rd_split_ct_sc<s, c>(ctx, d_in, d_out, planes, rv);
    MMW_SPLIT_CT_SHAPES(X)
#undef X
#define X(s, c) if (S == s && C == c) return launch_rd_split4_ct_sc<s, c>(ctx, d_in, d_out, planes, rv);
    MMW_SPLIT4_CT_SHAPES(X)
#undef X
    return set_error(MMW_ERR_UNSUPPORTED, "no split range-Doppler kernel for %dx%d", S, C);
}
#endif

}  // namespace mmw
