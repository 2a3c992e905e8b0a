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
#include "mmw_bf16x3.h"

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
    for (int a = 1; a <= N && a <= 32; ++a) {
        if (N % a) continue;
        // 63 x 115 (one shipped cfg): radix 23 in registers; 512 / 1024 samples: radix 32 (RegFFT<32>, ~70 VGPRs)
        if (a > MAX_RADIX && !(a == 23 && N == 115) && !(a == 32 && (N == 512 || N == 1024))) continue;
        const int b = N / a;
        if (b > a || (b > MAX_RADIX && !(b == 32 && N == 1024))) continue;
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
    const long lds = ((long)S * (C | 1) + S + C + (S + C + 1) / 2) * 8;
    const int wgs = (int)(160 * 1024 / lds);
    // 90 x 80 and 200 x 40 need ~100 VGPRs: 1024 threads would leave ONE workgroup per CU although two planes fit its
    // LDS; two 512-thread workgroups overlap each other's load and store phases (measured +12 % on both)
    if ((S == 90 && C == 80) || (S == 200 && C == 40)) return 512;
    if (S == 63 && C == 115) return 512;        // RegDFT<23> spills under the 128-VGPR cap of 1024 threads
    if (S == 512 && C == 32) return 512;        // RegFFT<32> + the carried next plane spill at 1024 threads (same speed)
    if (S == 254 && C == 50) return 512;        // the bfloat16 form of the 127-point level needs ~150 VGPRs (two tile jobs per wave)
    return wgs < 3 ? 1024 : (wgs < 6 ? 512 : 256);
}
}  // namespace mixct

constexpr size_t mixct_lds_base(int S, int C) {      // plane, inter-level twiddles, big-prime table, Hann tables, SYNC control words
    const bool big = mixct::best_n1(S) == mixct::BIG_PRIME || mixct::best_n1(C) == mixct::BIG_PRIME;
    return (((size_t)S * (C | 1) + S + C + (big ? mixct::BIG_PRIME : 0) + (S + C + 1) / 2) * sizeof(cplx<float>) + 16 + 64 + 15) & ~(size_t)15;
}
constexpr size_t MIXCT_ABF_BYTES = (size_t)2 * 2 * 4 * 3 * 64 * 16;      // the big-prime level's bfloat16 x 3 operand table
// (254 x 50 with 1024 threads -- 128 registers each, the big prime on the range axis -- spills in the bfloat16 form: measured
//  slower than the float32 MFMAs, 1.13 against 1.02 us per 12-antenna frame; it runs with 512 threads, two tile jobs per wave)
constexpr bool mixct_abf_fits(int S, int C) {
    const bool big = mixct::best_n1(S) == mixct::BIG_PRIME || mixct::best_n1(C) == mixct::BIG_PRIME;
    const bool room = mixct::threads_for(S, C) <= 512 || mixct::best_n1(S) != mixct::BIG_PRIME;     // (1024 threads: 128 registers)
    return big && room && mixct_lds_base(S, C) + MIXCT_ABF_BYTES <= 160 * 1024 - 256;
}
constexpr size_t mixct_lds_bytes(int S, int C, bool bf = false) { return mixct_lds_base(S, C) + (bf && mixct_abf_fits(S, C) ? MIXCT_ABF_BYTES : 0); }
// which launches take the bfloat16 form (A/B at 2048 frames, tools/bigprime_ab.sh -> profiles/r04_bigprime_ab.log): the stand-alone
// range-Doppler of 63 x 127 and 254 x 50 (+8-10 % at 12 antennas, +2-7 % at 8).  NOT the producer of the device-synchronised
// chain: there the two forms are within 2 % of each other for 63 x 127 / 254 x 50 (the chain is paced by the angle stage and by
// the producer's CU share, not by this level) and the bfloat16 form's 96 KB table costs 127 x 32 its second workgroup per CU
// (12 antennas: 0.67 against 0.51 us/frame) -- so every chain producer, raw-cube or not, runs the float32 form and the two
// stay bit-identical.  127 x 32 stand-alone: four small float32-form workgroups per CU are faster.
constexpr bool mixct_use_bf(int S, int C, bool sync) { return mixct_abf_fits(S, C) && !sync && !(S == 127 && C == 32); }

struct RdMixedCtArgs {
    const void *in;             // complex64 planes
    void *out;                  // complex64 planes
    long in_plane_stride;       // complex elements between consecutive input planes
    const float *win_s, *win_c;
    const cplx<float> *tw2_s, *tw2_c;       // [S2][S1] and [C2][C1]: W_S^(n2 k1), W_C^(m2 k1)
    const float *cs_big;        // coefficient table of the big-prime level (nullptr when no axis has one)
    const void *abf_big;        // the same coefficients as bfloat16 x 3 MFMA operands (get_bigprime_bf16_table; nullptr: float32 MFMAs)
    RawView raw;
    long planes;
    long long *clk;             // diagnostics (MMW_PHASE_CLOCKS=1): s_memtime at the phase boundaries of workgroup 0
    ChainSync cs;               // MODE 2: the chain's device-synchronised hand-over (mmw_fft_fused.h); `out` is the ring
    float *l1;                  // MODE 0 / 1, optional: l1[plane] = sum |re| + |im| of the windowed plane (the error-bound
                                // scale of mmw_angle_argmax_exact, as the 256 x 128 kernel's L1N variant: no extra pass)
};

__device__ __forceinline__ void phase_mark(long long *clk, int i, int tid) {
    if (clk && blockIdx.x == 0 && tid == 0) clk[i] = (long long)__builtin_amdgcn_s_memtime();
}

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

// A level whose radix is a prime P = 127 too large for registers (three shipped cfgs), same contract as dft_level_ct.
// Real-symmetric direct form as two real matrix products on the matrix cores, in place:
//   with s_j = x_j + x_{P-j}, d_j = x_j - x_{P-j} (j = 1 .. H = 63), s_0 = x_0, d_0 = 0 and the columns n = (group, re | im)
//     Dc[k][n] = sum_{j=0..63} cos(2 pi jk/P) s_j[n],   Ds[k][n] = sum_j sin(2 pi jk/P) d_j[n]      (k = 0 .. 63)
//     X_k = Dc - i Ds,   X_{P-k} = Dc + i Ds
//   -- a [64 x 64] x [64 x 2 N_GROUPS] real GEMM each, run as v_mfma_f32_32x32x2_f32 tiles (exact float32 FMA chains):
//   a wave owns one 32 (k) x 32 (n) output tile; the B operand (s_j, d_j) is formed from two LDS reads of the plane, the A
//   operand is cos / sin((j k mod P) 2 pi / P) out of a 2 P-float LDS table, its index advanced by 2k mod P per step.
// Round 2's first version kept the outputs in VALU registers and fetched 8..16 coefficients per j by wave-uniform scalar
// loads: phase clocks showed that loop at 65 k of the 93 k clocks of a 254 x 50 plane (the 32 KB table overflows the
// 16 KB scalar cache; ~500 clocks per load) against ~16 k for its FMAs.  The MFMA form needs no coefficient traffic.
// cst (LDS): cos(2 pi m / P), m = 0 .. P-1, then the P sines.
typedef float mixct_v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float lane_xor1(float v) {      // the value of lane ^ 1 (quad_perm [1, 0, 3, 2])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
// abf != nullptr: the two products on v_mfma_f32_32x32x16_bf16 with exact three-way operand splits (mmw_bf16x3.h) -- the
// float32 MFMAs above cannot overlap with the vector work next to them (operand reads, s / d, index updates: 64 x 64 MFMA cycles
// PLUS ~1.3 k vector cycles per tile), the bfloat16 ones can, and six of nine partial products over 16 values of k cost 6 x 32
// cycles where 8 float32 MFMAs cost 512.  The coefficient pieces are precomputed in operand layout (abf[matrix][kt][t][piece][lane],
// 16 bytes each: get_bigprime_bf16_table); the data operand is split on the fly.  Error budget: all SMALL partial products of
// the 64-term sum (a1 b2, a2 b1, a1 b3, a2 b2, a3 b1: <= 2^-7 of the leading ones) are accumulated FIRST, the four MFMAs of the
// leading pieces last, so only those 4 x 17 float32 additions round at the scale of the result: 68 + 20 x 17 x 2^-7 (small
// phase) + 3 (dropped products) + pair sums and the combine = under 80 eps sum |x_j|, the budget rd_error_ulps() books for this
// level in either form.  (The leading pieces of the data are formed twice -- once per phase -- instead of being kept: registers.)
template <int P, int NT, int N_INNER, int INNER_STRIDE, int N_OUTER, int OUTER_STRIDE, int ESTRIDE, bool TW>
__device__ __forceinline__ void dft_level_bigprime_ct(cplx<float> *lds, const cplx<float> *tw2, const float *cst, int tid,
                                                      long long *clk = nullptr, const void *abf = nullptr) {
    constexpr int H = (P - 1) / 2, N_GROUPS = N_INNER * N_OUTER, NCOL = 2 * N_GROUPS, NTILES = (NCOL + 31) / 32;
    constexpr int JOBS = 2 * NTILES, NW = NT / 64, JPW = (JOBS + NW - 1) / NW;      // jobs per wave: job = wave, wave + NW, ...
    static_assert(H + 1 == 64, "two 32-row output tiles");
    static_assert(JPW <= 2, "at most two output tiles per wave (their accumulators wait in registers for the barrier)");
    float *lf = reinterpret_cast<float *>(lds);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kk = lane >> 5;
    mixct_v16f dcs[JPW], dss[JPW];
    // the columns of job q of this wave: n = (group, re | im), float index of the group's element 0 for this component
    auto job_cols = [&](int job, int &kt, bool &valid, int &comp, int &o, int &boff) {
        kt = job / NTILES;
        const int nt = job - kt * NTILES, n = nt * 32 + (lane & 31);
        valid = n < NCOL;
        const int g = valid ? n >> 1 : 0;
        comp = n & 1;
        o = g / N_INNER;
        const int i = g - o * N_INNER;
        boff = 2 * (i * INNER_STRIDE + o * OUTER_STRIDE) + comp;
    };
#pragma unroll
    for (int q = 0; q < JPW; ++q) {
        const int job = wave + q * NW;
        int kt, comp, o, boff;
        bool valid;
        job_cols(job, kt, valid, comp, o, boff);
        mixct_v16f dc = {0}, ds = {0};
        if (abf) {
            if (job < JOBS) {
                const u32x4 *tc = reinterpret_cast<const u32x4 *>(abf) + (size_t)(kt * 4) * 3 * 64 + lane;          // cosines
                const u32x4 *ts = reinterpret_cast<const u32x4 *>(abf) + (size_t)((2 + kt) * 4) * 3 * 64 + lane;    // sines
                const float *pa = lf + boff + 16 * ESTRIDE * kk;             // x_j,     j = 16 t + 8 kk + i
                const float *pb = lf + boff + 2 * ESTRIDE * (P - 8 * kk);    // x_{P-j}
                const float *p0 = kk ? pb : pa;                              // (j = 0 has no partner: s_0 = x_0, d_0 = 0)
                // SUM: s_j = x_j + x_{P-j}, else d_j = x_j - x_{P-j}
                auto gather = [&](int t, auto SUM, float (&v)[8]) {
                    constexpr bool sum = decltype(SUM)::value;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float xa = pa[2 * ESTRIDE * (16 * t + i)];
                        const float xb = i == 0 ? (t == 0 ? p0 : pb - 2 * ESTRIDE * 16 * t)[0] : pb[-2 * ESTRIDE * (16 * t + i)];
                        const bool first = i == 0 && t == 0 && kk == 0;
                        v[i] = sum ? (first ? xa : xa + xb) : (first ? 0.f : xa - xb);
                    }
                };
                auto piece = [](const u32x4 *tab, int idx) { return __builtin_bit_cast(bf16x8, tab[idx * 64]); };
                // One matrix at a time (registers), TWO accumulators taking the MFMAs in turn: a wave issues in order, so a chain
                // of MFMAs on one accumulator stalls it for the whole latency of each -- nothing else of the wave, not even the
                // next step's operand arithmetic, gets issued meanwhile.
                auto product = [&](const u32x4 *tab, auto SUM) {
                    mixct_v16f acc0 = {0}, acc1 = {0};
                    bf16x8 lead[4];                                         // the data's leading pieces, for phase 2
#pragma unroll
                    for (int t = 0; t < 4; ++t) {                           // phase 1: every partial product below the leading one
                        float v[8];
                        gather(t, SUM, v);
                        bf16x8 b2, b3;
                        const bf16x8 a1 = piece(tab, t * 3), a2 = piece(tab, t * 3 + 1), a3 = piece(tab, t * 3 + 2);
                        split_bf16x3(v, lead[t], b2, b3);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, lead[t], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, lead[t], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc0, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);                  // (keep the steps apart: unrolled freely, all 24 coefficient reads come first)
                    }
#pragma unroll
                    for (int t = 0; t < 4; t += 2) {                        // phase 2: the leading pieces
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(piece(tab, t * 3), lead[t], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(piece(tab, t * 3 + 3), lead[t + 1], acc0, 0, 0, 0);
                    }
                    return acc0 + acc1;
                };
                dc = product(tc, std::true_type{});
                ds = product(ts, std::false_type{});
            }
        } else if (job < JOBS) {
            const int k = kt * 32 + (lane & 31);                // this lane's row of the A operand
            int idx = kk ? k : 0;                               // (j k) mod P, j = 2 t + kk
            const int step = 2 * k >= P ? 2 * k - P : 2 * k;
            const float *pa = lf + boff + 2 * ESTRIDE * kk;             // x_j
            const float *pb = lf + boff + 2 * ESTRIDE * (P - kk);       // x_{P-j}
#pragma unroll 8
            for (int t = 0; t < 32; ++t) {
                const float xa = pa[4 * ESTRIDE * t];
                const float xb = (t == 0 && kk == 0) ? xa : pb[-4 * ESTRIDE * t];
                const float sj = (t == 0 && kk == 0) ? xa : xa + xb, dj = xa - xb;
                const float c = cst[idx], sn = cst[P + idx];
                dc = __builtin_amdgcn_mfma_f32_32x32x2f32(c, sj, dc, 0, 0, 0);
                ds = __builtin_amdgcn_mfma_f32_32x32x2f32(sn, dj, ds, 0, 0, 0);
                idx += step;
                if (idx >= P) idx -= P;
            }
        }
        dcs[q] = dc;
        dss[q] = ds;
    }
    __syncthreads();            // every wave has read the x it needs: the groups may be overwritten
    phase_mark(clk, 9, tid);
#pragma unroll
    for (int q = 0; q < JPW; ++q) {
        const int job = wave + q * NW;
        int kt, comp, o, boff;
        bool valid;
        job_cols(job, kt, valid, comp, o, boff);
        if (job < JOBS) {
            const float sgn = comp ? -1.f : 1.f;
            // C/D map of the 32x32 shapes: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                const float pr = lane_xor1(dss[q][r]);          // Ds of the other component
                float lo = dcs[q][r] + sgn * pr, hi = dcs[q][r] - sgn * pr;     // re: Dc.re + Ds.im | im: Dc.im - Ds.re, and mirrored
                if constexpr (TW) {
                    const float lo_o = lane_xor1(lo), hi_o = lane_xor1(hi);
                    const cplx<float> wl = tw2[o * P + k], wh = tw2[o * P + (k ? P - k : 0)];
                    lo = lo * wl.x - sgn * lo_o * wl.y;          // re: a.x b.x - a.y b.y | im: a.y b.x + a.x b.y
                    hi = hi * wh.x - sgn * hi_o * wh.y;
                }
                if (valid) {
                    lf[boff + 2 * ESTRIDE * k] = lo;
                    if (k > 0) lf[boff + 2 * ESTRIDE * (P - k)] = hi;
                }
            }
        }
    }
}

// level dispatch: register-resident radix or the big prime
template <int R, int NT, int N_INNER, int INNER_STRIDE, int N_OUTER, int OUTER_STRIDE, int ESTRIDE, bool TW>
__device__ __forceinline__ void dft_level_any_ct(cplx<float> *lds, const cplx<float> *tw2, const float *cs, int tid,
                                                 long long *clk = nullptr, const void *abf = nullptr) {
    if constexpr (R == mixct::BIG_PRIME)
        dft_level_bigprime_ct<R, NT, N_INNER, INNER_STRIDE, N_OUTER, OUTER_STRIDE, ESTRIDE, TW>(lds, tw2, cs, tid, clk, abf);
    else
        dft_level_ct<R, NT, N_INNER, INNER_STRIDE, N_OUTER, OUTER_STRIDE, ESTRIDE, TW>(lds, tw2, tid);
}

// PERSIST (virtual-array cubes only): the grid is one residency of workgroups, each walks planes item, item + grid, ...
// and the NEXT plane's global loads are issued into registers before this plane's levels run, so the HBM latency of the
// load phase (~10 k of the 20..50 k clocks of a plane when the LDS holds only one or two planes per CU, phase clocks
// above) hides behind the arithmetic instead of serialising with it.
// MODE 2 (SYNC): the producer side of the device-synchronised chain for these shapes, same protocol as the 256 x 128
// kernel (k_rd_fused_256x128_persist<SYNC>): items (frame, live antenna) from a ticket counter, the output plane goes to
// the ring slot of its frame with sc1 (write-through) stores once the slot's previous frame has been consumed, every
// storing wave drains its stores, and one lane bumps the slot's counter after the workgroup barrier.
// BF: the big-prime level on bfloat16 x 3 MFMAs (a separate instantiation: the form needs ~150 registers and 48 KB of LDS,
// which the float32 form's callers -- e.g. four small 127 x 32 workgroups per CU -- must not pay for)
template <int S, int C, int NT, int MODE, bool BF = false>
__global__ __launch_bounds__(NT) void k_rd_mixed_ct(RdMixedCtArgs a) {
    constexpr bool PERSIST = MODE >= 1, SYNC = MODE >= 2, RAWSYNC = MODE == 3;     // MODE 3: SYNC on the raw [F][nrx][S][ntx C] cube
    constexpr int S1 = mixct::best_n1(S), S2 = S / S1, C1 = mixct::best_n1(C), C2 = C / C1;
    constexpr int Cp = C | 1, CELLS = S * C;
    static_assert(S1 > 0 && C1 > 0, "no factorisation within the register-resident radices");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(smem);          // [S][Cp]
    cplx<float> *tw_s = lds + S * Cp, *tw_c = tw_s + S;
    float *cst = reinterpret_cast<float *>(tw_c + C);                  // big-prime cos / sin table (2 P floats)
    constexpr bool BIG = S1 == mixct::BIG_PRIME || C1 == mixct::BIG_PRIME;
    // Hann tables in the LDS: a global load in the plane loop would wait (in-order vmcnt) for the previous plane's stores
    float *win_s = cst + (BIG ? 2 * mixct::BIG_PRIME : 0), *win_c = win_s + S;
    int *lds_ctl = reinterpret_cast<int *>(win_c + C);                 // SYNC: [0] / [1] tickets (double buffered), [2] abort
    float *lds_l1 = reinterpret_cast<float *>(lds_ctl + 4);            // one partial L1 norm per wave
    const int tid = threadIdx.x;
    const bool raw = !PERSIST && a.raw.ntx > 1;
    const int ntx = raw ? a.raw.ntx : (RAWSYNC ? a.cs.ntx : 1);
    constexpr bool PAIRED = C % 2 == 0 && !RAWSYNC;         // two adjacent chirps per 16-B load (virtual-array cubes)
    constexpr int PAIRS = CELLS / 2, PROUNDS = PAIRED ? (PAIRS + NT - 1) / NT : 1, EROUNDS = (CELLS + NT - 1) / NT;
    f32x4 pre4[PROUNDS];
    cplx<float> pre1[EROUNDS];
    float l1_acc = 0.f;
    auto fetch = [&](const cplx<float> *in) {
        if constexpr (PAIRED) {
            const f32x4 *in4 = reinterpret_cast<const f32x4 *>(in);
#pragma unroll
            for (int q = 0; q < PROUNDS; ++q) {
                const int e = tid + q * NT;
                if (PAIRS % NT == 0 || q + 1 < PROUNDS || e < PAIRS) pre4[q] = __builtin_nontemporal_load(in4 + e);
            }
        } else {
#pragma unroll
            for (int q = 0; q < EROUNDS; ++q) {
                const int e = tid + q * NT;
                if (CELLS % NT == 0 || q + 1 < EROUNDS || e < CELLS)
                    pre1[q] = __builtin_nontemporal_load(in + (RAWSYNC ? (long)e * ntx : (long)e));       // raw: every ntx-th chirp
            }
        }
    };
    auto stash = [&](int t) {                   // registers x Hann(S) x Hann(C) -> LDS
        if constexpr (PAIRED) {
#pragma unroll
            for (int q = 0; q < PROUNDS; ++q) {
                const int e = t + q * NT;
                if (PAIRS % NT == 0 || q + 1 < PROUNDS || e < PAIRS) {
                    const int s = (2 * e) / C, c = (2 * e) - s * C;
                    const f32x4 v = pre4[q];
                    const float ws = win_s[s], w0 = ws * win_c[c], w1 = ws * win_c[c + 1];
                    const cplx<float> y0 = cplx<float>{v.x * w0, v.y * w0}, y1 = cplx<float>{v.z * w1, v.w * w1};
                    lds[s * Cp + c] = y0;
                    lds[s * Cp + c + 1] = y1;
                    if constexpr (!SYNC) l1_acc += (fabsf(y0.x) + fabsf(y0.y)) + (fabsf(y1.x) + fabsf(y1.y));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < EROUNDS; ++q) {
                const int e = t + q * NT;
                if (CELLS % NT == 0 || q + 1 < EROUNDS || e < CELLS) {
                    const int s = e / C, c = e - s * C;
                    const cplx<float> y = pre1[q] * (win_s[s] * win_c[c]);
                    lds[s * Cp + c] = y;
                    if constexpr (!SYNC) l1_acc += fabsf(y.x) + fabsf(y.y);
                }
            }
        }
    };
    const cplx<float> *in_base = reinterpret_cast<const cplx<float> *>(a.in);
    const ChainSync &cs = a.cs;
    long item = blockIdx.x;
    const long n_items = SYNC ? (long)cs.n_frames * cs.v_live : (PERSIST ? skip_planes(a.planes, a.raw) : 0);
    // SYNC: item -> plane f * V + v (the end antennas are skipped when the chain drops them; raw cubes: the host's
    // rx-major order of the live antennas, packed in cs.vmap) and its first sample
    auto sync_plane = [&](long it) {
        const long f = it / cs.v_live;
        const int vi = (int)(it - f * cs.v_live);
        if constexpr (RAWSYNC) return f * cs.V + (long)((cs.vmap >> (4 * vi)) & 15);
        else return f * cs.V + vi + (cs.vskip > 2 ? 1 : 0);
    };
    auto sync_src = [&](long pl) {
        if constexpr (RAWSYNC) return raw_plane(in_base, pl, S, C, RawView{cs.ntx, cs.nrx, 0});
        else return in_base + pl * a.in_plane_stride;
    };
    long plane;
    if constexpr (SYNC) {
        if (tid == 0) {
            lds_ctl[0] = (int)(__hip_atomic_fetch_add(cs.ctl + CTL_RD_TICKET, 1u, MMW_RLX_AGENT) - cs.rd_base);
            lds_ctl[2] = 0;
        }
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(lds_ctl[0]);
        if (item >= n_items) return;
        plane = sync_plane(item);
    } else if (raw) {
        plane = raw_block_plane(blockIdx.x, a.planes, a.raw);
        if (plane < 0 || skip_raw_plane(plane, a.raw)) return;
    } else plane = skip_block_plane(item, a.raw);
    if constexpr (SYNC) fetch(sync_src(plane));
    else if (PERSIST || (PAIRED && !raw)) fetch(in_base + plane * a.in_plane_stride);
    int iter = 0;
    if constexpr (S2 > 1)
        for (int i = tid; i < S; i += NT) tw_s[i] = a.tw2_s[i];
    if constexpr (C2 > 1)
        for (int i = tid; i < C; i += NT) tw_c[i] = a.tw2_c[i];
    if constexpr (BIG)
        for (int i = tid; i < 2 * mixct::BIG_PRIME; i += NT) cst[i] = a.cs_big[i];
    // the operand table of the bfloat16 form of the big-prime level lives in the LDS: a global load inside the level would
    // have to wait (in-order vmcnt) for the next plane's prefetch
    [[maybe_unused]] u32x4 *abf_l = reinterpret_cast<u32x4 *>(smem + mixct_lds_base(S, C));
    const void *abf = nullptr;
    if constexpr (BF && mixct_abf_fits(S, C)) {
        for (int i = tid; i < (int)(MIXCT_ABF_BYTES / 16); i += NT) abf_l[i] = reinterpret_cast<const u32x4 *>(a.abf_big)[i];
        abf = abf_l;
    }
    for (int i = tid; i < S; i += NT) win_s[i] = a.win_s[i];
    for (int i = tid; i < C; i += NT) win_c[i] = a.win_c[i];
    __syncthreads();
    while (true) {
        // PERSIST: hide the thread index from loop-invariant code motion -- hoisting every plane-invariant index and window
        // product out of the plane loop costs ~60 VGPRs (spills at 254 x 50 and 120 x 126)
        int t = tid;
        if constexpr (PERSIST) asm volatile("" : "+v"(t));
        phase_mark(a.clk, 0, tid);
        // SYNC: ring slot of this plane's frame; thread 0 reads the slot's consumer counter and draws the next ticket now,
        // un-waited -- both values are needed only after the first level
        int slot = 0;
        long ring_plane = 0;
        unsigned free_target = 0, free_seen = 0, next_ticket = 0;
        if constexpr (SYNC) {
            const long f = item / cs.v_live;
            const unsigned g = cs.s0 + (unsigned)f;
            slot = (int)(g % (unsigned)cs.ring);
            free_target = (cs.u0 + g / (unsigned)cs.ring) * (unsigned)cs.tiles;
            ring_plane = (long)slot * cs.V + (plane - f * cs.V);
            if (tid == 0) {
                free_seen = __hip_atomic_load(cs.ctl + CTL_CNT + CTL_RING_MAX + slot, MMW_RLX_AGENT);
                next_ticket = __hip_atomic_fetch_add(cs.ctl + CTL_RD_TICKET, 1u, MMW_RLX_AGENT) - cs.rd_base;
            }
        }
        // ---- load + windows
        if (!PERSIST && (!PAIRED || raw)) {
            // element loads (odd C, or the raw cube's tx-strided view): a few at a time, straight into the LDS
            const cplx<float> *in = raw ? raw_plane(in_base, plane, S, C, a.raw) : in_base + plane * a.in_plane_stride;
            const short2 *in16 = reinterpret_cast<const short2 *>(a.in) + (raw ? raw_plane_off(plane, S, C, a.raw) : 0);
            const bool i16 = raw && a.raw.i16;          // (uniform) int16 (I, Q) cells, converted here
#pragma unroll 4
            for (int q = 0; q < EROUNDS; ++q) {
                const int e = t + q * NT;
                if (CELLS % NT == 0 || q + 1 < EROUNDS || e < CELLS) {
                    const int s = e / C, c = e - s * C;
                    cplx<float> x;
                    if (i16) {
                        const short2 v = in16[(long)e * ntx];
                        x = cplx<float>{(float)v.x, (float)v.y};
                    } else
                        x = __builtin_nontemporal_load(in + (long)e * ntx);
                    const cplx<float> y = x * (win_s[s] * win_c[c]);
                    lds[s * Cp + c] = y;
                    if constexpr (!SYNC) l1_acc += fabsf(y.x) + fabsf(y.y);
                }
            }
        } else
            stash(t);
        // (a big-prime FIRST level keeps two MFMA accumulators, operand pieces and coefficient pieces in registers: the next
        //  plane's loads are issued behind it -- the three remaining levels and the store still cover their flight)
        constexpr bool FETCH_LATE = BF && S1 == mixct::BIG_PRIME && mixct_abf_fits(S, C);
        if constexpr (PERSIST && !SYNC && !FETCH_LATE) {
            if (item + gridDim.x < n_items) fetch(in_base + skip_block_plane(item + gridDim.x, a.raw) * a.in_plane_stride);
        }
        if constexpr (!SYNC) {
            if (a.l1) {             // (uniform) wave partial by a fixed shuffle tree, then a fixed-order sum: deterministic
                for (int d = 32; d >= 1; d >>= 1) l1_acc += __shfl_xor(l1_acc, d, 64);
                if ((tid & 63) == 0) lds_l1[tid >> 6] = l1_acc;
            }
            l1_acc = 0.f;
        }
        __syncthreads();
        if constexpr (!SYNC) {
            if (a.l1 && tid == 0) {
                float acc = 0.f;
                for (int i = 0; i < NT / 64; ++i) acc += lds_l1[i];
                a.l1[plane] = acc;
            }
        }
        phase_mark(a.clk, 1, tid);
        // ---- range axis: sample s = S2 n1 + n2 lives in row s.  A: groups (column, n2), radix S1; B: groups (column, k1), radix S2
        dft_level_any_ct<S1, NT, C, 1, S2, Cp, S2 * Cp, (S2 > 1)>(lds, tw_s, cst, t, a.clk, abf);
        if constexpr (SYNC) {
            if (tid == 0) lds_ctl[(iter + 1) & 1] = (int)next_ticket;      // the other waves learn it at this barrier
        }
        __syncthreads();
        if constexpr (PERSIST && !SYNC && FETCH_LATE) {
            if (item + gridDim.x < n_items) fetch(in_base + skip_block_plane(item + gridDim.x, a.raw) * a.in_plane_stride);
        }
        long next_item = 0;
        if constexpr (SYNC) {
            next_item = __builtin_amdgcn_readfirstlane(lds_ctl[(iter + 1) & 1]);
            if (next_item < n_items) fetch(sync_src(sync_plane(next_item)));    // in flight behind the levels
        }
        phase_mark(a.clk, 2, tid);
        if constexpr (S2 > 1) {
            dft_level_ct<S2, NT, C, 1, S1, S2 * Cp, Cp, false>(lds, nullptr, t);
            __syncthreads();
        }
        phase_mark(a.clk, 3, tid);
        // ---- Doppler axis: chirp c = C2 m1 + m2 lives in column c; lanes walk the rows (odd pitch: conflict free)
        dft_level_any_ct<C1, NT, S, Cp, C2, 1, C2, (C2 > 1)>(lds, tw_c, cst, t, C1 == mixct::BIG_PRIME && S1 != mixct::BIG_PRIME ? a.clk : nullptr, abf);
        __syncthreads();
        phase_mark(a.clk, 4, tid);
        if constexpr (C2 > 1) {
            dft_level_ct<C2, NT, S, Cp, C1, C2, 1, false>(lds, nullptr, t);
            __syncthreads();
        }
        phase_mark(a.clk, 5, tid);
        // ---- store: range bin k = k1 + S1 k2 sits in row S2 k1 + k2, Doppler bin d = d1 + C1 d2 in column C2 d1 + d2;
        //      fftshift: out[(d + C/2) % C] = X[d]
        cplx<float> *out = reinterpret_cast<cplx<float> *>(a.out) + plane * CELLS;
        [[maybe_unused]] auto ring_rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0, 0x00020000);
        [[maybe_unused]] unsigned ring_soff = 0;
        if constexpr (SYNC) {
            // the slot's previous frame must have been consumed before the first store
            if (tid == 0 && !chain_wait(cs.ctl + CTL_CNT + CTL_RING_MAX + slot, free_target, free_seen, cs.ctl, cs.timeout, cs.naps_rd))
                lds_ctl[2] = 1;
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(lds_ctl[2])) return;         // timed out / aborted: no stores, everybody leaves
            ring_rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)cs.ring * (unsigned)cs.V * (unsigned)(CELLS * 8)), 0x00020000);
            ring_soff = (unsigned)ring_plane * (unsigned)(CELLS * 8);
        }
        constexpr int HALF = C / 2, ROUNDS = (CELLS + NT - 1) / NT;
#pragma unroll 4
        for (int q = 0; q < ROUNDS; ++q) {
            const int e = t + q * NT;
            if (CELLS % NT == 0 || q + 1 < ROUNDS || e < CELLS) {
                const int k = e / C, dd = e - k * C;
                int d = dd - HALF;
                if (d < 0) d += C;
                const int k2 = k / S1, k1 = k - k2 * S1, d2 = d / C1, d1 = d - d2 * C1;
                const cplx<float> v = lds[(S2 * k1 + k2) * Cp + C2 * d1 + d2];
                if constexpr (SYNC)     // sc1 (write-through, aux = 16): plane base in the scalar offset, lane part in voffset
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), ring_rs, (unsigned)e * 8u, ring_soff, 16);
                else __builtin_nontemporal_store(v, out + e);
            }
        }
        phase_mark(a.clk, 6, tid);
        if constexpr (!PERSIST) break;
        if constexpr (SYNC) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its stores ...
            __syncthreads();                                         // ... before one lane publishes the plane
            if (tid == 0) __hip_atomic_fetch_add(cs.ctl + CTL_CNT + slot, 1u, MMW_RLX_AGENT);
            item = next_item;
            ++iter;
            if (item >= n_items) break;
            plane = sync_plane(item);
            continue;           // (the barrier above also covers the LDS reuse)
        }
        item += gridDim.x;
        if (item >= n_items) break;
        plane = skip_block_plane(item, a.raw);
        __syncthreads();        // the store has read the LDS plane: the next stash may overwrite it
    }
}

// the non-power-of-two planes of the shipped cfgs (tests/golden/cfg_scalars.json) that need no prime radix above 20
#define MMW_MIXED_CT_SHAPES_A(X) X(63, 70) X(63, 100) X(64, 40) X(70, 40) X(90, 80) X(100, 30) X(254, 50) X(127, 32) X(512, 32) X(512, 8)
// the remaining power-of-two planes that fit the LDS (k_rd_lds' list): same kernel, and with it the synchronised chain
#define MMW_MIXED_CT_SHAPES_C(X) X(32, 32) X(64, 32) X(128, 32) X(256, 32) X(32, 64) X(256, 64) X(32, 128) X(64, 128)
#define MMW_MIXED_CT_SHAPES_B(X) X(90, 100) X(100, 100) X(120, 126) X(130, 50) X(200, 40) X(63, 127) X(64, 64) X(128, 64) X(128, 128) X(63, 115)

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

// cos(2 pi m / P), m = 0 .. P-1, then the P sines: the coefficient table of dft_level_bigprime_ct, cached per context
inline int get_bigprime_table(mmw_ctx *ctx, int P, const void **out) {
    const auto key = std::make_tuple(201, P, 0);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return MMW_OK;
    }
    std::vector<float> h(2 * (size_t)P);
    for (int m = 0; m < P; ++m) {
        const long double ang = 2.0L * M_PIl * (long double)m / (long double)P;
        h[m] = (float)cosl(ang);
        h[P + m] = (float)sinl(ang);
    }
    void *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return set_error(MMW_ERR_NOMEM, "hipMalloc for DFT coefficient table failed");
    MMW_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tables[key] = d;
    *out = d;
    return MMW_OK;
}

// The same coefficients as operands of v_mfma_f32_32x32x16_bf16, split exactly into three bfloat16 pieces (truncation, as
// split_bf16x3 does on the device): [matrix: cos, sin][kt][t][piece][lane] x 16 bytes -- lane (r = lane & 31, h = lane >> 5)
// holds A[row k = 32 kt + r][j = 16 t + 8 h + i], i = 0 .. 7 (element 2 q in the low half of dword q).  48 KB, cached per context.
inline int get_bigprime_bf16_table(mmw_ctx *ctx, int P, const void **out) {
    const auto key = std::make_tuple(202, P, 0);
    auto it = ctx->tables.find(key);
    if (it != ctx->tables.end()) {
        *out = it->second;
        return MMW_OK;
    }
    std::vector<uint32_t> h((size_t)2 * 2 * 4 * 3 * 64 * 4);
    auto bits = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; };
    auto val = [](uint32_t u) { float v; std::memcpy(&v, &u, 4); return v; };
    for (int m = 0; m < 2; ++m)
        for (int kt = 0; kt < 2; ++kt)
            for (int t = 0; t < 4; ++t)
                for (int lane = 0; lane < 64; ++lane) {
                    const int k = kt * 32 + (lane & 31), hh = lane >> 5;
                    uint16_t pc[3][8];
                    for (int i = 0; i < 8; ++i) {
                        const int j = 16 * t + 8 * hh + i;
                        const long double ang = 2.0L * M_PIl * (long double)(((long)j * k) % P) / (long double)P;
                        const float a = m ? (float)sinl(ang) : (float)cosl(ang);
                        const uint32_t a1 = bits(a) & 0xffff0000u;
                        const float ra = a - val(a1);
                        const uint32_t a2 = bits(ra) & 0xffff0000u;
                        const float sa = ra - val(a2);
                        pc[0][i] = (uint16_t)(a1 >> 16);
                        pc[1][i] = (uint16_t)(a2 >> 16);
                        pc[2][i] = (uint16_t)(bits(sa) >> 16);
                    }
                    for (int p = 0; p < 3; ++p)
                        for (int q = 0; q < 4; ++q)
                            h[((((size_t)(m * 2 + kt) * 4 + t) * 3 + p) * 64 + lane) * 4 + q] = (uint32_t)pc[p][2 * q] | ((uint32_t)pc[p][2 * q + 1] << 16);
                }
    void *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(uint32_t)) != hipSuccess) return set_error(MMW_ERR_NOMEM, "hipMalloc for DFT coefficient table failed");
    MMW_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tables[key] = d;
    *out = d;
    return MMW_OK;
}

// cs != nullptr: the device-synchronised producer (MODE 2) on sync_cus CUs; *sync_grid returns the workgroups launched
// (each draws one ticket past the end, the host mirrors that in its counter base).  sync_grid only: just report the grid.
template <int S, int C>
int launch_rd_mixed_ct_sc(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, RawView rv,
                          const ChainSync *cs = nullptr, int sync_cus = 0, int *sync_grid = nullptr, bool query_only = false,
                          float *d_l1 = nullptr) {
    constexpr int NT = mixct::threads_for(S, C), S1 = mixct::best_n1(S), C1 = mixct::best_n1(C);
    RdMixedCtArgs a{};
    a.in = d_in;
    a.out = d_out;
    a.in_plane_stride = in_plane_stride;
    a.raw = rv;
    a.planes = planes;
    a.l1 = d_l1;
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
        if (opt_int(ctx, "MMW_BIGPRIME_BF16", 1)) {         // (0: the float32 MFMA form)
            MMW_TRY(get_bigprime_bf16_table(ctx, mixct::BIG_PRIME, &p));
            a.abf_big = p;
        }
    }
    const bool sync_call = cs || query_only;
    const bool raw_call = (cs && cs->ntx > 1) || (query_only && rv.ntx > 1);        // the raw-cube producer: the float32 form (measured:
    const bool bf = a.abf_big != nullptr && mixct_use_bf(S, C, sync_call) && !raw_call;     // its registers leave the other no room)
    const size_t lds_bytes = mixct_lds_bytes(S, C, bf);
    if (cs || query_only) {
        const bool raw_sync = (cs && cs->ntx > 1) || (query_only && rv.ntx > 1);
        auto kern = raw_sync ? (bf ? k_rd_mixed_ct<S, C, NT, 3, mixct_abf_fits(S, C)> : k_rd_mixed_ct<S, C, NT, 3>)
                             : (bf ? k_rd_mixed_ct<S, C, NT, 2, mixct_abf_fits(S, C)> : k_rd_mixed_ct<S, C, NT, 2>);
        // (the attribute is per device: set it on every call, a process may drive several devices)
        if (lds_bytes > 64 * 1024)
            MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        static int per_cu_tab[4] = {0, 0, 0, 0};
        int &per_cu = per_cu_tab[(raw_sync ? 1 : 0) + (bf ? 2 : 0)];
        if (!per_cu) {
            int nb = 0;
            MMW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), NT, lds_bytes));
            per_cu = nb > 0 ? nb : 1;
        }
        int grid = sync_cus * per_cu;
        if (grid > planes) grid = planes;
        if (sync_grid) *sync_grid = grid;
        if (query_only) return MMW_OK;
        a.cs = *cs;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds_bytes, ctx->stream, a);
        return check_launch("rd_mixed_ct_sync");
    }
    // persistent + next-plane prefetch where nothing else would overlap the load phase
    static int persist_tab[2] = {-1, -1};       // persistent where only ONE one-plane workgroup would be resident per CU (LDS or registers)
    int &persist_dflt = persist_tab[bf ? 1 : 0];
    if (persist_dflt < 0) {
        int nb = 0;
        const void *k0 = bf ? reinterpret_cast<const void *>(k_rd_mixed_ct<S, C, NT, 0, mixct_abf_fits(S, C)>)
                            : reinterpret_cast<const void *>(k_rd_mixed_ct<S, C, NT, 0>);
        MMW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k0, NT, lds_bytes));
        persist_dflt = nb <= 1 ? 1 : 0;
    }
    const bool persist = rv.ntx <= 1 && persist_dflt != 0;
    auto kern = persist ? (bf ? k_rd_mixed_ct<S, C, NT, 1, mixct_abf_fits(S, C)> : k_rd_mixed_ct<S, C, NT, 1>)
                        : (bf ? k_rd_mixed_ct<S, C, NT, 0, mixct_abf_fits(S, C)> : k_rd_mixed_ct<S, C, NT, 0>);
    if (lds_bytes > 64 * 1024)
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    unsigned grid = rv.ntx > 1 ? (unsigned)raw_grid(planes, rv) : (unsigned)skip_planes(planes, rv);
    if (persist) {
        static int per_cu_tab2[2] = {0, 0};     // residency by LDS AND registers (per shape: this function is a template; per form)
        int &per_cu = per_cu_tab2[bf ? 1 : 0];
        if (!per_cu) {
            int nb = 0;
            MMW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), NT, lds_bytes));
            per_cu = nb > 0 ? nb : 1;
        }
        // (rd_leave_cus: CUs left to a pending tail of mmw_detect_points that runs beside this launch)
        const int cus = (ctx->active_cus > 0 ? ctx->active_cus : ctx->num_cu);
        const unsigned resident = (unsigned)((cus > 2 * ctx->rd_leave_cus ? cus - ctx->rd_leave_cus : cus) * per_cu);
        if (grid > resident) grid = resident;
    }
    if (tune_int("MMW_PHASE_CLOCKS", 0)) {
        // diagnostics: phase boundaries of workgroup 0 in shader clocks (marks 0..6 = start, load, range A, range B,
        // Doppler A, Doppler B, store; 8, 9 = inside the big-prime level), printed to stderr
        long long *d = nullptr, h[10] = {0};
        MMW_HIP(hipMalloc((void **)&d, sizeof(h)));
        MMW_HIP(hipMemsetAsync(d, 0, sizeof(h), ctx->stream));
        a.clk = d;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_bytes, ctx->stream, a);
        MMW_HIP(hipStreamSynchronize(ctx->stream));
        MMW_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        MMW_HIP(hipFree(d));
        std::fprintf(stderr, "rd_mixed_ct %dx%d NT=%d clocks:", S, C, NT);
        for (int i = 1; i < 7; ++i) std::fprintf(stderr, " %lld", h[i] - h[i - 1]);
        if (h[9]) std::fprintf(stderr, " | bigprime mfma %lld", h[9] - h[S1 == mixct::BIG_PRIME ? 1 : 3]);
        if (h[8]) std::fprintf(stderr, " (s / d pre-pass %lld)", h[8] - h[S1 == mixct::BIG_PRIME ? 1 : 3]);
        std::fprintf(stderr, "\n");
        return check_launch("rd_mixed_ct");
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_bytes, ctx->stream, a);
    return check_launch("rd_mixed_ct");
}

// one translation unit per part of the shape list (mmw_tu_mixed_ct_a/b/c.hip), so the library still builds in parallel
#if defined(MMW_TU_MIXED_CT_A) || defined(MMW_TU_MIXED_CT_B) || defined(MMW_TU_MIXED_CT_C)
#if defined(MMW_TU_MIXED_CT_A)
#define MMW_MIXED_CT_LIST MMW_MIXED_CT_SHAPES_A
#define MMW_MIXED_CT_FN launch_rd_mixed_ct_a
#elif defined(MMW_TU_MIXED_CT_B)
#define MMW_MIXED_CT_LIST MMW_MIXED_CT_SHAPES_B
#define MMW_MIXED_CT_FN launch_rd_mixed_ct_b
#else
#define MMW_MIXED_CT_LIST MMW_MIXED_CT_SHAPES_C
#define MMW_MIXED_CT_FN launch_rd_mixed_ct_c
#endif
int MMW_MIXED_CT_FN(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv,
                    const ChainSync *cs, int sync_cus, int *sync_grid, bool query_only, float *d_l1) {
#define X(s, c) if (S == s && C == c) return launch_rd_mixed_ct_sc<s, c>(ctx, d_in, in_plane_stride, d_out, planes, rv, cs, sync_cus, sync_grid, query_only, d_l1);
    MMW_MIXED_CT_LIST(X)
#undef X
    return MMW_ERR_UNSUPPORTED;
}
#endif

#ifdef MMW_TU_MIXED_CT_A
int launch_rd_mixed_ct_b(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv,
                         const ChainSync *cs, int sync_cus, int *sync_grid, bool query_only, float *d_l1);
int launch_rd_mixed_ct_c(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv,
                         const ChainSync *cs, int sync_cus, int *sync_grid, bool query_only, float *d_l1);
// MODE 3 (raw-cube producer of the synchronised chain): two of the 16384-cell power-of-two planes would spill there
// (32 carried registers of strided element loads beside a 32-point register FFT at 1024 threads); their raw chains
// keep the event schedule
bool rd_mixed_ct_raw_sync_supported(int S, int C) {
    if ((S == 128 && C == 128) || (S == 256 && C == 64)) return false;
    return rd_mixed_ct_supported(S, C);
}
bool rd_mixed_ct_supported(int S, int C) {
#define X(s, c) if (S == s && C == c) return true;
    MMW_MIXED_CT_SHAPES_A(X)
    MMW_MIXED_CT_SHAPES_B(X)
    MMW_MIXED_CT_SHAPES_C(X)
#undef X
    return false;
}
int launch_rd_mixed_ct(mmw_ctx *ctx, const void *d_in, long in_plane_stride, void *d_out, int planes, int S, int C, RawView rv,
                       const ChainSync *cs, int sync_cus, int *sync_grid, bool query_only, float *d_l1) {
    int rc = launch_rd_mixed_ct_a(ctx, d_in, in_plane_stride, d_out, planes, S, C, rv, cs, sync_cus, sync_grid, query_only, d_l1);
    if (rc == MMW_ERR_UNSUPPORTED) rc = launch_rd_mixed_ct_b(ctx, d_in, in_plane_stride, d_out, planes, S, C, rv, cs, sync_cus, sync_grid, query_only, d_l1);
    if (rc == MMW_ERR_UNSUPPORTED) rc = launch_rd_mixed_ct_c(ctx, d_in, in_plane_stride, d_out, planes, S, C, rv, cs, sync_cus, sync_grid, query_only, d_l1);
    if (rc == MMW_ERR_UNSUPPORTED) return set_error(MMW_ERR_UNSUPPORTED, "no compile-time mixed-radix kernel for %dx%d", S, C);
    return rc;
}
#endif

}  // namespace mmw
