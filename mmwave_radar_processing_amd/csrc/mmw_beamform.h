// Steering-matrix beamformers on the gfx950 matrix cores.
//
// bartlett(): the reference's delay-and-sum beamformer
//   (processors/simple_synthetic_array_beamformer_processor_multiFrame.py:499-585) is a Python double loop
//   over steering angles of "multiply [S,E] by a phase row, sum over E, window, FFT".  Restated as
//     Y = FFT_S( hann(S) . ( X[S,E] x W[E,T] ) ),   W[e,t] = hamming(E)[e] exp(j 2 pi d_t.p_e / lambda)
//   the contraction is one complex GEMM.  It runs on v_mfma_f32_32x32x2_f32 (exact f32 FMA chains, so the
//   1e-5 spectrum tolerance holds; bf16 MFMA would not): 4 real MFMAs per complex k-pair, operands staged
//   through LDS in 64x16 / 16x64 tiles, one 32x32 complex tile per wave.
// capon(): MVDR spectrum, float64 end to end on v_mfma_f64_16x16x4_f64 -- no upstream implementation exists
//   (SURVEY.md F2); definition in DESIGN.md / oracle_np.capon_spectrum.
#pragma once
#include "mmw_ctx.h"
#include "mmw_fft_generic.h"
#include "mmw_bf16x3.h"

namespace mmw {

typedef double v4d __attribute__((ext_vector_type(4)));

// LDS traffic inside one wave needs no hardware barrier (DS instructions of a wave execute in order); the compiler
// must not move accesses across the hand-over
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v where keep, +0 elsewhere, as a bit mask: a select lets the compiler sink the LOAD of v into a branch on the condition
// (and every load under a branch is followed by its own s_waitcnt)
__device__ __forceinline__ cplx<float> and_mask(cplx<float> v, bool keep) {
    const unsigned m = 0u - (unsigned)keep;
    const float re = v.x, im = v.y;         // (copies first: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0)
    return cplx<float>{__builtin_bit_cast(float, __builtin_bit_cast(unsigned, re) & m), __builtin_bit_cast(float, __builtin_bit_cast(unsigned, im) & m)};
}

// W[f][e][t] (complex64, row-major [E][Tp] per frame); phase reduced mod 1 turn in float64 before the sincos.
// P [F][3][E] element positions of each frame's (synthetic) array, dirs [3][T] steering directions.
__global__ __launch_bounds__(256) void k_steer(cplx<float> *W, const double *P, const double *dirs,
                                                const float *hamming, int E, int T, int Tp, double inv_lambda) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)E * Tp) return;
    const int t = (int)(gid % Tp), e = (int)(gid / Tp);
    const double *Pf = P + (long)blockIdx.y * 3 * E;
    cplx<float> w = cplx<float>{0.f, 0.f};
    if (t < T) {
        double turns = (dirs[t] * Pf[e] + dirs[T + t] * Pf[E + e] + dirs[2 * T + t] * Pf[2 * E + e]) * inv_lambda;
        turns -= rint(turns);
        double sn, cs;
        sincospi(2.0 * turns, &sn, &cs);
        w = cplx<float>{(float)(cs * hamming[e]), (float)(sn * hamming[e])};
    }
    W[(long)blockIdx.y * E * Tp + gid] = w;
}

// C[b][M][N] = A[b][M][K] x B[b][K][N], complex64, row-major, leading dimensions lda / ldb / ldc and batch strides
// sa / sb / sc (elements); blockIdx.z = b.
// Workgroup = 4 waves = 128 x 64 tile, each wave 64 x 32 (two 32 x 32 MFMA blocks, real and imaginary accumulators);
// K advances 16 at a time through a DOUBLE-BUFFERED planar LDS tile: the global loads of step k + 1 are issued before
// the MFMAs of step k and land in the other buffer afterwards -- one barrier per step.  v_mfma_f32_32x32x2_f32 is
// exact float32 (an fmaf chain in k order), so the 1e-5 spectrum tolerance holds; 4 real MFMAs per complex k pair.
constexpr int CG_TM = 128, CG_TN = 64, CG_TK = 16, CG_PA = CG_TM + 1, CG_PB = CG_TN + 1;
constexpr int CG_LDS_FLOATS = 2 * CG_TK * CG_PA + 2 * CG_TK * CG_PB;       // one buffer: Ar, Ai, Br, Bi
// ksplit > 1 (small batches: a 256 x 64 x 256 product is two workgroups stepping through K one latency at a time):
// blockIdx.z = batch * ksplit + kz, workgroup kz multiplies the K range [kz kc, (kz + 1) kc) and writes its partial
// product to Cm + kz * spart; k_sum_parts adds the partials in a fixed order.
__global__ __launch_bounds__(256) void k_cgemm_mfma(const cplx<float> *__restrict__ A, const cplx<float> *__restrict__ B,
                                                     cplx<float> *__restrict__ Cm, int M, int N, int K, int lda,
                                                     int ldb, int ldc, long sa, long sb, long sc, int ksplit, int kc, long spart) {
    __shared__ float lds[2 * CG_LDS_FLOATS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bz = blockIdx.z / ksplit, kz = blockIdx.z - bz * ksplit;
    A += (long)bz * sa;
    B += (long)bz * sb;
    Cm += (long)bz * sc + (long)kz * spart;
    const int k_begin = kz * kc;
    if (ksplit > 1) K = K < k_begin + kc ? K : k_begin + kc;
    const int m0 = blockIdx.y * CG_TM, n0 = blockIdx.x * CG_TN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 32;
    // global -> register staging: A rows (thread = row, 8 consecutive k), B rows (thread = k, 4 consecutive n)
    const int a_row = t >> 1, a_k = (t & 1) * 8, b_k = t >> 4, b_n = (t & 15) * 4;
    cplx<float> ra[8], rb[4];
    auto fetch = [&](int k0) {
        // unconditional, clamped loads + a select afterwards: a load under a condition is followed by its own s_waitcnt, and the
        // twelve loads of a step then cost twelve trips to memory one after the other (this was 5 k of a step's 7 k clocks)
        const int gm = m0 + a_row, gmc = gm < M ? gm : M - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gk = k0 + a_k + j;
            const cplx<float> v = A[(long)gmc * lda + (gk < K ? gk : K - 1)];
            ra[j] = and_mask(v, (gm < M) & (gk < K));
        }
        const int gk = k0 + b_k, gkc = gk < K ? gk : K - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + b_n + j;
            const cplx<float> v = B[(long)gkc * ldb + (gn < N ? gn : N - 1)];
            rb[j] = and_mask(v, (gk < K) & (gn < N));
        }
    };
    auto stash = [&](int buf) {
        float *Ar = lds + buf * CG_LDS_FLOATS, *Ai = Ar + CG_TK * CG_PA, *Br = Ai + CG_TK * CG_PA, *Bi = Br + CG_TK * CG_PB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            Ar[(a_k + j) * CG_PA + a_row] = ra[j].x;
            Ai[(a_k + j) * CG_PA + a_row] = ra[j].y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Br[b_k * CG_PB + b_n + j] = rb[j].x;
            Bi[b_k * CG_PB + b_n + j] = rb[j].y;
        }
    };
    v16f acc_r[2] = {{0}, {0}}, acc_i[2] = {{0}, {0}};
    fetch(k_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < K; k0 += CG_TK, buf ^= 1) {
        const bool more = k0 + CG_TK < K;
        if (more) fetch(k0 + CG_TK);                    // in flight while this step's MFMAs run
        const float *Ar = lds + buf * CG_LDS_FLOATS, *Ai = Ar + CG_TK * CG_PA, *Br = Ai + CG_TK * CG_PA, *Bi = Br + CG_TK * CG_PB;
#pragma unroll
        for (int ks = 0; ks < CG_TK; ks += 2) {
            // 32x32x2 operand maps: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
            const int kk = ks + (lane >> 5), ij = lane & 31;
            const float br = Br[kk * CG_PB + wn + ij], bi = Bi[kk * CG_PB + wn + ij];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const float ar = Ar[kk * CG_PA + wm + 32 * mb + ij], ai = Ai[kk * CG_PA + wm + 32 * mb + ij];
                acc_r[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, acc_r[mb], 0, 0, 0);
                acc_r[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, acc_r[mb], 0, 0, 0);
                acc_i[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, acc_i[mb], 0, 0, 0);
                acc_i[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, acc_i[mb], 0, 0, 0);
            }
        }
        if (more) stash(buf ^ 1);                       // the other buffer: nobody reads it during this step
        __syncthreads();
    }
    // C/D map of the 32x32 shapes: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = n0 + wn + (lane & 31);
            if (row < M && col < N) Cm[(long)row * ldc + col] = cplx<float>{acc_r[mb][r], acc_i[mb][r]};
        }
}

// The same product on bf16 x 3 (see split_bf16x3 / mfma_bf16x3): float32 MFMAs never co-execute with vector instructions on
// gfx950 and run at 1 / 12.8 of the bf16 rate, so the tile's operands are split EXACTLY into three bfloat16 pieces each while
// they are staged (vector work, under the matrix work of the previous step) and six of the nine partial products are formed
// by v_mfma_f32_32x32x16_bf16: 48 MFMAs of 32 cycles per wave and K step of 16 instead of 64 of 64 cycles; the error per
// product is 3 * 2^-24, the size of a float32 rounding (the 1e-5 tolerance of the spectra holds: tests).
// LDS per buffer: A [3 pieces][re, im][128 rows][16 k] and B [3][re, im][64 columns][16 k] bfloat16, 32 bytes per row; the two
// 16-byte halves of a row (k 0..7 / 8..15) swap places in rows 8..15 of every 16 (cb_slot): the 16-byte fragment reads of 16
// consecutive rows then fall into 16 different 16-byte slots.  72 KB double-buffered: two workgroups per CU.
constexpr int CB_PITCH = 16;                                                // bfloat16 per row
__device__ __forceinline__ int cb_slot(int row, int half) { return 8 * (half ^ ((row >> 3) & 1)); }
constexpr int CB_A = 3 * 2 * CG_TM * CB_PITCH, CB_B = 3 * 2 * CG_TN * CB_PITCH;   // bfloat16 per buffer
constexpr size_t CB_LDS_BYTES = 2 * (size_t)(CB_A + CB_B) * 2;
__device__ __forceinline__ void split_bf16x3_4(const float (&v)[4], unsigned (&p1)[2], unsigned (&p2)[2], unsigned (&p3)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        const unsigned a1 = __builtin_bit_cast(unsigned, a) & 0xffff0000u, b1 = __builtin_bit_cast(unsigned, b) & 0xffff0000u;
        const float ra = a - __builtin_bit_cast(float, a1), rb = b - __builtin_bit_cast(float, b1);
        const unsigned a2 = __builtin_bit_cast(unsigned, ra) & 0xffff0000u, b2 = __builtin_bit_cast(unsigned, rb) & 0xffff0000u;
        const float sa = ra - __builtin_bit_cast(float, a2), sb = rb - __builtin_bit_cast(float, b2);
        p1[q] = (a1 >> 16) | b1;
        p2[q] = (a2 >> 16) | b2;
        p3[q] = (__builtin_bit_cast(unsigned, sa) >> 16) | (__builtin_bit_cast(unsigned, sb) & 0xffff0000u);
    }
}
__global__ __launch_bounds__(256, 2) void k_cgemm_bf16x3(const cplx<float> *__restrict__ A, const cplx<float> *__restrict__ B,
                                                       cplx<float> *__restrict__ Cm, int M, int N, int K, int lda,
                                                       int ldb, int ldc, long sa, long sb, long sc, int ksplit, int kc, long spart) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short *lds = reinterpret_cast<unsigned short *>(smem);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int bz = blockIdx.z / ksplit, kz = blockIdx.z - bz * ksplit;
    A += (long)bz * sa;
    B += (long)bz * sb;
    Cm += (long)bz * sc + (long)kz * spart;
    const int k_begin = kz * kc;
    if (ksplit > 1) K = K < k_begin + kc ? K : k_begin + kc;
    const int m0 = blockIdx.y * CG_TM, n0 = blockIdx.x * CG_TN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 32;
    // global -> registers: A rows (thread = row, 8 consecutive k), B columns (thread = column, 4 consecutive k)
    const int a_row = t >> 1, a_k = (t & 1) * 8, b_n = t & 63, b_k = (t >> 6) * 4;
    cplx<float> ra[8], rb[4];
    auto fetch = [&](int k0) {
        // unconditional, clamped loads + a select afterwards: a load under a condition is followed by its own s_waitcnt, and the
        // twelve loads of a step then cost twelve trips to memory one after the other (this was 5 k of a step's 7 k clocks)
        const int gm = m0 + a_row, gmc = gm < M ? gm : M - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gk = k0 + a_k + j;
            ra[j] = A[(long)gmc * lda + (gk < K ? gk : K - 1)];           // (masked in stash(): the AND would wait for the load here)
        }
        const int gn = n0 + b_n, gnc = gn < N ? gn : N - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gk = k0 + b_k + j;
            rb[j] = B[(long)(gk < K ? gk : K - 1) * ldb + gnc];
        }
    };
    // piece p, component c (0 re, 1 im): A at ((p * 2 + c) * CG_TM + row) * CB_PITCH + k, B behind all of A
    auto stash = [&](int buf, int k0) {                 // k0: the step the registers were fetched for
        unsigned short *Ab = lds + buf * (CB_A + CB_B), *Bb = Ab + CB_A;
        float xr[8], xi[8];
        const bool row_ok = m0 + a_row < M, col_ok = n0 + b_n < N;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const cplx<float> v = and_mask(ra[j], row_ok & (k0 + a_k + j < K));
            xr[j] = v.x;
            xi[j] = v.y;
        }
        bf16x8 f[3];
        split_bf16x3(xr, f[0], f[1], f[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8 *>(Ab + ((p * 2 + 0) * CG_TM + a_row) * CB_PITCH + cb_slot(a_row, a_k >> 3)) = f[p];
        split_bf16x3(xi, f[0], f[1], f[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8 *>(Ab + ((p * 2 + 1) * CG_TM + a_row) * CB_PITCH + cb_slot(a_row, a_k >> 3)) = f[p];
        float yr[4], yi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const cplx<float> v = and_mask(rb[j], col_ok & (k0 + b_k + j < K));
            yr[j] = v.x;
            yi[j] = v.y;
        }
        unsigned q1[2], q2[2], q3[2];
        split_bf16x3_4(yr, q1, q2, q3);
        typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2v *>(Bb + ((0 * 2 + 0) * CG_TN + b_n) * CB_PITCH + cb_slot(b_n, b_k >> 3) + (b_k & 7)) = u32x2v{q1[0], q1[1]};
        *reinterpret_cast<u32x2v *>(Bb + ((1 * 2 + 0) * CG_TN + b_n) * CB_PITCH + cb_slot(b_n, b_k >> 3) + (b_k & 7)) = u32x2v{q2[0], q2[1]};
        *reinterpret_cast<u32x2v *>(Bb + ((2 * 2 + 0) * CG_TN + b_n) * CB_PITCH + cb_slot(b_n, b_k >> 3) + (b_k & 7)) = u32x2v{q3[0], q3[1]};
        split_bf16x3_4(yi, q1, q2, q3);
        *reinterpret_cast<u32x2v *>(Bb + ((0 * 2 + 1) * CG_TN + b_n) * CB_PITCH + cb_slot(b_n, b_k >> 3) + (b_k & 7)) = u32x2v{q1[0], q1[1]};
        *reinterpret_cast<u32x2v *>(Bb + ((1 * 2 + 1) * CG_TN + b_n) * CB_PITCH + cb_slot(b_n, b_k >> 3) + (b_k & 7)) = u32x2v{q2[0], q2[1]};
        *reinterpret_cast<u32x2v *>(Bb + ((2 * 2 + 1) * CG_TN + b_n) * CB_PITCH + cb_slot(b_n, b_k >> 3) + (b_k & 7)) = u32x2v{q3[0], q3[1]};
    };
    v16f acc_r[2] = {{0}, {0}}, acc_r2[2] = {{0}, {0}}, acc_i[2] = {{0}, {0}};
    fetch(k_begin);
    stash(0, k_begin);
    if (k_begin + CG_TK < K) fetch(k_begin + CG_TK);
    __syncthreads();
    int buf = 0;
    const int ij = lane & 31;
    // Software pipeline, two steps deep: the registers hold step k0 + 16 (fetched during the previous iteration, long arrived),
    // so splitting + staging them has no wait in front of it and can be scheduled INTO the MFMA block of step k0 (bf16 MFMAs
    // leave 24 of their 32 cycles to vector instructions); the loads of step k0 + 32 go out behind it.
    for (int k0 = k_begin; k0 < K; k0 += CG_TK, buf ^= 1) {
        const unsigned short *Ab = lds + buf * (CB_A + CB_B), *Bb = Ab + CB_A;
        // operand maps of v_mfma_f32_32x32x16_bf16: lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8 h + j], B[k = 8 h + j][col r]
        bf16x8 br3[3], bi3[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            br3[p] = *reinterpret_cast<const bf16x8 *>(Bb + ((p * 2 + 0) * CG_TN + wn + ij) * CB_PITCH + cb_slot(wn + ij, lane >> 5));
            bi3[p] = *reinterpret_cast<const bf16x8 *>(Bb + ((p * 2 + 1) * CG_TN + wn + ij) * CB_PITCH + cb_slot(wn + ij, lane >> 5));
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            bf16x8 ar3[3], ai3[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                ar3[p] = *reinterpret_cast<const bf16x8 *>(Ab + ((p * 2 + 0) * CG_TM + wm + 32 * mb + ij) * CB_PITCH + cb_slot(wm + 32 * mb + ij, lane >> 5));
                ai3[p] = *reinterpret_cast<const bf16x8 *>(Ab + ((p * 2 + 1) * CG_TM + wm + 32 * mb + ij) * CB_PITCH + cb_slot(wm + 32 * mb + ij, lane >> 5));
            }
            acc_r[mb] = mfma_bf16x3(ar3, br3, acc_r[mb]);
            acc_r2[mb] = mfma_bf16x3(ai3, bi3, acc_r2[mb]);
            acc_i[mb] = mfma_bf16x3(ar3, bi3, acc_i[mb]);
            acc_i[mb] = mfma_bf16x3(ai3, br3, acc_i[mb]);
        }
        // (unconditional -- after the last step it stages stale registers into the buffer nobody reads -- so that it shares the
        //  MFMAs' basic block and the scheduler may interleave: one MFMA, four vector instructions, ...)
        stash(buf ^ 1, k0 + CG_TK);
#pragma unroll
        for (int g = 0; g < 48; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        if (k0 + 2 * CG_TK < K) fetch(k0 + 2 * CG_TK);
        __syncthreads();
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = n0 + wn + (lane & 31);
            if (row < M && col < N) Cm[(long)row * ldc + col] = cplx<float>{acc_r[mb][r] - acc_r2[mb][r], acc_i[mb][r]};
        }
}

__global__ __launch_bounds__(256) void k_sum_parts(const cplx<float> *__restrict__ parts, cplx<float> *__restrict__ out, long n,
                                                    int ksplit) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    cplx<float> acc = parts[i];
    for (int k = 1; k < ksplit; ++k) acc = acc + parts[(long)k * n + i];
    out[i] = acc;
}

// ------------------------------------------------------------------ small contractions: steering fused, no LDS operand tiles
// At the reference's own sizes (S = 256 samples, E = 256 synthetic elements, 60-64 steering directions, a handful of
// frames) the tiled kernel above is a few workgroups stepping through K one load latency at a time, behind a separate
// k_steer launch.  Here ONE 32 x 32 output tile per workgroup, its NW waves splitting K in chunks of 64:
//   * A operand straight from global memory into registers: the MFMA sums over k, so WHICH k a lane feeds to a step is
//     free -- half-wave h takes k0 + 32 h + s in step s and every lane reads 256 contiguous bytes of its own row;
//   * B operand never exists in memory: lane (j, h) evaluates W[k0 + 32 h + s][n0 + j] itself (element positions of
//     the chunk in a wave-private LDS strip, phase reduced mod one turn in float64 as in k_steer, then a float32
//     sine / cosine: 2e-7 of |W|), interleaved with the MFMAs of the previous step;
//   * the NW partial tiles meet in LDS; no k_sum_parts pass, no W round trip.
// Workgroup ids are decoded so that the column tiles of one row tile run on the same XCD back to back (ids are dealt
// round-robin over the 8 XCDs): the second reader of an X row tile finds it in that XCD's L2.
__device__ __forceinline__ void sincos_turns_f32(float r, float &sn, float &cs) {     // r in [-0.5, 0.5] turns
    const float q = rintf(4.f * r);
    const float x = fmaf(q, -0.25f, r) * 6.2831853071795865f;                          // [-pi/4, pi/4], reduction exact
    const float x2 = x * x;
    float sp = fmaf(x2, 2.7557319e-6f, -1.9841270e-4f);
    sp = fmaf(sp, x2, 8.3333333e-3f);
    sp = fmaf(sp, x2, -1.6666667e-1f);
    const float s = fmaf(x * x2, sp, x);
    float cp = fmaf(x2, 2.4801587e-5f, -1.3888889e-3f);
    cp = fmaf(cp, x2, 4.1666667e-2f);
    cp = fmaf(cp, x2, -0.5f);
    const float c = fmaf(x2, cp, 1.f);
    const int qi = (int)q & 3;                  // quarter turns: 0: (c, s)  1: (-s, c)  2: (-c, -s)  3: (s, -c)
    const float a = (qi & 1) ? s : c, b = (qi & 1) ? c : s;
    cs = (qi == 1 || qi == 2) ? -a : a;
    sn = (qi >= 2) ? -b : b;
}

constexpr int BT_STRIP = 3 * 64 * 8 + 64 * 4;          // wave-private LDS: element positions + taper of one chunk
// NW waves per workgroup, NS MFMA steps per chunk (a chunk = 2 NS values of k: half-wave h feeds k0 + NS h + s to step s).
// FAST: E a multiple of 2 NS -- every chunk whole, 16-byte row loads (no second load path to merge).
// VAR 1: v_sin_f32 / v_cos_f32 (input in turns) instead of the polynomial: the same end-to-end error (2.1e-7 against
// 1.7e-7 of the peak at 256 x 256 x 60) for a third of the instructions.  That matters because float32 MFMAs and float32
// vector instructions do NOT overlap on this chip (mmw_diag_mfma_peak kinds 2-5: eight independent v_fma_f32 behind every
// v_mfma_f32_32x32x2 take the rate from 150 to 99 TF with two waves per SIMD, to 74 TF with one -- the matrix and the
// vector float32 rates are the same 256 flop/clk/CU, evidently the same multipliers): every vector instruction in this
// loop is paid in full, so the steering is trimmed to the minimum (float64 dot product + fract, one convert, two
// transcendentals, two multiplies per element) and eight waves per workgroup (two per SIMD) only hide latencies.
template <int NW, int NS, bool FAST, int VAR>
__global__ __launch_bounds__(64 * NW) void k_bartlett_tile(const cplx<float> *__restrict__ X, const double *__restrict__ P,
                                                            const double *__restrict__ dirs, const float *__restrict__ hamming,
                                                            cplx<float> *__restrict__ Cm, int S, int E, int T, int tiles_s, int NT,
                                                            int MT, double inv_lambda, long long *clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = 2 * NS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, ij = lane & 31;
    const int slot = blockIdx.x >> 3, st = slot / NT, nt = slot - st * NT;
    const int mt = st * 8 + (blockIdx.x & 7);
    if (mt >= MT) return;                               // padding of the row-tile count to the 8 XCDs (whole workgroup)
    auto mark = [&](int i) {        // diagnostics (MMW_PHASE_CLOCKS=1): workgroup 0, wave 0
        if (clk && blockIdx.x == 0 && threadIdx.x == 0) clk[i] = (long long)__builtin_amdgcn_s_memtime();
    };
    mark(0);
    const int fi = mt / tiles_s;
    const long f = fi;
    const int m0 = (mt - fi * tiles_s) * 32, n0 = nt * 32;
    double *Ps = reinterpret_cast<double *>(smem + wave * BT_STRIP);
    float *hs = reinterpret_cast<float *>(smem + wave * BT_STRIP + 3 * 64 * 8);
    float *red = reinterpret_cast<float *>(smem + NW * BT_STRIP);                     // [NW][2][16][64]
    const int tc = n0 + ij < T ? n0 + ij : T - 1;
    const double dx = dirs[tc] * inv_lambda, dy = dirs[T + tc] * inv_lambda, dz = dirs[2 * T + tc] * inv_lambda;
    const int gm = m0 + ij < S ? m0 + ij : S - 1;
    const cplx<float> *xrow = X + (f * S + gm) * (long)E;
    const double *Pf = P + f * 3 * E;
    // four accumulators: no MFMA waits for the result of the one issued just before it (summed after the loop)
    v16f acc_r = {0}, acc_i = {0}, acc_r2 = {0}, acc_i2 = {0};
    const int n_chunks = (E + CH - 1) / CH;
    for (int q = wave; q < n_chunks; q += NW) {
        const int k0 = q * CH, kb = k0 + NS * h;
        // element positions first: their loads retire before the (later issued) row loads, so the phase reduction below
        // overlaps the rows' flight
        const int e_st = k0 + lane < E ? k0 + lane : E - 1;
        const bool st_on = lane < CH;
        const double p0 = Pf[e_st], p1 = Pf[E + e_st], p2 = Pf[2 * E + e_st];
        const float hv = (st_on && k0 + lane < E) ? hamming[e_st] : 0.f;
        f32x4 ra[NS / 2];
        if (FAST) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(xrow + kb);
#pragma unroll
            for (int j = 0; j < NS / 2; ++j) ra[j] = src[j];
        } else {                                        // last chunk / odd row pitch: clamped 8-byte loads, taper 0 beyond E
#pragma unroll
            for (int j = 0; j < NS / 2; ++j) {
                const int ka = kb + 2 * j < E ? kb + 2 * j : E - 1, kc = kb + 2 * j + 1 < E ? kb + 2 * j + 1 : E - 1;
                const cplx<float> a = xrow[ka], c = xrow[kc];
                ra[j] = f32x4{a.x, a.y, c.x, c.y};
            }
        }
        mark(1);
        if (st_on) {
            Ps[lane] = p0;
            Ps[64 + lane] = p1;
            Ps[128 + lane] = p2;
            hs[lane] = hv;
        }
        wave_lds_sync();
        // Reduced phases (float64 up to the reduction mod one turn) and tapers of the chunk's steps first -- this runs
        // while the X rows are still in flight; the sine / cosine of step s + 1 sits between the MFMAs of step s.
        float rt[NS], hm[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ke = NS * h + s;
            const double turns = dx * Ps[ke] + dy * Ps[64 + ke] + dz * Ps[128 + ke];
            rt[s] = (float)(VAR & 1 ? __builtin_amdgcn_fract(turns) : turns - rint(turns));      // [0, 1) or [-0.5, 0.5] turns
            hm[s] = hs[ke];
        }
        auto steer = [&](int s, float &br, float &bi) {
            float sn, cs;
            if (VAR & 1) {
                sn = __builtin_amdgcn_sinf(rt[s]);
                cs = __builtin_amdgcn_cosf(rt[s]);
            } else {
                sincos_turns_f32(rt[s], sn, cs);
            }
            br = cs * hm[s];
            bi = sn * hm[s];
        };
        mark(2);
        if constexpr (VAR & 2) {
            // bf16 x 3: float32 MFMAs and vector instructions never co-execute on this chip (profiles/r04_coexec.json), bf16 MFMAs do,
            // and v_mfma_f32_32x32x16_bf16 moves 8 x the k of the float32 instruction in half the cycles: the operands are split
            // exactly into three bfloat16 pieces each (vector work that now runs UNDER the matrix work) and six of the nine
            // partial products are kept -- the error per product is 3 * 2^-24, the size of a float32 rounding.
            static_assert(NS == 16, "two blocks of 8 k per half-wave");
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float arv[8], aiv[8], brv[8], biv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int s_ = 8 * b + j;
                    arv[j] = (s_ & 1) ? ra[s_ >> 1].z : ra[s_ >> 1].x;
                    aiv[j] = (s_ & 1) ? ra[s_ >> 1].w : ra[s_ >> 1].y;
                    steer(s_, brv[j], biv[j]);
                }
                bf16x8 ar3[3], ai3[3], br3[3], bi3[3];
                split_bf16x3(arv, ar3[0], ar3[1], ar3[2]);
                split_bf16x3(aiv, ai3[0], ai3[1], ai3[2]);
                split_bf16x3(brv, br3[0], br3[1], br3[2]);
                split_bf16x3(biv, bi3[0], bi3[1], bi3[2]);
                acc_r = mfma_bf16x3(ar3, br3, acc_r);
                acc_i = mfma_bf16x3(ar3, bi3, acc_i);
                acc_r2 = mfma_bf16x3(ai3, bi3, acc_r2);         // (subtracted after the loop)
                acc_i2 = mfma_bf16x3(ai3, br3, acc_i2);
            }
        } else {
        float br, bi;
        steer(0, br, bi);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float nbr = 0.f, nbi = 0.f;
            if (s + 1 < NS) steer(s + 1, nbr, nbi);
            const float ar = (s & 1) ? ra[s >> 1].z : ra[s >> 1].x, ai = (s & 1) ? ra[s >> 1].w : ra[s >> 1].y;
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, acc_r, 0, 0, 0);
            acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, acc_i, 0, 0, 0);
            acc_r2 = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, acc_r2, 0, 0, 0);
            acc_i2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, acc_i2, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, (VAR & 1) ? 3 : 8, 0);      // vector-ALU work of the next element
            }
            br = nbr;
            bi = nbi;
        }
        }
        mark(3);
        wave_lds_sync();                                // the strip is rewritten by the next chunk
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[((wave * 2 + 0) * 16 + r) * 64 + lane] = (VAR & 2) ? acc_r[r] - acc_r2[r] : acc_r[r] + acc_r2[r];
        red[((wave * 2 + 1) * 16 + r) * 64 + lane] = acc_i[r] + acc_i2[r];
    }
    __syncthreads();
    // C/D map of the 32x32 shapes: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); partial tiles added in wave order
    for (int r = wave; r < 16; r += NW) {
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            sr += red[((w * 2 + 0) * 16 + r) * 64 + lane];
            si += red[((w * 2 + 1) * 16 + r) * 64 + lane];
        }
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + ij;
        if (row < S && col < T) Cm[(f * S + row) * (long)T + col] = cplx<float>{sr, si};
    }
    mark(4);
}

// ONE frame at the reference's size (256 x 256 x 60-64) is sixteen 32 x 32 tiles: sixteen workgroups, each a single latency
// chain (positions -> rows -> phases -> 48 MFMAs + 16 sines / cosines per lane -> LDS reduction) on a chip of 256 CUs.  The
// same kernel on 16 x 16 tiles (v_mfma_f32_16x16x32_bf16, the bfloat16 x 3 form only) spreads the frame over 64 workgroups
// whose waves steer 8 values of k each (group g = lane >> 4 feeds k0 + 8 g + j) and issue 24 MFMAs of 16 cycles: the chain
// per workgroup is half as long.  Used while the 32 x 32 tiling would leave three quarters of the chip without a workgroup.
template <int NW, bool FAST>
__global__ __launch_bounds__(64 * NW) void k_bartlett_tile16(const cplx<float> *__restrict__ X, const double *__restrict__ P,
                                                              const double *__restrict__ dirs, const float *__restrict__ hamming,
                                                              cplx<float> *__restrict__ Cm, int S, int E, int T, int tiles_s, int NT,
                                                              int MT, double inv_lambda) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CH = 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i16 = lane & 15;
    const int slot = blockIdx.x >> 3, st = slot / NT, nt = slot - st * NT;
    const int mt = st * 8 + (blockIdx.x & 7);
    if (mt >= MT) return;                               // padding of the row-tile count to the 8 XCDs (whole workgroup)
    const int fi = mt / tiles_s;
    const long f = fi;
    const int m0 = (mt - fi * tiles_s) * 16, n0 = nt * 16;
    double *Ps = reinterpret_cast<double *>(smem + wave * BT_STRIP);
    float *hs = reinterpret_cast<float *>(smem + wave * BT_STRIP + 3 * 64 * 8);
    float *red = reinterpret_cast<float *>(smem + NW * BT_STRIP);                     // [NW][2][4][64]
    const int tc = n0 + i16 < T ? n0 + i16 : T - 1;
    const double dx = dirs[tc] * inv_lambda, dy = dirs[T + tc] * inv_lambda, dz = dirs[2 * T + tc] * inv_lambda;
    const int gm = m0 + i16 < S ? m0 + i16 : S - 1;
    const cplx<float> *xrow = X + (f * S + gm) * (long)E;
    const double *Pf = P + f * 3 * E;
    v4f acc_r = {0}, acc_i = {0}, acc_r2 = {0}, acc_i2 = {0};
    const int n_chunks = (E + CH - 1) / CH;
    for (int q = wave; q < n_chunks; q += NW) {
        const int k0 = q * CH, kb = k0 + 8 * g;
        const int e_st = k0 + lane < E ? k0 + lane : E - 1;
        const bool st_on = lane < CH;
        const double p0 = Pf[e_st], p1 = Pf[E + e_st], p2 = Pf[2 * E + e_st];
        const float hv = (st_on && k0 + lane < E) ? hamming[e_st] : 0.f;
        f32x4 ra[4];
        if (FAST) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(xrow + kb);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = src[j];
        } else {                                        // last chunk / odd row pitch: clamped 8-byte loads, taper 0 beyond E
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ka = kb + 2 * j < E ? kb + 2 * j : E - 1, kc = kb + 2 * j + 1 < E ? kb + 2 * j + 1 : E - 1;
                const cplx<float> a = xrow[ka], c = xrow[kc];
                ra[j] = f32x4{a.x, a.y, c.x, c.y};
            }
        }
        if (st_on) {
            Ps[lane] = p0;
            Ps[64 + lane] = p1;
            Ps[128 + lane] = p2;
            hs[lane] = hv;
        }
        wave_lds_sync();
        float arv[8], aiv[8], brv[8], biv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ke = 8 * g + j;
            const double turns = dx * Ps[ke] + dy * Ps[64 + ke] + dz * Ps[128 + ke];
            const float rt = (float)__builtin_amdgcn_fract(turns), hm = hs[ke];
            brv[j] = __builtin_amdgcn_cosf(rt) * hm;
            biv[j] = __builtin_amdgcn_sinf(rt) * hm;
            arv[j] = (j & 1) ? ra[j >> 1].z : ra[j >> 1].x;
            aiv[j] = (j & 1) ? ra[j >> 1].w : ra[j >> 1].y;
        }
        bf16x8 ar3[3], ai3[3], br3[3], bi3[3];
        split_bf16x3(arv, ar3[0], ar3[1], ar3[2]);
        split_bf16x3(aiv, ai3[0], ai3[1], ai3[2]);
        split_bf16x3(brv, br3[0], br3[1], br3[2]);
        split_bf16x3(biv, bi3[0], bi3[1], bi3[2]);
        acc_r = mfma16_bf16x3(ar3, br3, acc_r);
        acc_i = mfma16_bf16x3(ar3, bi3, acc_i);
        acc_r2 = mfma16_bf16x3(ai3, bi3, acc_r2);         // (subtracted below)
        acc_i2 = mfma16_bf16x3(ai3, br3, acc_i2);
        wave_lds_sync();                                // the strip is rewritten by the next chunk
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[((wave * 2 + 0) * 4 + r) * 64 + lane] = acc_r[r] - acc_r2[r];
        red[((wave * 2 + 1) * 4 + r) * 64 + lane] = acc_i[r] + acc_i2[r];
    }
    __syncthreads();
    // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg; partial tiles added in wave order
    for (int r = wave; r < 4; r += NW) {
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            sr += red[((w * 2 + 0) * 4 + r) * 64 + lane];
            si += red[((w * 2 + 1) * 4 + r) * 64 + lane];
        }
        const int row = m0 + 4 * g + r, col = n0 + i16;
        if (row < S && col < T) Cm[(f * S + row) * (long)T + col] = cplx<float>{sr, si};
    }
}

// d_X [F][S][E] c64, d_P [F][3][E] f64, d_dirs [3][T] f64 -> d_out [F][S][T] c64
inline int bartlett(mmw_ctx *ctx, const void *d_X, const double *d_P, const double *d_dirs, void *d_out, int n_frames, int S,
                    int E, int T, double lambda_m) {
    const int Tp = (T + 3) & ~3;
    const size_t w_bytes = (size_t)n_frames * E * Tp * sizeof(cplx<float>), c_bytes = (size_t)n_frames * S * T * sizeof(cplx<float>);
    // small batches: split K over workgroups until about half the chip is busy (each part at least two K steps)
    const long wgs = (long)((T + CG_TN - 1) / CG_TN) * ((S + CG_TM - 1) / CG_TM) * n_frames;
    int ksplit = 1;
    if (wgs * 2 <= ctx->num_cu && E >= 4 * CG_TK)
        ksplit = (int)std::min<long>(std::min<long>(16, E / (2 * CG_TK)), ctx->num_cu / (2 * wgs));
    if (ksplit < 2) ksplit = 1;
    const int kc = ksplit > 1 ? ((E + ksplit - 1) / ksplit + CG_TK - 1) / CG_TK * CG_TK : E;
    if (ksplit > 1) ksplit = (E + kc - 1) / kc;         // no empty part
    MMW_TRY(ensure_scratch(ctx, w_bytes + c_bytes * (ksplit > 1 ? ksplit + 1 : 1)));
    cplx<float> *W = (cplx<float> *)ctx->scratch;
    cplx<float> *Cm = (cplx<float> *)((char *)ctx->scratch + w_bytes);
    cplx<float> *Cparts = Cm + (size_t)n_frames * S * T;
    const void *ham, *hann;
    MMW_TRY(get_table<float>(ctx, TAB_HAMMING, E, &ham));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &hann));
    ProfScope ps(ctx, "bartlett");
    // small contractions (at most four 32 x 32 tiles per CU): steering evaluated inside the tile kernel
    const int tiles_s = (S + 31) / 32, NT = (T + 31) / 32;
    const long MT = (long)n_frames * tiles_s;     // (< 2^31: n_frames <= 65535)
    const int path = opt_int(ctx, "MMW_BARTLETT_PATH", 0);      // 1: tile kernel, 2: tiled GEMM (experiments / tests)
    if (path == 1 || (path == 0 && MT * NT <= 4L * ctx->num_cu)) {
        ProfScope pg(ctx, "cgemm");
        const int nw = 8;
        const unsigned grid = (unsigned)(8 * NT * ((MT + 7) / 8));
        const size_t lds = (size_t)nw * BT_STRIP + (size_t)nw * 2 * 16 * 64 * 4;
        const bool poly = opt_int(ctx, "MMW_BARTLETT_POLY", 0) != 0;       // polynomial sine / cosine instead of v_sin / v_cos
        // MMW_BARTLETT_BF16 (default 1): the contraction on bf16 x 3 MFMAs; 0: float32 MFMAs
        const bool bf3 = opt_int(ctx, "MMW_BARTLETT_BF16", 1) != 0 && !poly;
        // a frame or two (less than a 32 x 32 tile per four CUs): 16 x 16 tiles (MMW_BARTLETT_TILE16=0: never, 1: always)
        const int t16 = opt_int(ctx, "MMW_BARTLETT_TILE16", -1);
        if (bf3 && !tune_int("MMW_PHASE_CLOCKS", 0) && (t16 == 1 || (t16 != 0 && MT * NT * 4 <= ctx->num_cu))) {
            const int tiles16 = (S + 15) / 16, NT16 = (T + 15) / 16;
            const long MT16 = (long)n_frames * tiles16;
            const unsigned grid16 = (unsigned)(8 * NT16 * ((MT16 + 7) / 8));
            const size_t lds16 = (size_t)nw * BT_STRIP + (size_t)nw * 2 * 4 * 64 * 4;
            auto k16 = (E & 31) == 0 ? k_bartlett_tile16<8, true> : k_bartlett_tile16<8, false>;
            hipLaunchKernelGGL(k16, dim3(grid16), dim3(64 * nw), lds16, ctx->stream, (const cplx<float> *)d_X, d_P, d_dirs,
                               (const float *)ham, Cm, S, E, T, tiles16, NT16, (int)MT16, 1.0 / lambda_m);
            MMW_TRY(check_launch("bartlett_tile16"));
        } else {
        auto kern = bf3 ? k_bartlett_tile<8, 16, false, 3> : k_bartlett_tile<8, 16, false, 1>;
        if ((E & 31) == 0) kern = poly ? k_bartlett_tile<8, 16, true, 0> : bf3 ? k_bartlett_tile<8, 16, true, 3> : k_bartlett_tile<8, 16, true, 1>;
        else if (poly) kern = k_bartlett_tile<8, 16, false, 0>;
        long long *d_clk = nullptr;
        if (tune_int("MMW_PHASE_CLOCKS", 0)) {
            MMW_HIP(hipMalloc((void **)&d_clk, 5 * sizeof(long long)));
            MMW_HIP(hipMemsetAsync(d_clk, 0, 5 * sizeof(long long), ctx->stream));
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, ctx->stream, (const cplx<float> *)d_X, d_P, d_dirs,
                           (const float *)ham, Cm, S, E, T, tiles_s, NT, (int)MT, 1.0 / lambda_m, d_clk);
        if (d_clk) {
            long long h[5] = {0};
            MMW_HIP(hipStreamSynchronize(ctx->stream));
            MMW_HIP(hipMemcpy(h, d_clk, sizeof(h), hipMemcpyDeviceToHost));
            MMW_HIP(hipFree(d_clk));
            std::fprintf(stderr, "bartlett tile clocks (workgroup 0 wave 0, last chunk): loads issued %lld, phase reduction %lld, "
                                 "MFMA loop %lld, reduce + store %lld\n",
                         h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3]);
        }
        MMW_TRY(check_launch("bartlett_tile"));
        }
    } else {
        const long nW = (long)E * Tp;
        hipLaunchKernelGGL(k_steer, dim3((unsigned)((nW + 255) / 256), (unsigned)n_frames), dim3(256), 0, ctx->stream, W, d_P, d_dirs,
                           (const float *)ham, E, T, Tp, 1.0 / lambda_m);
        MMW_TRY(check_launch("steer"));
        {
            ProfScope pg(ctx, "cgemm");
            dim3 grid((T + CG_TN - 1) / CG_TN, (S + CG_TM - 1) / CG_TM, (unsigned)(n_frames * ksplit));
            const long n_c = (long)n_frames * S * T;
            if (opt_int(ctx, "MMW_BARTLETT_BF16", 1)) {
                MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cgemm_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)CB_LDS_BYTES));
                hipLaunchKernelGGL(k_cgemm_bf16x3, grid, dim3(256), CB_LDS_BYTES, ctx->stream, (const cplx<float> *)d_X, W,
                                   ksplit > 1 ? Cparts : Cm, S, T, E, E, Tp, T, (long)S * E, (long)E * Tp, (long)S * T, ksplit, kc, n_c);
            } else
                hipLaunchKernelGGL(k_cgemm_mfma, grid, dim3(256), 0, ctx->stream, (const cplx<float> *)d_X, W, ksplit > 1 ? Cparts : Cm, S, T, E,
                                   E, Tp, T, (long)S * E, (long)E * Tp, (long)S * T, ksplit, kc, n_c);
            MMW_TRY(check_launch("cgemm_mfma"));
            if (ksplit > 1) {
                hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((n_c + 255) / 256)), dim3(256), 0, ctx->stream, Cparts, Cm, n_c, ksplit);
                MMW_TRY(check_launch("sum_parts"));
            }
        }
    }
    // hann(S) window and FFT along S for every steering column of every frame (:537-540)
    FftArgs a{};
    a.in = Cm;
    a.out = d_out;
    a.outer = n_frames;
    a.inner = T;
    a.n_in = S;
    a.in_outer_stride = a.out_outer_stride = (long)S * T;
    a.in_axis_stride = a.out_axis_stride = T;
    a.in_inner_stride = a.out_inner_stride = 1;
    a.win_axis = hann;
    a.scale = 1.0;
    return launch_fft_axis<float, float>(ctx, a, S, false);
}

// ------------------------------------------------------------------ matrix-core peak probe (diagnostics)
// Back-to-back MFMAs on four independent accumulators per wave, operands in registers: the rate the matrix pipe itself
// sustains on this device, used as the `peak` the beamformer kernels are priced against (the guide lists 157.3 TF for
// f32 MFMA and no figure for f64).  kind 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f64_16x16x4_f64.
__global__ __launch_bounds__(256) void k_diag_mfma(float *sink, int iters, int kind) {
    const float seed = (float)(threadIdx.x & 7) * 0.25f + 0.5f;
    if (kind == 0) {
        v16f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.5f, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.25f, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.125f, a3, 0, 0, 0);
        }
        const float v = a0[0] + a1[1] + a2[2] + a3[3];
        if (v == 12345.678f) sink[0] = v;
    } else if (kind >= 6) {
        // kind 6 / 7: the contrast case for the co-execution question -- v_mfma_f32_32x32x16_bf16 (a matrix instruction the
        // guide says has its own pipe) alone / with 8 independent float32 FMAs after every MFMA
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        v16f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        bf16x8 x, y;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = (__bf16)seed;
            y[j] = (__bf16)0.5f;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = seed + (float)j;
        const float m = 0.999f, c = 0.001f;
        auto valu = [&]() {
            if (kind == 7) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
            }
        };
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
        }
        float t = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) t += v[j];
        if (t == 12345.678f) sink[0] = t;
    } else if (kind >= 2) {
        // kind 2 / 3: the same MFMA stream with 8 / 16 independent float32 FMAs after every MFMA -- do a wave's (or the
        // SIMD's other wave's) vector instructions run under an MFMA in flight?  (rate reported counts the MFMAs only)
        v16f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = seed + (float)j;
        const float m = 0.999f, c = 0.001f;
        auto valu = [&]() {
#pragma unroll
            for (int rep = 0; rep < (kind == 3 ? 2 : 1); ++rep)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
        };
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.5f, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.25f, a2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 0.125f, a3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            valu();
            __builtin_amdgcn_sched_barrier(0);
        }
        float t = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) t += v[j];
        if (t == 12345.678f) sink[0] = t;
    } else {
        v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        const double sd = seed;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sd, 1.0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sd, 0.5, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(sd, 0.25, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(sd, 0.125, a3, 0, 0, 0);
        }
        const double v = a0[0] + a1[1] + a2[2] + a3[3];
        if (v == 12345.678) sink[0] = (float)v;
    }
}

// ------------------------------------------------------------------ Capon / MVDR (float64)
// X [F][V][R][K] complex64 (K snapshots per range bin), angle table Zt [T] complex128 = a_1(theta_t), out [F][R][T] float32:
//   P = 1 / Re( a^H (Rxx + delta tr(Rxx)/V I)^-1 a ),  Rxx = X_r X_r^H / K,  a_v = exp(-j pi v sin(theta)).
// Earlier forms, for the record: round 1 ran one 64-thread workgroup per bin with a serial Cholesky (160 ns per bin);
// round 2 one wave per bin with an LDS snapshot tile, a wave-synchronous Cholesky and a forward substitution per angle
// (V (V + 1) / 2 complex multiply-adds per angle: ~10 k vector-ALU clocks per bin next to the 8.2 k clocks its 128 MFMAs
// occupy the matrix pipe -- 6.7 us per frame at 32 x 512 bins, matrix pipe 31 % busy).  This form:
// For a uniform line array a_v = z^v, so a^H Rxx^-1 a = g_0 + 2 Re sum_{d >= 1} g_d conj(z)^d with g_d the sum of the
// d-th lower diagonal of Rxx^-1: V - 1 complex multiply-adds per steering angle instead of the V (V + 1) / 2 of a
// forward substitution, once the inverse is there.  Per (frame, range bin), one wave:
//   1. snapshots straight from global memory into MFMA operand registers (the covariance sums over snapshots, so WHICH
//      snapshot a lane feeds to a step is free: lane (antenna i, slot s) takes 16-byte pairs 8 j + 2 s of each 32-snapshot
//      chunk -- every load instruction reads whole 64-byte runs); the next chunk (or the next bin's first) is in flight
//      during the MFMAs; no LDS tile;
//   2. Rxx on v_mfma_f64_16x16x4_f64 as before; the C/D registers ARE the matrix layout of step 3 (column = lane & 15,
//      rows (lane >> 4) + 4 q);
//   3. diagonal loading, then V symmetric sweeps (Gauss-Jordan on a Hermitian positive definite matrix, no pivoting):
//      the pivot column goes through a 16-entry LDS strip (A_kc = conj(A_ck)), everything else stays in registers;
//      one reciprocal per sweep, no square roots; the result is -Rxx^-1;
//   4. the V diagonal sums by V lanes out of an LDS copy, then Horner in conj(z) for every angle of the lane.
// About 5 k vector-ALU clocks per bin next to the 8.2 k clocks its 128 MFMAs occupy the matrix pipe, four waves per SIMD
// so that one wave's sweeps run under another's MFMAs (the Cholesky / substitution form above needed ~10 k).
constexpr int CAPON2_WAVE_LDS = 16 * 16 + 16 * 17 * 16 + 16 * 16;     // pivot column + matrix copy + diagonal sums
constexpr int CAPON2_UTAB = 192;                                     // conj(z) of the first 192 angles, per workgroup

// Snapshot chunks (32 per chunk) alternate between two register slots; the loads of chunk c + 2 (or of the next bin's
// chunk of the same slot) are issued right after the MFMAs that consumed the slot, so two chunks = 64 MFMAs of look-ahead
// are always in flight -- with one chunk the 2 k clocks of 32 MFMAs did not cover an HBM round trip under load.
template <int NQ, bool ALIGNED>     // antennas V <= 4 NQ: register rows per lane; ALIGNED: K a multiple of 32 (no chunk tail)
__global__ __launch_bounds__(256, 4) void k_capon_sweep(const cplx<float> *__restrict__ X, const cplx<double> *__restrict__ Zt,
                                                        float *__restrict__ out, int V, int R, int K, int T, int n_bins,
                                                        double delta, long long *clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, li = l & 15, lk = l >> 4;
    cplx<double> *utab = reinterpret_cast<cplx<double> *>(smem);                           // [CAPON2_UTAB]
    char *base = smem + CAPON2_UTAB * 16 + (size_t)w * CAPON2_WAVE_LDS;
    cplx<double> *colbuf = reinterpret_cast<cplx<double> *>(base);                         // [16]
    cplx<double>(*Mx)[17] = reinterpret_cast<cplx<double>(*)[17]>(base + 256);
    cplx<double> *gbuf = reinterpret_cast<cplx<double> *>(base + 256 + 16 * 17 * 16);      // [16]
    for (int t = threadIdx.x; t < CAPON2_UTAB; t += 256) {
        const cplx<double> z = t < T ? Zt[t] : cplx<double>{1.0, 0.0};
        utab[t] = cplx<double>{z.x, -z.y};
    }
    __syncthreads();
    constexpr int ZR = CAPON2_UTAB / 64;
    const int n_chunks = (K + 31) >> 5;
    // lanes of the padding rows (li >= V) read antenna 0 again: their products only reach covariance entries with a row or
    // a column >= V, which nothing below reads
    const long lane_off = (long)(li < V ? li : 0) * R * K + 2 * lk;
    f32x4 slot0[4], slot1[4];
    auto fetch = [&](f32x4 *dst, const cplx<float> *xrow, int chunk) {        // xrow: antenna row of the lane, + 2 lk
        if (ALIGNED) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const f32x4 *>(xrow + 32 * chunk + 8 * j);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 32 * chunk + 8 * j + 2 * lk, ka = k < K ? k : K - 1, kb = k + 1 < K ? k + 1 : K - 1;
                const cplx<float> a = xrow[ka - 2 * lk], b = xrow[kb - 2 * lk];
                dst[j] = f32x4{k < K ? a.x : 0.f, k < K ? a.y : 0.f, k + 1 < K ? b.x : 0.f, k + 1 < K ? b.y : 0.f};
            }
        }
    };
    v4d cr, ci;
    auto mfma_chunk = [&](const f32x4 *cur) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const double xr = e ? cur[j].z : cur[j].x, xi = e ? cur[j].w : cur[j].y;
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, xr, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xr, ci, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xi, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr, xi, ci, 0, 0, 0);
            }
        }
    };
    // (frame, range bin) of this wave's bins, advanced without divisions: bin -> bin + 4 gridDim.x
    const int first_bin = blockIdx.x * 4 + w, bin_step = gridDim.x * 4;
    const int step_f = bin_step / R, step_r = bin_step - step_f * R;
    int nf = first_bin / R, nr = first_bin - nf * R;        // of the NEXT bin to be fetched
    auto row_of = [&](int f, int r) { return X + ((long)f * V * R + r) * (long)K + lane_off; };
    const cplx<float> *nrow = row_of(nf, nr);
    if (first_bin < n_bins) {
        fetch(slot0, nrow, 0);
        if (n_chunks > 1) fetch(slot1, nrow, 1);
    }
    for (int bin = first_bin; bin < n_bins; bin += bin_step) {
        auto mark = [&](int i) {        // diagnostics (MMW_PHASE_CLOCKS=1): phase boundaries of workgroup 0, wave 0
            if (clk && blockIdx.x == 0 && threadIdx.x == 0) clk[i] = (long long)__builtin_amdgcn_s_memtime();
        };
        mark(0);
        // ---- 1 + 2: covariance
        cr = v4d{0, 0, 0, 0};
        ci = v4d{0, 0, 0, 0};
        const cplx<float> *crow = nrow;
        const int nbin = bin + bin_step;
        nf += step_f;
        nr += step_r;
        if (nr >= R) {
            nr -= R;
            ++nf;
        }
        const bool more = nbin < n_bins;
        nrow = row_of(more ? nf : 0, more ? nr : 0);
        for (int c = 0; c < n_chunks; c += 2) {
            mfma_chunk(slot0);
            if (c + 2 < n_chunks) fetch(slot0, crow, c + 2);
            else if (more) fetch(slot0, nrow, 0);
            if (c + 1 < n_chunks) {
                mfma_chunk(slot1);
                if (c + 3 < n_chunks) fetch(slot1, crow, c + 3);
                else if (more && n_chunks > 1) fetch(slot1, nrow, 1);
            }
        }
        mark(1);
        // ---- 3: f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg.  Diagonal loading, sweeps.
        const double invK = 1.0 / (double)K;
        cplx<double> a[NQ];
        double tr = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            a[q] = cplx<double>{cr[q] * invK, ci[q] * invK};
            if (lk + 4 * q == li && li < V) tr += a[q].x;
        }
        for (int d = 32; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);
        const double load = delta * tr / (double)V;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (lk + 4 * q == li && li < V) a[q] = cplx<double>{a[q].x + load, 0.0};
        // sweep k = 4 kq + kk: the pivot row lives in register kq of the lanes with lk == kk
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) {
            for (int kk = 0; kk < 4 && 4 * kq + kk < V; ++kk) {
                const int k = 4 * kq + kk;
                if (li == k) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) colbuf[lk + 4 * q] = a[q];
                }
                wave_lds_sync();
                const double d = colbuf[k].x;
                const cplx<double> pc = colbuf[li];
                cplx<double> t[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) t[q] = colbuf[lk + 4 * q];
                double inv = __builtin_amdgcn_rcp(d);
                inv = fma(fma(-d, inv, 1.0), inv, inv);
                inv = fma(fma(-d, inv, 1.0), inv, inv);
                // everywhere: A_ic -= (A_ik / d) conj(A_ck); then the pivot column becomes A_ik / d, the pivot row
                // conj(A_ck) / d and the pivot itself -1 / d   (symmetric sweep: the matrix ends as -Rxx^-1)
                const bool ck = li == k;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    t[q] = cplx<double>{t[q].x * inv, t[q].y * inv};
                    const double nx = fma(-t[q].y, pc.y, fma(-t[q].x, pc.x, a[q].x));
                    const double ny = fma(-t[q].y, pc.x, fma(t[q].x, pc.y, a[q].y));
                    a[q] = ck ? t[q] : cplx<double>{nx, ny};
                }
                if (lk == kk) a[kq] = ck ? cplx<double>{-inv, 0.0} : cplx<double>{pc.x * inv, -pc.y * inv};
                wave_lds_sync();            // the strip is rewritten by the next sweep
            }
        }
        mark(2);
        // ---- 4: diagonal sums of Rxx^-1 = -a, then P = 1 / (g_0 + 2 Re sum_d g_d conj(z)^d)
#pragma unroll
        for (int q = 0; q < NQ; ++q) Mx[lk + 4 * q][li] = a[q];
        wave_lds_sync();
        if (l < V) {
            cplx<double> g = cplx<double>{0.0, 0.0};
            for (int qd = 0; qd + l < V; ++qd) g = g - Mx[qd + l][qd];
            gbuf[l] = g;
        }
        wave_lds_sync();
        auto spectrum = [&](double re_sum, double g0) {
            const double den = g0 + 2.0 * re_sum;
            double r = __builtin_amdgcn_rcp(den);
            r = fma(fma(-den, r, 1.0), r, r);
            return (float)fma(fma(-den, r, 1.0), r, r);
        };
        const double g0 = gbuf[0].x;
        // Horner: s = u (g_1 + u (g_2 + ... + u g_{V-1})), only Re(s) of the last product is needed
        cplx<double> u[ZR], acc[ZR];
#pragma unroll
        for (int q = 0; q < ZR; ++q) {
            u[q] = utab[l + 64 * q];
            acc[q] = V > 1 ? gbuf[V - 1] : cplx<double>{0.0, 0.0};
        }
        for (int d = V - 2; d >= 1; --d) {
            const cplx<double> g = gbuf[d];
#pragma unroll
            for (int q = 0; q < ZR; ++q) {
                const double nx = fma(-acc[q].y, u[q].y, fma(acc[q].x, u[q].x, g.x));
                acc[q].y = fma(acc[q].y, u[q].x, fma(acc[q].x, u[q].y, g.y));
                acc[q].x = nx;
            }
        }
#pragma unroll
        for (int q = 0; q < ZR; ++q)
            if (l + 64 * q < T) out[(long)bin * T + l + 64 * q] = spectrum(fma(-acc[q].y, u[q].y, acc[q].x * u[q].x), g0);
        for (int tt = l + 64 * ZR; tt < T; tt += 64) {
            const cplx<double> z = Zt[tt], uu = cplx<double>{z.x, -z.y};
            cplx<double> s = V > 1 ? gbuf[V - 1] : cplx<double>{0.0, 0.0};
            for (int d = V - 2; d >= 1; --d) s = cmul(s, uu) + gbuf[d];
            out[(long)bin * T + tt] = spectrum(s.x * uu.x - s.y * uu.y, g0);
        }
        mark(3);
        wave_lds_sync();
    }
}

inline int capon(mmw_ctx *ctx, const void *d_X, const double *h_thetas, float *d_out, int n_frames, int V, int R, int K,
                 int T, double delta) {
    // z_t = a_1(theta_t) = exp(-j pi sin(theta_t)); the table stays on the device while the caller keeps its angle grid
    // (an upload per call was a host synchronisation per call)
    if (ctx->capon_key.size() != (size_t)T || std::memcmp(ctx->capon_key.data(), h_thetas, (size_t)T * sizeof(double)) != 0) {
        std::vector<double> zt((size_t)T * 2);
        for (int t = 0; t < T; ++t) {
            const double ph = -M_PI * std::sin(h_thetas[t]);
            zt[2 * t] = std::cos(ph);
            zt[2 * t + 1] = std::sin(ph);
        }
        MMW_HIP(hipStreamSynchronize(ctx->stream));                     // a launch in flight may still read the old table
        if (ctx->capon_z) (void)hipFree(ctx->capon_z);
        ctx->capon_z = nullptr;
        ctx->capon_key.clear();
        MMW_HIP(hipMalloc(&ctx->capon_z, zt.size() * sizeof(double)));
        MMW_HIP(hipMemcpy(ctx->capon_z, zt.data(), zt.size() * sizeof(double), hipMemcpyHostToDevice));
        ctx->capon_key.assign(h_thetas, h_thetas + T);
    }
    ProfScope ps(ctx, "capon");
    const int n_bins = n_frames * R;
    // sixteen waves per CU (four per SIMD), each wave walking its share of the bins
    const long want = ((long)n_bins + 3) / 4;
    const int grid = (int)std::min<long>(want, (long)ctx->num_cu * 4);
    const int lds = CAPON2_UTAB * 16 + 4 * CAPON2_WAVE_LDS;
    if ((long)n_frames * R >= (1L << 31)) return set_error(MMW_ERR_INVALID, "capon: more than 2^31 range bins in one call");
    auto kern = (K & 31) == 0 ? (V <= 4 ? k_capon_sweep<1, true> : V <= 8 ? k_capon_sweep<2, true> : V <= 12 ? k_capon_sweep<3, true>
                                                                                             : k_capon_sweep<4, true>)
                              : (V <= 4 ? k_capon_sweep<1, false> : V <= 8 ? k_capon_sweep<2, false> : V <= 12 ? k_capon_sweep<3, false>
                                                                                              : k_capon_sweep<4, false>);
    if (tune_int("MMW_PHASE_CLOCKS", 0)) {
        long long *d = nullptr, h[4] = {0};
        MMW_HIP(hipMalloc((void **)&d, sizeof(h)));
        MMW_HIP(hipMemsetAsync(d, 0, sizeof(h), ctx->stream));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, (const cplx<float> *)d_X,
                           (const cplx<double> *)ctx->capon_z, d_out, V, R, K, T, n_bins, delta, d);
        MMW_HIP(hipStreamSynchronize(ctx->stream));
        MMW_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        MMW_HIP(hipFree(d));
        std::fprintf(stderr, "capon clocks (last bin of workgroup 0 wave 0): covariance %lld, sweeps %lld, spectrum %lld\n",
                     h[1] - h[0], h[2] - h[1], h[3] - h[2]);
        return check_launch("capon");
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, (const cplx<float> *)d_X,
                       (const cplx<double> *)ctx->capon_z, d_out, V, R, K, T, n_bins, delta, (long long *)nullptr);
    return check_launch("capon");
}

}  // namespace mmw
