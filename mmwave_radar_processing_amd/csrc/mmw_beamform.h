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

namespace mmw {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

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
        const int gm = m0 + a_row;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gk = k0 + a_k + j;
            ra[j] = (gm < M && gk < K) ? A[(long)gm * lda + gk] : cplx<float>{0.f, 0.f};
        }
        const int gk = k0 + b_k;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + b_n + j;
            rb[j] = (gk < K && gn < N) ? B[(long)gk * ldb + gn] : cplx<float>{0.f, 0.f};
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

__global__ __launch_bounds__(256) void k_sum_parts(const cplx<float> *__restrict__ parts, cplx<float> *__restrict__ out, long n,
                                                    int ksplit) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    cplx<float> acc = parts[i];
    for (int k = 1; k < ksplit; ++k) acc = acc + parts[(long)k * n + i];
    out[i] = acc;
}

// d_X [F][S][E] c64, d_P [F][3][E] f64, d_dirs [3][T] f64 -> d_out [F][S][T] c64
inline int bartlett(mmw_ctx *ctx, const void *d_X, const double *d_P, const double *d_dirs, void *d_out, int n_frames, int S,
                    int E, int T, double lambda_m) {
    const int Tp = (T + 3) & ~3;
    const size_t w_bytes = (size_t)n_frames * E * Tp * sizeof(cplx<float>), c_bytes = (size_t)n_frames * S * T * sizeof(cplx<float>);
    // small batches: split K over workgroups until about half the chip is busy (each part at least two K steps)
    const long wgs = (long)((T + CG_TN - 1) / CG_TN) * ((S + CG_TM - 1) / CG_TM) * n_frames;
    int ksplit = 1;
    if (wgs * 2 <= ctx->num_cu && E >= 4 * CG_TK && !tune_int("MMW_CGEMM_NO_KSPLIT", 0))
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
    const long nW = (long)E * Tp;
    hipLaunchKernelGGL(k_steer, dim3((unsigned)((nW + 255) / 256), (unsigned)n_frames), dim3(256), 0, ctx->stream, W, d_P, d_dirs,
                       (const float *)ham, E, T, Tp, 1.0 / lambda_m);
    MMW_TRY(check_launch("steer"));
    {
        ProfScope pg(ctx, "cgemm");
        dim3 grid((T + CG_TN - 1) / CG_TN, (S + CG_TM - 1) / CG_TM, (unsigned)(n_frames * ksplit));
        const long n_c = (long)n_frames * S * T;
        hipLaunchKernelGGL(k_cgemm_mfma, grid, dim3(256), 0, ctx->stream, (const cplx<float> *)d_X, W, ksplit > 1 ? Cparts : Cm, S, T, E,
                           E, Tp, T, (long)S * E, (long)E * Tp, (long)S * T, ksplit, kc, n_c);
        MMW_TRY(check_launch("cgemm_mfma"));
        if (ksplit > 1) {
            hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)((n_c + 255) / 256)), dim3(256), 0, ctx->stream, Cparts, Cm, n_c, ksplit);
            MMW_TRY(check_launch("sum_parts"));
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
// X [F][V][R][K] complex64 (K snapshots per range bin), steering table Ast [V][T] complex128, out [F][R][T] float32:
//   P = 1 / Re( a^H (Rxx + delta tr(Rxx)/V I)^-1 a ),  Rxx = X_r X_r^H / K,  a_v = exp(-j pi v sin(theta)).
// One WAVE per (frame, range bin), four waves per workgroup walking the bins of a batch (the single-frame kernel of
// round 1 -- one 64-thread workgroup per bin, serial Cholesky / inverse on one to sixteen lanes, a workgroup barrier
// per step -- was latency bound at 160 ns per bin).  Per bin:
//   1. the V snapshot rows (K x 8 B, contiguous) arrive by coalesced 16-B loads into a wave-private LDS tile
//      (pitch KT + 2: the MFMA operand reads that follow are bank-conflict free);
//   2. Rxx on v_mfma_f64_16x16x4_f64: 4 real MFMAs per 4 snapshots (Re = xr xr^T + xi xi^T, Im = xi xr^T - xr xi^T);
//   3. trace + diagonal loading by a 16-lane shuffle sum; right-looking Cholesky Rxx = L L^H in LDS with all 64 lanes on
//      the trailing update, V steps, wave-synchronous (no workgroup barriers: each wave owns its matrix);
//   4. a^H Rxx^-1 a = |L^-1 a|^2: every lane forward-substitutes L y = a for its steering angles with L broadcast from
//      LDS -- V (V + 1) / 2 complex MACs per angle instead of the 16 x 16 of an explicit inverse and a second GEMM.
constexpr int CAPON_KT = 64;                        // snapshots staged per pass
constexpr int CAPON_XP = CAPON_KT + 2;              // LDS row pitch of the snapshot tile (complex64 elements)
constexpr int CAPON_WAVE_LDS = 16 * CAPON_XP * 8 + 16 * 17 * 16 + 16 * 8;   // tile + matrix + 1 / diag

// LDS traffic inside one wave needs no hardware barrier (DS instructions of a wave execute in order); the compiler
// must not move accesses across the hand-over
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int HV>       // antennas V <= 2 HV: sizes the register arrays (prefetched rows, substitution vector)
__global__ __launch_bounds__(256, 3) void k_capon_batch(const cplx<float> *__restrict__ X, const cplx<double> *__restrict__ Ast,
                                                      float *__restrict__ out, int V, int R, int K, int T, long n_bins,
                                                      double delta, long long *clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    char *base = smem + (size_t)w * CAPON_WAVE_LDS;
    cplx<float> *Xs = reinterpret_cast<cplx<float> *>(base);                              // [16][CAPON_XP]
    cplx<double>(*Mx)[17] = reinterpret_cast<cplx<double>(*)[17]>(base + 16 * CAPON_XP * 8);
    double *inv_d = reinterpret_cast<double *>(base + 16 * CAPON_XP * 8 + 16 * 17 * 16);
    const int li = l & 15, lk = l >> 4;
    // rows >= V of the tile stay zero
    for (int e = l; e < 16 * CAPON_XP; e += 64) Xs[e] = cplx<float>{0.f, 0.f};
    wave_lds_sync();
    // Steering vectors by recurrence: a_v(theta) = z^v with z = a_1 = exp(-j pi sin(theta)).  A lane serves the same
    // angles t = l, l + 64, ... for every bin, so their z live in registers for the whole kernel; the table form read
    // a_i[t] from global memory inside the substitution loop -- twelve dependent L2 round trips per angle and bin, which
    // made phase 4 ~60 k of the ~67 k clocks a wave spent on a bin.  (11 complex products: ~3e-15 relative.)
    constexpr int ZR = 1;                       // angle rounds held in registers (T <= 256); further rounds reload z
    cplx<double> zreg[ZR];
#pragma unroll
    for (int q = 0; q < ZR; ++q) {
        const int t = l + 64 * q;
        zreg[q] = (t < T && V > 1) ? Ast[(long)T + t] : cplx<double>{1.0, 0.0};
    }
    // half a wave per antenna row: lane h = l & 31 brings snapshots k0 + 2h, 2h + 1 of rows (l >> 5), (l >> 5) + 2, ...
    f32x4 pre[HV];
    auto fetch_chunk = [&](const cplx<float> *xrow, int k0) {
#pragma unroll
        for (int i = 0; i < HV; ++i) {
            const int v = (l >> 5) + 2 * i, k = k0 + 2 * (l & 31);
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (v < V) {
                const cplx<float> *src = xrow + (long)v * R * K + k;
                if (k + 1 < K && ((K & 1) == 0)) q = *reinterpret_cast<const f32x4 *>(src);     // 16-B aligned when K is even
                else {
                    if (k < K) { q.x = src[0].x; q.y = src[0].y; }
                    if (k + 1 < K) { q.z = src[1].x; q.w = src[1].y; }
                }
            }
            pre[i] = q;
        }
    };
    auto stash_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < HV; ++i) {
            const int v = (l >> 5) + 2 * i;
            if (v < V) *reinterpret_cast<f32x4 *>(&Xs[v * CAPON_XP + 2 * (l & 31)]) = pre[i];
        }
    };
    const long first_bin = (long)blockIdx.x * 4 + w, bin_step = (long)gridDim.x * 4;
    for (long bin = first_bin; bin < n_bins; bin += bin_step) {
        const long f = bin / R;
        const int r = (int)(bin - f * R);
        const cplx<float> *xb = X + ((f * V) * R + r) * (long)K;        // antenna v at xb + v * R * K
        auto mark = [&](int i) {        // diagnostics (MMW_PHASE_CLOCKS=1): phase boundaries of workgroup 0, wave 0
            if (clk && blockIdx.x == 0 && threadIdx.x == 0) clk[i] = (long long)__builtin_amdgcn_s_memtime();
        };
        mark(0);
        // ---- 1 + 2: covariance.  The snapshot chunks are software pipelined: chunk c + 1 travels from global memory to
        //      registers while the MFMAs of chunk c run out of the LDS tile, and chunk 0 of the NEXT bin during this bin's
        //      Cholesky (below) -- the exposed load latency was ~2/3 of this phase's 18 k clocks.
        v4d cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        if (bin == first_bin) {
            fetch_chunk(xb, 0);
            stash_chunk();
            wave_lds_sync();
        }
        for (int k0 = 0; k0 < K; k0 += CAPON_KT) {
            const bool more = k0 + CAPON_KT < K;
            if (more) fetch_chunk(xb, k0 + CAPON_KT);
#pragma unroll 4
            for (int kk = 0; kk < CAPON_KT; kk += 4) {
                const cplx<float> x = Xs[li * CAPON_XP + kk + lk];       // operand maps: A[i = l&15][k = l>>4], B[k][j = l&15]
                const double xr = x.x, xi = x.y;
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, xr, cr, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xi, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xr, ci, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr, xi, ci, 0, 0, 0);
            }
            wave_lds_sync();
            if (more) {
                stash_chunk();
                wave_lds_sync();
            }
        }
        mark(1);
        // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
        const double invK = 1.0 / (double)K;
#pragma unroll
        for (int q = 0; q < 4; ++q) Mx[lk + 4 * q][li] = cplx<double>{cr[q] * invK, ci[q] * invK};
        wave_lds_sync();
        // next bin's first chunk: in flight during the Cholesky, into the (free) tile before the substitution
        const long nbin = bin + bin_step;
        if (nbin < n_bins) {
            const long nf = nbin / R;
            fetch_chunk(X + ((nf * V) * R + (nbin - nf * R)) * (long)K, 0);
        }
        // ---- 3: diagonal loading, Cholesky (lower triangle of Mx becomes L, diagonal real)
        double tr = (l < V) ? Mx[l][l].x : 0.0;
        for (int d = 8; d >= 1; d >>= 1) tr += __shfl_xor(tr, d, 64);      // lanes 0..15 hold the 16 diagonal terms
        tr = __shfl(tr, 0, 64);
        if (l < V) Mx[l][l] = cplx<double>{Mx[l][l].x + delta * tr / (double)V, 0.0};
        wave_lds_sync();
        for (int j = 0; j < V; ++j) {
            const double d = sqrt(Mx[j][j].x), inv = 1.0 / d;
            wave_lds_sync();
            if (l == j) {
                Mx[j][j] = cplx<double>{d, 0.0};
                inv_d[j] = inv;
            } else if (l > j && l < V) Mx[l][j] = Mx[l][j] * inv;
            wave_lds_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = l + 64 * q, i = idx >> 4, c = idx & 15;
                if (c > j && c <= i && i < V) {
                    const cplx<double> a = Mx[i][j], b = Mx[c][j];
                    Mx[i][c] = Mx[i][c] - cplx<double>{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};   // a * conj(b)
                }
            }
            wave_lds_sync();
        }
        if (nbin < n_bins) {
            stash_chunk();
            wave_lds_sync();
        }
        mark(2);
        // ---- 4: P(theta) = 1 / |L^-1 a(theta)|^2
        auto solve = [&](int t, cplx<double> z) {
            cplx<double> y[2 * HV];
            cplx<double> a = cplx<double>{1.0, 0.0};      // a_0
            double p = 0.0;
#pragma unroll
            for (int i = 0; i < 2 * HV; ++i) {
                if (i < V) {
                    cplx<double> s = a;
#pragma unroll
                    for (int k = 0; k < i; ++k) s = s - cmul(Mx[i][k], y[k]);      // Mx[i][k]: same address in every lane
                    y[i] = s * inv_d[i];
                    p += y[i].x * y[i].x + y[i].y * y[i].y;
                    a = cmul(a, z);
                }
            }
            out[bin * T + t] = (float)(1.0 / p);
        };
#pragma unroll
        for (int q = 0; q < ZR; ++q)
            if (l + 64 * q < T) solve(l + 64 * q, zreg[q]);
        for (int t = l + 64 * ZR; t < T; t += 64) solve(t, V > 1 ? Ast[(long)T + t] : cplx<double>{1.0, 0.0});
        mark(3);
        wave_lds_sync();
    }
}

inline int capon(mmw_ctx *ctx, const void *d_X, const double *h_thetas, float *d_out, int n_frames, int V, int R, int K,
                 int T, double delta) {
    std::vector<double> ast((size_t)V * T * 2, 0.0);
    for (int v = 0; v < V; ++v)
        for (int t = 0; t < T; ++t) {
            const double ph = -M_PI * (double)v * std::sin(h_thetas[t]);     // a_v = exp(-j pi v sin(theta))
            ast[((size_t)v * T + t) * 2] = std::cos(ph);
            ast[((size_t)v * T + t) * 2 + 1] = std::sin(ph);
        }
    MMW_TRY(ensure_scratch(ctx, ast.size() * sizeof(double)));
    MMW_HIP(hipMemcpyAsync(ctx->scratch, ast.data(), ast.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));   // ast is a host temporary
    ProfScope ps(ctx, "capon");
    const long n_bins = (long)n_frames * R;
    const int lds = 4 * CAPON_WAVE_LDS;
    // three workgroups (12 waves) fit a CU's LDS; a few per CU, each wave walking its share of the bins
    const long want = (n_bins + 3) / 4;
    const int grid = (int)std::min<long>(want, (long)ctx->num_cu * 3 * std::max(1, tune_int("MMW_CAPON_WG_ROUNDS", 2)));
    auto kern = V <= 4 ? k_capon_batch<2> : V <= 8 ? k_capon_batch<4> : V <= 12 ? k_capon_batch<6> : k_capon_batch<8>;
    if (tune_int("MMW_PHASE_CLOCKS", 0)) {
        long long *d = nullptr, h[4] = {0};
        MMW_HIP(hipMalloc((void **)&d, sizeof(h)));
        MMW_HIP(hipMemsetAsync(d, 0, sizeof(h), ctx->stream));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, (const cplx<float> *)d_X,
                           (const cplx<double> *)ctx->scratch, d_out, V, R, K, T, n_bins, delta, d);
        MMW_HIP(hipStreamSynchronize(ctx->stream));
        MMW_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        MMW_HIP(hipFree(d));
        std::fprintf(stderr, "capon clocks (last bin of workgroup 0 wave 0): covariance %lld, Cholesky %lld, substitution %lld\n",
                     h[1] - h[0], h[2] - h[1], h[3] - h[2]);
        return check_launch("capon");
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx->stream, (const cplx<float> *)d_X,
                       (const cplx<double> *)ctx->scratch, d_out, V, R, K, T, n_bins, delta, (long long *)nullptr);
    return check_launch("capon");
}

}  // namespace mmw
