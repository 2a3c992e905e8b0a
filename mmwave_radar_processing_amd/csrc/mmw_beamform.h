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

// W[e][t] (complex64, row-major [E][Tp]); phase reduced mod 1 turn in float64 before the sincos.
__global__ __launch_bounds__(256) void k_steer(cplx<float> *W, const double *P, const double *dirs,
                                                const float *hamming, int E, int T, int Tp, double inv_lambda) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)E * Tp) return;
    const int t = (int)(gid % Tp), e = (int)(gid / Tp);
    cplx<float> w = cplx<float>{0.f, 0.f};
    if (t < T) {
        double turns = (dirs[t] * P[e] + dirs[T + t] * P[E + e] + dirs[2 * T + t] * P[2 * E + e]) * inv_lambda;
        turns -= rint(turns);
        double sn, cs;
        sincospi(2.0 * turns, &sn, &cs);
        w = cplx<float>{(float)(cs * hamming[e]), (float)(sn * hamming[e])};
    }
    W[gid] = w;
}

// C[M][N] = A[M][K] x B[K][N], complex64, row-major, leading dimensions lda / ldb / ldc (elements).
// Workgroup = 4 waves = 64x64 tile (2x2 waves of 32x32); K advanced 16 at a time through LDS.
constexpr int CG_TM = 64, CG_TN = 64, CG_TK = 16;
__global__ __launch_bounds__(256) void k_cgemm_mfma(const cplx<float> *__restrict__ A, const cplx<float> *__restrict__ B,
                                                     cplx<float> *__restrict__ Cm, int M, int N, int K, int lda,
                                                     int ldb, int ldc) {
    // planar tiles so each MFMA operand is one conflict-free 4-byte LDS read
    __shared__ float sAr[CG_TK][CG_TM + 1], sAi[CG_TK][CG_TM + 1];
    __shared__ float sBr[CG_TK][CG_TN + 1], sBi[CG_TK][CG_TN + 1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.y * CG_TM, n0 = blockIdx.x * CG_TN;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    v16f acc_r = {0}, acc_i = {0};
    for (int k0 = 0; k0 < K; k0 += CG_TK) {
        // A tile: 64 rows x 16 k (1024 elements, 4 per thread), lanes along k for 128-B row segments
        for (int q = t; q < CG_TM * CG_TK; q += 256) {
            const int kk = q % CG_TK, mm = q / CG_TK;
            const int gm = m0 + mm, gk = k0 + kk;
            const cplx<float> v = (gm < M && gk < K) ? A[(long)gm * lda + gk] : cplx<float>{0.f, 0.f};
            sAr[kk][mm] = v.x;
            sAi[kk][mm] = v.y;
        }
        for (int q = t; q < CG_TK * CG_TN; q += 256) {
            const int nn = q % CG_TN, kk = q / CG_TN;
            const int gn = n0 + nn, gk = k0 + kk;
            const cplx<float> v = (gn < N && gk < K) ? B[(long)gk * ldb + gn] : cplx<float>{0.f, 0.f};
            sBr[kk][nn] = v.x;
            sBi[kk][nn] = v.y;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < CG_TK; ks += 2) {
            // 32x32x2 operand maps: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
            const int kk = ks + (lane >> 5), ij = lane & 31;
            const float ar = sAr[kk][wm + ij], ai = sAi[kk][wm + ij];
            const float br = sBr[kk][wn + ij], bi = sBi[kk][wn + ij];
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, acc_r, 0, 0, 0);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, acc_r, 0, 0, 0);
            acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, acc_i, 0, 0, 0);
            acc_i = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, acc_i, 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 shapes: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = n0 + wn + (lane & 31);
        if (row < M && col < N) Cm[(long)row * ldc + col] = cplx<float>{acc_r[r], acc_i[r]};
    }
}

inline int bartlett(mmw_ctx *ctx, const void *d_X, const double *d_P, const double *d_dirs, void *d_out, int S,
                    int E, int T, double lambda_m) {
    const int Tp = (T + 3) & ~3;
    const size_t w_bytes = (size_t)E * Tp * sizeof(cplx<float>), c_bytes = (size_t)S * T * sizeof(cplx<float>);
    MMW_TRY(ensure_scratch(ctx, w_bytes + c_bytes));
    cplx<float> *W = (cplx<float> *)ctx->scratch;
    cplx<float> *Cm = (cplx<float> *)((char *)ctx->scratch + w_bytes);
    const void *ham, *hann;
    MMW_TRY(get_table<float>(ctx, TAB_HAMMING, E, &ham));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &hann));
    ProfScope ps(ctx, "bartlett");
    const long nW = (long)E * Tp;
    hipLaunchKernelGGL(k_steer, dim3((unsigned)((nW + 255) / 256)), dim3(256), 0, ctx->stream, W, d_P, d_dirs,
                       (const float *)ham, E, T, Tp, 1.0 / lambda_m);
    MMW_TRY(check_launch("steer"));
    dim3 grid((T + CG_TN - 1) / CG_TN, (S + CG_TM - 1) / CG_TM);
    hipLaunchKernelGGL(k_cgemm_mfma, grid, dim3(256), 0, ctx->stream, (const cplx<float> *)d_X, W, Cm, S, T, E, E, Tp, T);
    MMW_TRY(check_launch("cgemm_mfma"));
    // hann(S) window and FFT along S for every steering column (:537-540)
    FftArgs a{};
    a.in = Cm;
    a.out = d_out;
    a.outer = 1;
    a.inner = T;
    a.n_in = S;
    a.in_axis_stride = a.out_axis_stride = T;
    a.in_inner_stride = a.out_inner_stride = 1;
    a.win_axis = hann;
    a.scale = 1.0;
    return launch_fft_axis<float, float>(ctx, a, S, false);
}

// ------------------------------------------------------------------ Capon / MVDR (float64)
// One wave per range bin.  X [V][R][K] complex64, steering table Ast [16][Tp] complex128 (rows >= V zero),
// out [R][T] float32:  P = 1 / Re( a^H (Rxx + delta tr(Rxx)/V I)^-1 a ),  Rxx = X_r X_r^H / K.
__global__ __launch_bounds__(64) void k_capon(const cplx<float> *__restrict__ X, const cplx<double> *__restrict__ Ast,
                                               float *__restrict__ out, int V, int R, int K, int T, int Tp,
                                               double delta) {
    __shared__ cplx<double> Mx[16][17], Li[16][17], Ri[16][17];
    const int l = threadIdx.x, r = blockIdx.x;
    const int li = l & 15, lk = l >> 4;
    // ---- covariance on the f64 matrix cores: operand maps A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]
    v4d cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int k = k0 + lk;
        double xr = 0.0, xi = 0.0;
        if (li < V && k < K) {
            const cplx<float> x = X[((long)li * R + r) * K + k];
            xr = x.x;
            xi = x.y;
        }
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, xr, cr, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xi, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, xr, ci, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr, xi, ci, 0, 0, 0);
    }
    // f64 C/D map: col = lane&15, row = (lane>>4) + 4*reg
    const double invK = 1.0 / (double)K;
#pragma unroll
    for (int q = 0; q < 4; ++q) Mx[lk + 4 * q][li] = cplx<double>{cr[q] * invK, ci[q] * invK};
    __syncthreads();
    if (l == 0) {
        double tr = 0.0;
        for (int i = 0; i < V; ++i) tr += Mx[i][i].x;
        const double load = delta * tr / (double)V;
        for (int i = 0; i < 16; ++i) Mx[i][i] = cplx<double>{i < V ? Mx[i][i].x + load : 1.0, 0.0};
    }
    __syncthreads();
    // ---- Cholesky Mx = L L^H (lower triangle in place)
    for (int j = 0; j < 16; ++j) {
        const double d = sqrt(Mx[j][j].x);
        __syncthreads();
        if (l == j) Mx[j][j] = cplx<double>{d, 0.0};
        if (l > j && l < 16) Mx[l][j] = Mx[l][j] * (1.0 / d);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = l + 64 * q, i = idx >> 4, c = idx & 15;
            if (c > j && c <= i) {
                const cplx<double> a = Mx[i][j], b = Mx[c][j];
                Mx[i][c] = Mx[i][c] - cplx<double>{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};   // a * conj(b)
            }
        }
        __syncthreads();
    }
    // ---- Li = L^-1 by forward substitution, one column per lane
    if (l < 16) {
        for (int i = 0; i < 16; ++i) {
            cplx<double> s = cplx<double>{i == l ? 1.0 : 0.0, 0.0};
            for (int k = l; k < i; ++k) s = s - cmul(Mx[i][k], Li[k][l]);
            Li[i][l] = (i < l) ? cplx<double>{0.0, 0.0} : s * (1.0 / Mx[i][i].x);
        }
    }
    __syncthreads();
    // ---- Ri = Li^H Li
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = l + 64 * q, i = idx >> 4, j = idx & 15;
        cplx<double> s = cplx<double>{0.0, 0.0};
        for (int k = (i > j ? i : j); k < 16; ++k) {
            const cplx<double> a = Li[k][i], b = Li[k][j];
            s = s + cplx<double>{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x};                      // conj(a) * b
        }
        Ri[i][j] = s;
    }
    __syncthreads();
    // ---- Z = Ri x A(theta) on the matrix cores, 16 angles per pass; P = 1 / Re(sum_v conj(a_v) Z_v)
    for (int t0 = 0; t0 < T; t0 += 16) {
        v4d zr = {0, 0, 0, 0}, zi = {0, 0, 0, 0};
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4) {
            const cplx<double> a = Ri[li][k0 + lk];
            const int tc = t0 + li;
            const cplx<double> b = tc < Tp ? Ast[(long)(k0 + lk) * Tp + tc] : cplx<double>{0.0, 0.0};
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, zr, 0, 0, 0);
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, b.y, zr, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.y, zi, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.x, zi, 0, 0, 0);
        }
        double acc = 0.0;
        const int tc = t0 + li;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = lk + 4 * q;
            const cplx<double> a = tc < Tp ? Ast[(long)row * Tp + tc] : cplx<double>{0.0, 0.0};
            acc += a.x * zr[q] + a.y * zi[q];
        }
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (l < 16 && tc < T) out[(long)r * T + tc] = (float)(1.0 / acc);
    }
}

inline int capon(mmw_ctx *ctx, const void *d_X, const double *h_thetas, float *d_out, int V, int R, int K, int T,
                 double delta) {
    const int Tp = (T + 15) & ~15;
    std::vector<double> ast((size_t)16 * Tp * 2, 0.0);
    for (int v = 0; v < V; ++v)
        for (int t = 0; t < T; ++t) {
            const double ph = -M_PI * (double)v * std::sin(h_thetas[t]);     // a_v = exp(-j pi v sin(theta))
            ast[((size_t)v * Tp + t) * 2] = std::cos(ph);
            ast[((size_t)v * Tp + t) * 2 + 1] = std::sin(ph);
        }
    MMW_TRY(ensure_scratch(ctx, ast.size() * sizeof(double)));
    MMW_HIP(hipMemcpyAsync(ctx->scratch, ast.data(), ast.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MMW_HIP(hipStreamSynchronize(ctx->stream));   // ast is a host temporary
    ProfScope ps(ctx, "capon");
    hipLaunchKernelGGL(k_capon, dim3(R), dim3(64), 0, ctx->stream, (const cplx<float> *)d_X,
                       (const cplx<double> *)ctx->scratch, d_out, V, R, K, T, Tp, delta);
    return check_launch("capon");
}

}  // namespace mmw
