// Float64 range-Doppler CELLS of whole frames at once: the dense form of the exact argmax's refinement.
//
//   PointCloudGenerator._compute_angle_estimation   processors/point_cloud_generator.py:143-214  (complex128 cells of a complex128
//   RangeDopplerProcessor.process                   processors/range_doppler_resp.py:49-110       range-Doppler cube)
//
// mmw_angle_argmax_exact proves its float32 argmax with a worst-case rounding bound and re-evaluates what the bound cannot
// decide from float64 cells.  For strong detections that is one evaluation in several hundred and k_argmax_refine_* computes
// each needed cell as a direct S*C-term float64 sum of the input cube.  For noise-level detections (the GUI's OS-CFAR: ~470
// per 256 x 128 frame, flat angle spectra) the bound flags ~13 % of the evaluations -- 60 per frame and antenna list -- and
// the direct sums (32768 complex multiply-adds per cell and antenna) took 29-36 us per frame, which is why round 3 shipped
// an empirical eighth of the bound as the default.  A cell costs far less when a frame's cells are computed TOGETHER:
//
//   Z[s][k]   = FFT_C( hann(C)[c] x[s][c] )[k]         one 128-point float64 FFT per sample row: 256 rows, 1.1 MFLOP per plane
//   X[r][k]   = sum_s hann(S)[s] W_S^(r s) Z[s][k]     256 complex multiply-adds per needed cell
//
// One 1024-thread workgroup per (frame, antenna of the list) walks the plane in passes of 64 rows: a group of 16 lanes
// transforms one row (8 points per lane in registers, then a 16-point transform across the lanes through the row's own 2 KB
// of LDS), the pass's spectra stay in the LDS (128 KB) and four lanes per needed cell add up their 16 rows each; the partial
// sums live in registers across the passes.  Then k_argmax64_list runs the float64 angle DFT + first-maximum argmax per
// flagged evaluation (argmax64_wave, the routine behind every other float64 argmax of the library).
// The frame's flagged detections are found through flagpos[f][det] (1 + position in the flagged list, written by
// k_angle_argmax next to the list itself), so nothing is sorted.
#pragma once
#include "mmw_ctx.h"
#include "mmw_misc.h"

namespace mmw {

struct Cells64Args {
    const float2 *cubes;        // [F][V][S][C] input cube
    const int32_t *dets;        // [F][cap][2] (range bin, fftshifted Doppler index)
    const int32_t *counts;      // [F]
    const int *flagpos;         // [F][cap]: 1 + position in the flagged list (0: not flagged, or beyond dense_cap)
    const int *n_flag;          // flagged evaluations of the call
    int dense_min;              // the dense form runs when *n_flag >= dense_min (the direct kernels when it is below)
    cplx<double> *out;          // [dense_cap][n_ant] float64 cells of the flagged evaluations
    int V, S, cap, n_ant, max_cells;
    AntList ants;
    const double *ws, *wc;      // np.hanning(S), np.hanning(C)
    const cplx<double> *twS, *twC;
};

constexpr int C64_NT = 1024, C64_ROWS = 64;
// LDS: the pass's spectra [64][C + 1], W_S, W_C, both windows, the needed cells (r << 16 | FFT bin; list position)
inline size_t cells64_lds(int S, int C, int max_cells) {
    return ((size_t)C64_ROWS * (C + 1) + S + C) * 16 + ((size_t)S + C) * 8 + (size_t)max_cells * 8 + 64;
}
inline int cells64_max_cells(int S, int C) {
    const long room = 160L * 1024 - 256 - (long)cells64_lds(S, C, 0);
    return room < 8 * 64 ? 0 : (int)std::min<long>(room / 8, 1024);
}

template <int C>
__global__ __launch_bounds__(C64_NT) void k_cells64(Cells64Args a) {
    static_assert(C == 128, "one Doppler row = 16 lanes x 8 points");
    constexpr int R = C / 16, P = C + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int n_cells_s;
    if (*a.n_flag < a.dense_min) return;
    const int S = a.S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n1 = lane & 15, g4 = lane >> 4;
    cplx<double> *Z = reinterpret_cast<cplx<double> *>(smem);
    cplx<double> *twS = Z + C64_ROWS * P, *twC = twS + S;
    double *wsl = reinterpret_cast<double *>(twC + C), *wcl = wsl + S;
    int *cell_rk = reinterpret_cast<int *>(wcl + C), *cell_e = cell_rk + a.max_cells;
    const long f = blockIdx.y;
    const int ai = blockIdx.x;
    int n_det = a.counts[f];
    if (n_det > a.cap) n_det = a.cap;
    if (n_det <= 0) return;
    for (int i = tid; i < S; i += C64_NT) {
        twS[i] = a.twS[i];
        wsl[i] = a.ws[i];
    }
    for (int i = tid; i < C; i += C64_NT) {
        twC[i] = a.twC[i];
        wcl[i] = a.wc[i];
    }
    const float2 *plane = a.cubes + (f * a.V + a.ants.idx[ai]) * (long)S * C;
    // chunks of max_cells detections (one for every realistic frame): the flagged ones of a chunk are this pass's cells
    for (int det0 = 0; det0 < n_det; det0 += a.max_cells) {
        if (tid == 0) n_cells_s = 0;
        __syncthreads();
        for (int det = det0 + tid; det < n_det && det < det0 + a.max_cells; det += C64_NT) {
            const int e = a.flagpos[f * a.cap + det] - 1;
            if (e >= 0) {
                const int r = a.dets[(f * a.cap + det) * 2];
                int k = a.dets[(f * a.cap + det) * 2 + 1] - C / 2;         // FFT bin behind the fftshifted Doppler index
                if (k < 0) k += C;
                const int pos = atomicAdd(&n_cells_s, 1);
                cell_rk[pos] = (r << 16) | k;
                cell_e[pos] = e;
            }
        }
        __syncthreads();
        const int n_cells = n_cells_s;
        if (n_cells == 0) continue;                                         // (uniform)
        // four lanes per cell; a thread's items are the same in every pass: partial sums in registers
        constexpr int ITEMS = 4;                                            // 4 * max_cells (<= 1024) / 1024 threads
        cplx<double> acc[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) acc[j] = cplx<double>{0.0, 0.0};
        // the samples of the NEXT pass travel while this one is transformed and summed (one workgroup per CU: nothing else
        // would cover the trip); unconditional clamped loads
        const int rl = 4 * wave + g4;
        float2 raw[R];
        auto fetch = [&](int s0) {
            const int s = s0 + rl, sc = s < S ? s : S - 1;
            const float2 *rowp = plane + (long)sc * C;
#pragma unroll
            for (int n2 = 0; n2 < R; ++n2) raw[n2] = rowp[n1 + 16 * n2];
        };
        fetch(0);
        for (int s0 = 0; s0 < S; s0 += C64_ROWS) {
            // ---- Doppler FFT of rows s0 .. s0 + 63: wave w, lane group g4 -> row 4 w + g4; lane n1 holds c = n1 + 16 n2
            {
                const int s = s0 + rl, sc = s < S ? s : S - 1;
                const double wrow = s < S ? wsl[sc] : 0.0;                  // hann(S)[s] rides along; rows past the plane: zero
                cplx<double> x[R];
#pragma unroll
                for (int n2 = 0; n2 < R; ++n2) {
                    const float2 t = raw[n2];
                    const double w = wcl[n1 + 16 * n2] * wrow;
                    x[n2] = cplx<double>{(double)t.x * w, (double)t.y * w};
                }
                fetch(s0 + C64_ROWS < S ? s0 + C64_ROWS : s0);
                RegFFT<R, double>::run(x);
                cplx<double> *slab = Z + rl * P;                            // the row's own LDS: [k2][n1] now, [k] afterwards
                static_for<R>([&](auto K) {
                    constexpr int k2 = decltype(K)::value;
                    slab[k2 * 16 + n1] = cmul(x[bitrev<R>(k2)], twC[(n1 * k2) & (C - 1)]);
                });
                // second level: lane (k2, h) = (n1 >> 1, n1 & 1) takes the bins k = k2 + 8 k1 with k1 = 2 j + h:
                //   Y[j] = FFT_8( (in[n] +- in[n + 8]) W_16^(h n) )   -- a radix-2 split of the 16-point transform over n1.
                // One wave, LDS operations in order: every lane's reads below are issued before any lane's writes.
                const int k2 = n1 >> 1, h = n1 & 1;
                cplx<double> y[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const cplx<double> p = slab[k2 * 16 + n], q = slab[k2 * 16 + n + 8];
                    const cplx<double> d = h ? p - q : p + q;
                    y[n] = h ? cmul(d, twC[(C / 16) * n]) : d;
                }
                RegFFT<8, double>::run(y);
                static_for<8>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    slab[k2 + R * (2 * j + h)] = y[bitrev<8>(j)];
                });
            }
            __syncthreads();
            // ---- range sums of the needed cells over this pass's rows (quarter q of a cell: rows s0 + 16 q .. + 15)
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int it = tid + j * C64_NT, i = it >> 2, q = it & 3;
                if (i < n_cells) {
                    const int rk = cell_rk[i], r = rk >> 16, k = rk & 0xffff;
                    const int sq = s0 + 16 * q;
                    cplx<double> c = twS[(int)(((long)r * sq) % S)];
                    const cplx<double> step = twS[r % S];
                    const cplx<double> *zp = Z + (16 * q) * P + k;
                    cplx<double> sum = acc[j];
#pragma unroll 4
                    for (int t = 0; t < 16; ++t) {                          // (rows past the plane hold zeros)
                        const cplx<double> z = zp[t * P];
                        sum.x = fma(z.x, c.x, fma(-z.y, c.y, sum.x));
                        sum.y = fma(z.x, c.y, fma(z.y, c.x, sum.y));
                        c = cmul(c, step);
                    }
                    acc[j] = sum;
                }
            }
            __syncthreads();
        }
        // ---- the four quarters of a cell sit in adjacent lanes
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int it = tid + j * C64_NT, i = it >> 2, q = it & 3;
            cplx<double> s = acc[j];
            for (int d = 1; d < 4; d <<= 1) {
                s.x += __shfl_xor(s.x, d, 64);
                s.y += __shfl_xor(s.y, d, 64);
            }
            if (i < n_cells && q == 0) a.out[(long)cell_e[i] * a.n_ant + ai] = s;
        }
        __syncthreads();
    }
}

// float64 angle DFT + first-maximum argmax of the flagged evaluations whose cells k_cells64 produced (one wave each)
struct Argmax64ListArgs {
    const cplx<double> *cells;  // [dense_cap][n_ant]
    const int *n_flag, *list;
    int dense_min, dense_cap, n_ant, A, shift;
    int32_t *out_idx;
    const cplx<double> *twA;
};
__global__ __launch_bounds__(256) void k_argmax64_list(Argmax64ListArgs a) {
    int n = *a.n_flag;
    if (n < a.dense_min) return;
    if (n > a.dense_cap) n = a.dense_cap;
    const int lane = threadIdx.x & 63;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
        const int idx = argmax64_wave(a.cells + (long)e * a.n_ant, a.n_ant, a.A, a.shift, a.twA, lane);
        if (lane == 0) a.out_idx[a.list[e] & 0x7fffffff] = idx;
    }
}

}  // namespace mmw
