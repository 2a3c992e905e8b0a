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
// One 512-thread workgroup per (frame, antenna of the list) walks the plane in passes of 64 rows: EIGHT lanes transform one row
// -- 16 points per lane: a 16-point register FFT, the W_128 twiddles from registers, one exchange through the row's own slab of
// the LDS (swizzled: element (k2, n1) at k2 * 8 + ((n1 + k2) & 7), so that the 128-byte runs a lane reads back spread over the
// banks), two 8-point register FFTs per lane -- the pass's spectra stay in the LDS and eight lanes per needed cell add up their
// 8 rows each; the partial sums live in registers across the passes.  Three 16-byte LDS operations per point (the first form --
// 16 lanes x 8 points, a radix-2 split of the cross-lane level, twiddles and windows from LDS tables -- made six, and the 16-byte
// stores cost 13 LDS cycles each: it was bound by them, 70 k clocks per plane).  Then k_argmax64_list runs the float64 angle
// DFT + first-maximum argmax per flagged evaluation (argmax64_wave, the routine behind every other float64 argmax of the library).
// The frame's flagged detections are found through flagpos[f][det] (1 + position in the flagged list, written by
// k_angle_argmax next to the list itself), so nothing is sorted.
#pragma once
#include "mmw_ctx.h"
#include "mmw_misc.h"
#include "mmw_bf16x3.h"

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

constexpr int C64_NT = 512, C64_ROWS = 64, C64_PITCH = 137, C64_CELLS = 256, C64_ITEMS = 8 * C64_CELLS / C64_NT;
// LDS: the pass's spectra [64][137], W_S, both windows, the chunk's cells (r << 16 | FFT bin; list position), wave counts
inline size_t cells64_lds(int S, int C, int) {
    return ((size_t)C64_ROWS * C64_PITCH + S) * 16 + ((size_t)S + C) * 8 + (size_t)C64_CELLS * 8 + 64;
}
inline int cells64_max_cells(int S, int C) {        // cells per chunk of a frame (0: the plane's tables do not fit the LDS)
    return C == 128 && cells64_lds(S, C, 0) <= 160 * 1024 - 512 ? C64_CELLS : 0;
}

template <int C>
__global__ __launch_bounds__(C64_NT) void k_cells64(Cells64Args a) {
    static_assert(C == 128, "one Doppler row = 8 lanes x 16 points");
    constexpr int P = C64_PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (*a.n_flag < a.dense_min) return;
    const int S = a.S, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n1 = tid & 7, rl = tid >> 3;
    cplx<double> *Z = reinterpret_cast<cplx<double> *>(smem);
    cplx<double> *twS = Z + C64_ROWS * P;
    double *wsl = reinterpret_cast<double *>(twS + S), *wcl = wsl + S;
    int *cell_rk = reinterpret_cast<int *>(wcl + C), *cell_e = cell_rk + C64_CELLS, *wcnt = cell_e + C64_CELLS;
    const long f = blockIdx.y;
    const int ai = blockIdx.x;
    int n_det = a.counts[f];
    if (n_det > a.cap) n_det = a.cap;
    if (n_det <= 0) return;
    for (int i = tid; i < S; i += C64_NT) {
        twS[i] = a.twS[i];
        wsl[i] = a.ws[i];
    }
    for (int i = tid; i < C; i += C64_NT) wcl[i] = a.wc[i];
    // W_128^(n1 k2), k2 = 1 .. 15: the lane's inter-level twiddles, in registers for the whole plane
    cplx<double> tw1[16];
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) tw1[k2] = a.twC[(n1 * k2) & (C - 1)];
    const float2 *plane = a.cubes + (f * a.V + a.ants.idx[ai]) * (long)S * C;
    cplx<double> *slab = Z + rl * P;                                    // the row's own LDS: (k2, n1) swizzled first, [k] afterwards
    // chunks of C64_CELLS flagged detections in list order (one chunk for every realistic frame)
    for (int c0 = 0;; c0 += C64_CELLS) {
        // ---- the chunk's cells: flagged detections with ordinal c0 .. c0 + 255 (ordinal = rank among the flagged, detection order)
        int running = 0;
        for (int det0 = 0; det0 < n_det; det0 += C64_NT) {
            const int det = det0 + tid;
            const int e = det < n_det ? a.flagpos[f * a.cap + det] - 1 : -1;
            const unsigned long long bal = __ballot(e >= 0);
            if (lane == 0) wcnt[wave] = __popcll(bal);
            __syncthreads();
            int before = running, total = running;
#pragma unroll
            for (int w = 0; w < C64_NT / 64; ++w) {
                const int cw = wcnt[w];
                if (w < wave) before += cw;
                total += cw;
            }
            const int ord = before + __popcll(bal & ((1ull << lane) - 1ull)) - c0;
            if (e >= 0 && ord >= 0 && ord < C64_CELLS) {
                const int r = a.dets[(f * a.cap + det) * 2];
                int k = a.dets[(f * a.cap + det) * 2 + 1] - C / 2;     // FFT bin behind the fftshifted Doppler index
                if (k < 0) k += C;
                cell_rk[ord] = (r << 16) | k;
                cell_e[ord] = e;
            }
            running = total;
            __syncthreads();
        }
        const int n_cells = running - c0 < C64_CELLS ? running - c0 : C64_CELLS;
        if (n_cells <= 0) break;                                        // (uniform)
        // eight lanes per cell; a thread's items are the same in every pass: partial sums in registers
        cplx<double> acc[C64_ITEMS];
#pragma unroll
        for (int j = 0; j < C64_ITEMS; ++j) acc[j] = cplx<double>{0.0, 0.0};
        // the samples of the NEXT pass travel while this one is transformed and summed; unconditional clamped loads
        float2 raw[16];
        auto fetch = [&](int s0) {
            const int s = s0 + rl, sc = s < S ? s : S - 1;
            const float2 *rowp = plane + (long)sc * C;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) raw[n2] = rowp[n1 + 8 * n2];
        };
        fetch(0);
        for (int s0 = 0; s0 < S; s0 += C64_ROWS) {
            // ---- Doppler FFT of rows s0 .. s0 + 63: lanes 8 rl .. 8 rl + 7 -> row rl; lane n1 holds c = n1 + 8 n2
            {
                const int s = s0 + rl;
                const double wrow = s < S ? wsl[s] : 0.0;                   // hann(S)[s] rides along; rows past the plane: zero
                cplx<double> x[16];
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) {
                    const double w = wcl[n1 + 8 * n2] * wrow;
                    x[n2] = cplx<double>{(double)raw[n2].x * w, (double)raw[n2].y * w};
                }
                fetch(s0 + C64_ROWS < S ? s0 + C64_ROWS : s0);
                RegFFT<16, double>::run(x);                                 // over n2: X1[k2] in x[bitrev(k2)]
                static_for<16>([&](auto K) {
                    constexpr int k2 = decltype(K)::value;
                    const cplx<double> v = x[bitrev<16>(k2)];
                    slab[k2 * 8 + ((n1 + k2) & 7)] = k2 == 0 ? v : cmul(v, tw1[k2]);
                });
                __builtin_amdgcn_wave_barrier();
                // second level over n1 (the row's eight lanes, one wave: LDS operations of a wave execute in order): lane n1 takes
                // k2 = n1 and n1 + 8, bins k = k2 + 16 k1
                cplx<double> ya[8], yb[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    ya[n] = slab[n1 * 8 + ((n + n1) & 7)];
                    yb[n] = slab[(n1 + 8) * 8 + ((n + n1) & 7)];           // ((n + n1 + 8) & 7 == (n + n1) & 7)
                }
                RegFFT<8, double>::run(ya);
                RegFFT<8, double>::run(yb);
                __builtin_amdgcn_wave_barrier();
                static_for<8>([&](auto K) {
                    constexpr int k1 = decltype(K)::value;
                    slab[n1 + 16 * k1] = ya[bitrev<8>(k1)];
                    slab[n1 + 8 + 16 * k1] = yb[bitrev<8>(k1)];
                });
            }
            __syncthreads();
            // ---- range sums of the needed cells over this pass's rows (eighth q of a cell: rows s0 + 8 q .. + 7)
#pragma unroll
            for (int j = 0; j < C64_ITEMS; ++j) {
                const int it = tid + j * C64_NT, i = it >> 3, q = it & 7;
                if (i < n_cells) {
                    const int rk = cell_rk[i], r = rk >> 16, k = rk & 0xffff;
                    const int sq = s0 + 8 * q;
                    cplx<double> c = twS[(int)(((long)r * sq) % S)];
                    const cplx<double> step = twS[r % S];
                    const cplx<double> *zp = Z + (8 * q) * P + k;
                    cplx<double> sum = acc[j];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {                           // (rows past the plane hold zeros)
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
        // ---- the eight parts of a cell sit in adjacent lanes
#pragma unroll
        for (int j = 0; j < C64_ITEMS; ++j) {
            const int it = tid + j * C64_NT, i = it >> 3, q = it & 7;
            cplx<double> s = acc[j];
            for (int d = 1; d < 8; d <<= 1) {
                s.x += __shfl_xor(s.x, d, 64);
                s.y += __shfl_xor(s.y, d, 64);
            }
            if (i < n_cells && q == 0) a.out[(long)cell_e[i] * a.n_ant + ai] = s;
        }
        if (running <= c0 + C64_CELLS) break;                           // (uniform) no further chunk
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// |RD| of ONE antenna in float64 for 256 x 128 planes, one launch (mmw_range_doppler_mag64: the plane the reference's detectors
// threshold, range_doppler_detector.py:62-80).  The generic path is two launches around a complex128 plane in memory
// (0.57 us per frame).  Here a 512-thread workgroup walks frames:
//   A  Doppler FFT of the rows in passes of 64 rows -- the 8 lanes x 16 points form of k_cells64 above --, each pass's spectra
//      leave the LDS TRANSPOSED into the workgroup's own 512 KB of scratch, T[k][s] (1 KB runs);
//   B  range FFT down the columns, 32 columns at a time: 16 lanes x 16 points per column (16-point register FFT over s2, the
//      W_256 twiddles, one swizzled exchange through the LDS, a second 16-point register FFT), |.| = hypot, fftshifted
//      Doppler index, straight to the output (32 consecutive doubles per row and store).
// The scratch is written and read by the same workgroup only and rewritten every frame: it lives in the L2 / Infinity Cache;
// the reads bypass the CU's L1 (sc0), which may still hold the previous frame's lines.
struct Rd64Args {
    const float2 *cubes;        // first sample of frame 0's plane
    long frame_stride;          // complex elements between the planes of consecutive frames
    double *mag;                // [F][256][128]
    cplx<double> *scratch;      // [grid][128][256]
    int n_frames;
    const double *ws, *wc;      // np.hanning(256), np.hanning(128)
    const cplx<double> *twS, *twC;
};
constexpr int R64_S = 256, R64_C = 128, R64_COLS = 32, R64_CP = 257;
inline size_t rd64_lds() { return (std::max((size_t)C64_ROWS * C64_PITCH, (size_t)R64_COLS * R64_CP) + R64_S + R64_C) * 16 + (R64_S + R64_C) * 8; }

__global__ __launch_bounds__(C64_NT) void k_rd_mag64_256x128(Rd64Args a) {
    constexpr int S = R64_S, C = R64_C, P = C64_PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    cplx<double> *Z = reinterpret_cast<cplx<double> *>(smem);
    cplx<double> *twl = Z + (C64_ROWS * P > R64_COLS * R64_CP ? C64_ROWS * P : R64_COLS * R64_CP);       // W_256
    cplx<double> *twc = twl + S;                                                                          // W_128
    double *wsl = reinterpret_cast<double *>(twc + C), *wcl = wsl + S;
    for (int i = tid; i < S; i += C64_NT) {
        wsl[i] = a.ws[i];
        twl[i] = a.twS[i];
    }
    for (int i = tid; i < C; i += C64_NT) {
        wcl[i] = a.wc[i];
        twc[i] = a.twC[i];
    }
    cplx<double> *T = a.scratch + (size_t)blockIdx.x * S * C;
    const auto t_rs = __builtin_amdgcn_make_buffer_rsrc(T, 0, S * C * 16, 0x00020000);
    __syncthreads();
    for (long f = blockIdx.x; f < a.n_frames; f += gridDim.x) {
        const float2 *plane = a.cubes + f * a.frame_stride;
        // ---- A: Doppler
        {
            const int n1 = tid & 7, rl = tid >> 3;
            cplx<double> *slab = Z + rl * P;
            float2 raw[16];
            auto fetch = [&](int s0) {
                const float2 *rowp = plane + (long)(s0 + rl) * C;
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) raw[n2] = rowp[n1 + 8 * n2];
            };
            fetch(0);
            for (int s0 = 0; s0 < S; s0 += C64_ROWS) {
                const double wrow = wsl[s0 + rl];
                cplx<double> x[16];
#pragma unroll
                for (int n2 = 0; n2 < 16; ++n2) {
                    const double w = wcl[n1 + 8 * n2] * wrow;
                    x[n2] = cplx<double>{(double)raw[n2].x * w, (double)raw[n2].y * w};
                }
                fetch(s0 + C64_ROWS < S ? s0 + C64_ROWS : s0);
                RegFFT<16, double>::run(x);
                static_for<16>([&](auto K) {
                    constexpr int k2 = decltype(K)::value;
                    const cplx<double> v = x[bitrev<16>(k2)];
                    slab[k2 * 8 + ((n1 + k2) & 7)] = k2 == 0 ? v : cmul(v, twc[(n1 * k2) & (C - 1)]);
                });
                __builtin_amdgcn_wave_barrier();
                cplx<double> ya[8], yb[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    ya[n] = slab[n1 * 8 + ((n + n1) & 7)];
                    yb[n] = slab[(n1 + 8) * 8 + ((n + n1) & 7)];
                }
                RegFFT<8, double>::run(ya);
                RegFFT<8, double>::run(yb);
                __builtin_amdgcn_wave_barrier();
                static_for<8>([&](auto K) {
                    constexpr int k1 = decltype(K)::value;
                    slab[n1 + 16 * k1] = ya[bitrev<8>(k1)];
                    slab[n1 + 8 + 16 * k1] = yb[bitrev<8>(k1)];
                });
                __syncthreads();
                // the pass's spectra, transposed: T[k][s0 .. s0 + 63]
#pragma unroll 4
                for (int i = tid; i < C * C64_ROWS; i += C64_NT) {
                    const int k = i >> 6, r = i & 63;
                    T[k * S + s0 + r] = Z[r * P + k];
                }
                __syncthreads();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave's stores of T have left ...
        __syncthreads();                                         // ... before any wave reads T
        // ---- B: range, 32 columns at a time
        {
            const int m1 = tid & 15, cl = tid >> 4;             // lane of the column, column of the round
            cplx<double> *slab = Z + cl * R64_CP;
            double *out = a.mag + f * (long)S * C;
            for (int k0 = 0; k0 < C; k0 += R64_COLS) {
                const int k = k0 + cl;
                cplx<double> x[16];
#pragma unroll
                for (int m2 = 0; m2 < 16; ++m2) {
                    const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(t_rs, (unsigned)((k * S + m1 + 16 * m2) * 16), 0, 1));
                    x[m2] = __builtin_bit_cast(cplx<double>, v);
                }
                RegFFT<16, double>::run(x);                     // over m2: Y1[q] in x[bitrev(q)]
                static_for<16>([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    const cplx<double> v = x[bitrev<16>(q)];
                    slab[q * 16 + ((m1 + q) & 15)] = q == 0 ? v : cmul(v, twl[(m1 * q) & (S - 1)]);
                });
                __builtin_amdgcn_wave_barrier();
                cplx<double> y[16];
#pragma unroll
                for (int n = 0; n < 16; ++n) y[n] = slab[m1 * 16 + ((n + m1) & 15)];
                RegFFT<16, double>::run(y);                     // over m1: range bin r = m1 + 16 p in y[bitrev(p)]
                int d = k + C / 2;
                if (d >= C) d -= C;
                static_for<16>([&](auto Pp) {
                    constexpr int p = decltype(Pp)::value;
                    const cplx<double> v = y[bitrev<16>(p)];
                    out[(long)(m1 + 16 * p) * C + d] = hypot(v.x, v.y);
                });
                __builtin_amdgcn_wave_barrier();                // (the column's slab is reused by the same lanes in the next round)
            }
        }
        __syncthreads();                                         // the LDS goes back to phase A
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
