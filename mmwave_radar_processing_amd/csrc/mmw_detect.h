// Fused per-frame detection stage of the batch pipeline (BASELINE configs[2]): everything behind the range-Doppler
// kernel in ONE launch, one 1024-thread workgroup per frame.
//
//   RangeDopplerDetector._compute_range_doppler_response   processors/range_doppler_detection/range_doppler_detector.py:62-80
//   CaCFAR2D / OsCFAR2D .detect                             detectors/ca_cfar.py:85-155, os_cfar.py:97-195, base.py:208-230
//   PointCloudGenerator._compute_angle_estimation           processors/point_cloud_generator.py:143-214
//
// The reference thresholds the float64 magnitudes of antenna 0.  Computing that plane in float64 for every frame, the
// CFAR over it, the compaction and two argmax launches took 60 % of the pipeline's time in round 2 (five latency-bound
// launches behind the one bandwidth-bound kernel).  Here the decision is SCREENED on the float32 range-Doppler plane
// the RD kernel has just written, with a rigorous error band:
//
//   |X32 - X64| <= B = k_fft * l1(plane 0)                      (same bound as the exact argmax: every RD cell is a sum
//                                                                of S*C products w x W with <= ulps roundings on the way)
//   T' = alpha * mean(X32 over the training cells), in float64   =>  |T' - T64| <= alpha * B   (CA)
//   T' = alpha * (k-th smallest X32 of the training cells)       =>  |T' - T64| <= alpha * B   (OS: order statistics are
//                                                                                               1-Lipschitz in the sup norm)
//   d = X32 - T':   d >  band  => detection for certain,   d <= -band => certainly none,   band = (1 + alpha) B + 1e-7 (X32 + T')
//
// (the 1e-7 term covers the rounding of |.| to float32 for the LDS plane).  Cells inside the band -- one frame in ~13 of
// the synthetic workload has one -- are decided EXACTLY: k_cfar_cell_exact evaluates the (2 hr + 1) x (2 hd + 1) window
// of float64 range-Doppler magnitudes around the cell as direct float64 DFT sums of the input cube and applies the
// reference's float64 rule.  Frames with such cells skip compaction in the screening kernel and are finished by
// k_detect_finish once their bit masks are complete.  A frame whose input is not finite, or that has more undecided cells
// than the list holds, gets counts[f] = -1: the caller runs it through the float64 path (mmw_detect_batch).
//
// Compaction is ordered (np.where order, base.py:229-230); the per-detection angle argmax (azimuth and elevation lists
// in the same pass, float32 with the error bound of k_angle_argmax, flagged near-ties re-evaluated in float64 by the
// k_argmax_refine_* kernels) runs in the same workgroup.
#pragma once
#include "mmw_ctx.h"
#include "mmw_misc.h"

namespace mmw {

constexpr int DET_NT = 1024;             // threads of the per-frame workgroup
constexpr int DET_MAX_ANT = 16;          // antennas per list inside the fused kernels
enum { DCTL_FLAG_FRAMES = 0, DCTL_CELLS = 1, DCTL_FALLBACK = 2, DCTL_AZ = 16, DCTL_EL = 32, DCTL_WORDS = 64 };

struct DetAnt {            // antenna list of one angle estimate (n == 0: not wanted)
    int n;
    int idx[DET_MAX_ANT];
};

struct DetectArgs {
    const float2 *rd;          // [F][V][S][C] float32 range-Doppler cube
    const float *l1;           // [F][V] plane L1 norms (error-bound scale)
    float *mag32;              // optional [F][S][C]: |RD| of antenna 0
    int32_t *dets, *counts;    // [F][cap][2], [F]
    int32_t *az_idx, *el_idx;  // [F][cap] each (nullptr with an empty list)
    unsigned *bits;            // [F][words] detection bit masks of the frames with undecided cells
    int *ctl;                  // DCTL_* counters
    int *flag_frames;          // [F] frames with undecided cells
    int *cells;                // [cell_cap][2] undecided cells: (frame, r * C + c)
    int cell_cap;
    int V, S, C, cap, words, band_rows;
    int kind, tr, td, gr, gd, n_train, k_rank;
    double scale;
    float k_fft;               // ulps * 2^-24 of the RD kernel that ran
    DetAnt az, el;
    int A, shift_az, shift_el;
    const float2 *twA;
    ArgmaxRefine rf_az, rf_el;
};

// exclusive prefix of v over the workgroup (thread order), total in *total; ws: 40 ints of LDS
__device__ __forceinline__ int block_excl_scan(int v, int *ws, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int s = lane < DET_NT / 64 ? ws[lane] : 0;
        int inc2 = s;
        for (int d = 1; d < DET_NT / 64; d <<= 1) {
            const int t = __shfl_up(inc2, d, 64);
            if (lane >= d) inc2 += t;
        }
        if (lane < DET_NT / 64) ws[20 + lane] = inc2 - s;
        if (lane == DET_NT / 64 - 1) ws[20 + DET_NT / 64] = inc2;
    }
    __syncthreads();
    const int excl = incl - v + ws[20 + wave];
    *total = ws[20 + DET_NT / 64];
    __syncthreads();
    return excl;
}

// antenna tables of the two lists in LDS (tab[0..16) azimuth, tab[16..32) elevation): constant kernel-argument indices
// here, lane-indexed reads later
__device__ __forceinline__ void detect_ant_table(const DetectArgs &a, int *tab) {
    if (threadIdx.x == 0) {
        static_for<DET_MAX_ANT>([&](auto I) {
            constexpr int i = decltype(I)::value;
            tab[i] = a.az.idx[i];
            tab[DET_MAX_ANT + i] = a.el.idx[i];
        });
    }
}

// One wave, one detection, one antenna list whose cells sit in lanes [base, base + n) of xl: zero-padded A-point DFT,
// |.|, first maximum -- k_angle_argmax's arithmetic with the cells read out of the lanes (v_readlane) instead of held
// n-fold in every lane's registers.
__device__ __forceinline__ void detect_argmax_lanes(const DetectArgs &a, float2 xl, int base, int n, int shift, int32_t *out_idx,
                                                    const ArgmaxRefine &rf, float sum_l1, long slot, int lane) {
    const float NEG = -__builtin_huge_valf();
    const int A = a.A;
    float best = NEG, second = NEG;
    int best_idx = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        float re = 0.f, im = 0.f;
        int t = 0;
        for (int i = 0; i < n; ++i) {
            const float xr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xl.x), base + i));
            const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xl.y), base + i));
            const float2 w = a.twA[t];
            re += xr * w.x - xi * w.y;
            im += xr * w.y + xi * w.x;
            t += k;
            if (t >= A) t -= A;
        }
        const float m = hypotf(re, im);
        const int kk = shift ? (k + A / 2) % A : k;
        if (best_idx == 0x7fffffff || mag_better(m, kk, best, best_idx)) {
            if (best_idx != 0x7fffffff) second = best;
            best = m;
            best_idx = kk;
        } else if (mag_gt(m, second)) second = m;
    }
    float wb = best;
    int wi = best_idx;
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(wb, d, 64);
        const int oi = __shfl_xor(wi, d, 64);
        if (oi != 0x7fffffff && (wi == 0x7fffffff || mag_better(ob, oi, wb, wi))) {
            wb = ob;
            wi = oi;
        }
    }
    float ws2 = (best_idx == wi) ? second : best;
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(ws2, d, 64);
        if (mag_gt(o, ws2)) ws2 = o;
    }
    // sum |re| + |im| of the list's cells (scale of the angle-DFT term of the bound)
    float sum_abs = (lane >= base && lane < base + n) ? fabsf(xl.x) + fabsf(xl.y) : 0.f;
    for (int d = 32; d >= 1; d >>= 1) sum_abs += __shfl_xor(sum_abs, d, 64);
    if (lane == 0) {
        out_idx[slot] = wi;
        argmax_flag(rf, sum_l1, sum_abs, wb, ws2, (int)slot);
    }
}

// Ordered compaction of the frame's bit mask (bit r * C + c, LDS) into dets / counts, then the angle argmax of every
// detection.  ws: 96 ints of LDS ([48, 80) = antenna table).  Ends with every thread past its last use of bits / ws.
__device__ __forceinline__ void detect_finish(const DetectArgs &a, long f, const unsigned *bits, int *ws) {
    const int tid = threadIdx.x, C = a.C;
    int base = 0;
    for (int w0 = 0; w0 < a.words; w0 += DET_NT) {
        const int w = w0 + tid;
        unsigned word = w < a.words ? bits[w] : 0u;
        int total;
        int pos = base + block_excl_scan(__popc(word), ws, &total);
        while (word) {
            const int b = __ffs(word) - 1;
            word &= word - 1;
            if (pos < a.cap) {
                const int idx = w * 32 + b, r = idx / C;
                a.dets[(f * a.cap + pos) * 2] = r;
                a.dets[(f * a.cap + pos) * 2 + 1] = idx - r * C;
            }
            ++pos;
        }
        base += total;
    }
    if (tid == 0) a.counts[f] = base;           // exact even beyond cap (MMW_ERR_TRUNCATED is the caller's check)
    const int n_az = a.az.n, n_el = a.el.n;
    if (n_az == 0 && n_el == 0) return;
    __syncthreads();                            // the workgroup's own dets are visible to all of its waves
    const int n_det = base < a.cap ? base : a.cap;
    const int lane = tid & 63, wave = tid >> 6;
    const int *tab = ws + 48;
    // lanes [0, n_az): azimuth list, lanes [16, 16 + n_el): elevation list
    const bool mine = (lane < n_az) || (lane >= DET_MAX_ANT && lane < DET_MAX_ANT + n_el);
    const long ant = mine ? tab[lane] : 0;
    float l1v = mine ? a.l1[f * a.V + ant] : 0.f, l1_az = lane < DET_MAX_ANT ? l1v : 0.f, l1_el = lane >= DET_MAX_ANT ? l1v : 0.f;
    for (int d = 32; d >= 1; d >>= 1) {
        l1_az += __shfl_xor(l1_az, d, 64);
        l1_el += __shfl_xor(l1_el, d, 64);
    }
    const float2 *plane = a.rd + (f * a.V + ant) * (long)a.S * C;
    for (int det = wave; det < n_det; det += DET_NT / 64) {
        const long slot = f * a.cap + det;
        const int r = a.dets[slot * 2], c = a.dets[slot * 2 + 1];
        const float2 xl = mine ? plane[(long)r * C + c] : make_float2(0.f, 0.f);
        if (n_az) detect_argmax_lanes(a, xl, 0, n_az, a.shift_az, a.az_idx, a.rf_az, l1_az, slot, lane);
        if (n_el) detect_argmax_lanes(a, xl, DET_MAX_ANT, n_el, a.shift_el, a.el_idx, a.rf_el, l1_el, slot, lane);
    }
}

// LDS of k_detect_screen: float32 plane, two float64 band buffers, bit mask, 96 ints
inline size_t detect_screen_lds(int S, int C, int band_rows) {
    const size_t n = (size_t)S * C, words = (n + 31) / 32;
    return ((n * 4 + 15) & ~(size_t)15) + 2 * (size_t)band_rows * C * 8 + ((words * 4 + 15) & ~(size_t)15) + 96 * 4;
}

__global__ __launch_bounds__(DET_NT) void k_detect_screen(DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, C = a.C, n = S * C, tid = threadIdx.x;
    const long f = blockIdx.x;
    float *Xs = reinterpret_cast<float *>(smem);
    size_t off = ((size_t)n * 4 + 15) & ~(size_t)15;
    double *Vw = reinterpret_cast<double *>(smem + off);
    off += (size_t)a.band_rows * C * 8;
    double *Vg = reinterpret_cast<double *>(smem + off);
    off += (size_t)a.band_rows * C * 8;
    unsigned *bits = reinterpret_cast<unsigned *>(smem + off);
    off += ((size_t)a.words * 4 + 15) & ~(size_t)15;
    int *ws = reinterpret_cast<int *>(smem + off);          // [0, 40) scan, 44 undecided, 45 overflow
    if (tid < 2) ws[44 + tid] = 0;
    detect_ant_table(a, ws + 48);
    for (int w = tid; w < a.words; w += DET_NT) bits[w] = 0u;

    // error band scale; 1.0001 covers the float32 summation of the L1 norm itself
    const double Bf = (double)a.k_fft * (double)a.l1[f * a.V] * 1.0001;
    const bool degenerate = !(Bf >= 0.0 && Bf <= 1e300);       // NaN / inf samples in antenna 0
    // |RD| of antenna 0: float64 square root of the exact float64 sum of squares, rounded once to float32
    {
        const float2 *p = a.rd + f * a.V * n;
        float *mg = a.mag32 ? a.mag32 + f * n : nullptr;
        if ((n & 1) == 0) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 *p4 = reinterpret_cast<const f4 *>(p);
            for (int i = tid; i < n / 2; i += DET_NT) {
                const f4 v = p4[i];
                const float m0 = (float)sqrt((double)v.x * v.x + (double)v.y * v.y);
                const float m1 = (float)sqrt((double)v.z * v.z + (double)v.w * v.w);
                *reinterpret_cast<float2 *>(Xs + 2 * i) = make_float2(m0, m1);
                if (mg) *reinterpret_cast<float2 *>(mg + 2 * i) = make_float2(m0, m1);
            }
        } else {
            for (int i = tid; i < n; i += DET_NT) {
                const float2 v = p[i];
                const float m = (float)sqrt((double)v.x * v.x + (double)v.y * v.y);
                Xs[i] = m;
                if (mg) mg[i] = m;
            }
        }
    }
    __syncthreads();

    const int hr = a.tr + a.gr, hd = a.td + a.gd;
    const int lane = tid & 63;
    if (!degenerate && S > 2 * hr && C > 2 * hd) {
        const double inv_n = 1.0 / (double)a.n_train, alpha = a.scale, band0 = (1.0 + fabs(alpha)) * Bf;
        for (int r0 = hr; r0 < S - hr; r0 += a.band_rows) {
            const int nb = min(a.band_rows, S - hr - r0), cells = nb * C;
            // column sums over the window rows (Vw) and over the guard rows (Vg), every column
            for (int i = tid; i < cells; i += DET_NT) {
                const int rr = i / C, c = i - rr * C;
                const float *col = Xs + (r0 + rr - hr) * C + c;
                double w = 0.0, g = 0.0;
                for (int dr = 0; dr <= 2 * hr; ++dr) {
                    const double v = (double)col[dr * C];
                    w += v;
                    if (dr >= a.tr && dr <= a.tr + 2 * a.gr) g += v;
                }
                Vw[i] = w;
                Vg[i] = g;
            }
            __syncthreads();
            for (int i0 = tid - lane; i0 < cells; i0 += DET_NT) {        // wave-uniform trip count (ballot below)
                const int i = i0 + lane;
                bool det = false, unc = false;
                int r = 0, c = 0;
                if (i < cells) {
                    const int rr = i / C;
                    c = i - rr * C;
                    r = r0 + rr;
                    if (c >= hd && c < C - hd) {
                        double tot = 0.0;
                        const double *pw = Vw + rr * C + c - hd;
                        for (int dc = 0; dc <= 2 * hd; ++dc) tot += pw[dc];
                        const double *pg = Vg + rr * C + c - a.gd;
                        for (int dc = 0; dc <= 2 * a.gd; ++dc) tot -= pg[dc];
                        const double X = (double)Xs[r * C + c], T = alpha * (tot * inv_n);
                        const double d = X - T, band = band0 + 1.0e-7 * (X + fabs(T));
                        det = d > band;
                        unc = !det && !(d <= -band);
                    }
                }
                const unsigned long long m = __ballot(det);
                if (m && lane == 0) {
                    const long b0 = (long)r0 * C + i0;
                    const int wd = (int)(b0 >> 5), sh = (int)(b0 & 31);
                    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
                    const unsigned w0 = lo << sh, w1 = (sh ? lo >> (32 - sh) : 0u) | (hi << sh), w2 = sh ? hi >> (32 - sh) : 0u;
                    if (w0) atomicOr(&bits[wd], w0);
                    if (w1) atomicOr(&bits[wd + 1], w1);
                    if (w2) atomicOr(&bits[wd + 2], w2);
                }
                if (unc) {
                    const int pos = atomicAdd(a.ctl + DCTL_CELLS, 1);
                    if (pos < a.cell_cap) {
                        a.cells[2 * pos] = (int)f;
                        a.cells[2 * pos + 1] = r * C + c;
                        ws[44] = 1;
                    } else
                        ws[45] = 1;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (degenerate || ws[45]) {                  // the float64 path decides this frame
        if (tid == 0) {
            a.counts[f] = -1;
            atomicAdd(a.ctl + DCTL_FALLBACK, 1);
        }
        return;
    }
    if (ws[44]) {                                // undecided cells: mask out, compaction after k_cfar_cell_exact
        for (int w = tid; w < a.words; w += DET_NT) a.bits[f * a.words + w] = bits[w];
        if (tid == 0) a.flag_frames[atomicAdd(a.ctl + DCTL_FLAG_FRAMES, 1)] = (int)f;
        return;
    }
    detect_finish(a, f, bits, ws);
}

// Frames whose masks were completed by k_cfar_cell_exact: compaction + argmax.  Persistent over the flagged list.
__global__ __launch_bounds__(DET_NT) void k_detect_finish(DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *bits = reinterpret_cast<unsigned *>(smem);
    int *ws = reinterpret_cast<int *>(smem + (((size_t)a.words * 4 + 15) & ~(size_t)15));
    const int n = a.ctl[DCTL_FLAG_FRAMES];
    detect_ant_table(a, ws + 48);
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        const long f = a.flag_frames[e];
        for (int w = threadIdx.x; w < a.words; w += DET_NT) bits[w] = a.bits[f * a.words + w];
        __syncthreads();
        detect_finish(a, f, bits, ws);
        __syncthreads();
    }
}

// Exact float64 decision of the undecided cells.  One 256-thread workgroup per cell:
//   Y[row][ch]  = hann(C)[ch] * sum_s hann(S)[s] x[s][ch] W_S^(row * s)        rows r - hr .. r + hr      (direct sums)
//   M[row][j]   = | sum_ch Y[row][ch] W_C^(k_j * ch) |,  k_j = FFT bin behind the fftshifted Doppler index c - hd + j
// then the reference's rule on the window M: CA  X > alpha * (sum of training cells / N)   (ca_cfar.py:134-153)
//                                            OS  X > alpha * (k-th smallest training cell) (os_cfar.py:176-193)
struct CellExactArgs {
    const float2 *cubes;       // [F][V][S][C] input
    const int *cells, *n_cells;
    int cell_cap;
    unsigned *bits;
    int V, S, C, words;
    int kind, tr, td, gr, gd, n_train, k_rank;
    double scale;
    const double *ws, *wc;
    const cplx<double> *twS, *twC;
};

inline size_t cell_exact_lds(int C, int Wr, int Wd) { return (size_t)Wr * C * 16 + (size_t)Wr * Wd * 8 + 64; }

__global__ __launch_bounds__(256) void k_cfar_cell_exact(CellExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, C = a.C, hr = a.tr + a.gr, hd = a.td + a.gd, Wr = 2 * hr + 1, Wd = 2 * hd + 1, tid = threadIdx.x;
    cplx<double> *Y = reinterpret_cast<cplx<double> *>(smem);
    double *M = reinterpret_cast<double *>(smem + (size_t)Wr * C * 16);
    double *res = M + Wr * Wd;          // [0] OS threshold cell
    int n = *a.n_cells;
    if (n > a.cell_cap) n = a.cell_cap;
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        const long f = a.cells[2 * e];
        const int cell = a.cells[2 * e + 1], r = cell / C, c = cell - r * C;
        const float2 *x = a.cubes + f * a.V * S * C;                 // antenna 0
        for (int o = tid; o < Wr * C; o += 256) {
            const int row = o / C, ch = o - row * C;
            int k = r - hr + row;                                   // range bin (inside the plane: valid region)
            cplx<double> acc = cplx<double>{0.0, 0.0};
            int idx = 0;
            for (int s = 0; s < S; ++s) {
                const float2 v = x[(long)s * C + ch];
                const double w = a.ws[s];
                acc = acc + cmul(cplx<double>{(double)v.x * w, (double)v.y * w}, a.twS[idx]);
                idx += k;
                if (idx >= S) idx -= S;
            }
            Y[o] = acc * a.wc[ch];
        }
        __syncthreads();
        for (int o = tid; o < Wr * Wd; o += 256) {
            const int row = o / Wd, j = o - row * Wd;
            int k = c - hd + j - C / 2;                              // np.fft.fftshift: out[i] = X[(i - C//2) mod C]
            if (k < 0) k += C;
            cplx<double> acc = cplx<double>{0.0, 0.0};
            int idx = 0;
            for (int ch = 0; ch < C; ++ch) {
                acc = acc + cmul(Y[row * C + ch], a.twC[idx]);
                idx += k;
                if (idx >= C) idx -= C;
            }
            M[o] = hypot(acc.x, acc.y);
        }
        __syncthreads();
        auto guard = [&](int o) {
            const int row = o / Wd, j = o - row * Wd;
            return row >= a.tr && row <= a.tr + 2 * a.gr && j >= a.td && j <= a.td + 2 * a.gd;
        };
        bool det = false;
        const double X = M[hr * Wd + hd];
        if (a.kind == MMW_CFAR_CA) {
            if (tid == 0) {
                double tot = 0.0;
                for (int o = 0; o < Wr * Wd; ++o) tot += guard(o) ? 0.0 : M[o];
                det = X > a.scale * (tot / (double)a.n_train);
            }
        } else {
            // k-th smallest training cell by rank counting (ties broken by position, as a stable sort would)
            for (int o = tid; o < Wr * Wd; o += 256) {
                if (guard(o)) continue;
                const double v = M[o];
                int rank = 0;
                for (int q = 0; q < Wr * Wd; ++q) {
                    if (guard(q)) continue;
                    const double u = M[q];
                    rank += (u < v || (u == v && q < o)) ? 1 : 0;
                }
                if (rank == a.k_rank - 1) res[0] = v;
            }
            __syncthreads();
            if (tid == 0) det = X > a.scale * res[0];
        }
        if (tid == 0 && det) atomicOr(a.bits + f * a.words + (cell >> 5), 1u << (cell & 31));
        __syncthreads();
    }
}

}  // namespace mmw
