// Small kernels around the FFT chain: synthetic input generation in HBM, TDM de-interleave,
// |.|, range-profile averaging and the per-detection angle FFT + argmax of the point cloud.
#pragma once
#include "mmw_ctx.h"

namespace mmw {

// ------------------------------------------------------------------ counter-based RNG
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(unsigned long long h) {           // (0, 1]
    return ((float)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);
}

constexpr int SYNTH_MAX_TARGETS = 16;

// One thread per cube element.  Target parameters are re-derived per thread from (seed0 + frame, k)
// so no parameter buffer is needed; values are rounded to integers like a 16-bit ADC.
__global__ __launch_bounds__(256) void k_synth(float2 *cubes, long total, int V, int S, int C,
                                                unsigned long long seed0, int num_targets, float sigma) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid % C);
    const int s = (int)((gid / C) % S);
    const int v = (int)((gid / ((long)C * S)) % V);
    const unsigned long long f = (unsigned long long)(gid / ((long)C * S * V));
    const unsigned long long fs = mix64(seed0 + f);
    float re = 0.f, im = 0.f;
    for (int k = 0; k < num_targets; ++k) {
        const unsigned long long b = mix64(fs ^ (0x1000ull * (k + 1)));
        const float f_r = 0.02f + 0.43f * u01(mix64(b + 1));
        const float f_d = -0.45f + 0.90f * u01(mix64(b + 2));
        const float f_a = -0.40f + 0.80f * u01(mix64(b + 3));
        const float amp = 20.f + 380.f * u01(mix64(b + 4));
        const float phi = u01(mix64(b + 5));
        float ph = f_r * (float)s;
        ph -= floorf(ph);
        float t = f_d * (float)c;
        ph += t - floorf(t);
        t = f_a * (float)v;
        ph += t - floorf(t) + phi;
        ph -= floorf(ph);
        float sn, cs;
        sincospif(2.0f * ph, &sn, &cs);
        re += amp * cs;
        im += amp * sn;
    }
    // noise keyed by (frame seed, element index inside the frame): a frame's bytes do not depend on where in a batch
    // (or on which device of a sharded batch) it is generated
    const unsigned long long e = (unsigned long long)(gid - (long)f * ((long)C * S * V));
    const unsigned long long h = mix64(fs ^ mix64(e * 2 + 0x51ull));
    const float u1 = u01(h), u2 = u01(mix64(h + 7));
    const float rad = sigma * sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    cubes[gid] = make_float2(rintf(re + rad * cs), rintf(im + rad * sn));
}

// raw[F][num_rx][S][num_tx*loops] -> virt[F][num_tx*num_rx][S][loops]; virtual antenna t*num_rx + r
// takes every num_tx-th chirp starting at t (processors/virtual_array_reformater.py:53-63).
__global__ __launch_bounds__(256) void k_reformat(const float2 *raw, float2 *virt, long total,
                                                   int num_rx, int num_tx, int S, int loops) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int l = (int)(gid % loops);
    const int s = (int)((gid / loops) % S);
    const int va = (int)((gid / ((long)loops * S)) % (num_rx * num_tx));
    const long f = gid / ((long)loops * S * num_rx * num_tx);
    const int t = va / num_rx, r = va % num_rx;
    virt[gid] = raw[((f * num_rx + r) * S + s) * ((long)num_tx * loops) + (long)l * num_tx + t];
}

// The same de-interleave from int16 I/Q samples: raw[F][num_rx][S][num_tx*loops][2] (I, Q) -> complex64 virtual-array cube.
// NO UPSTREAM ORACLE for the sample layout: the reference gets its cubes from the absent cpsl_datasets reader;
// this is "the raw cube mmw_*_raw accepts, with int16 I/Q instead of complex64" -- half the bytes over PCIe and HBM.
__global__ __launch_bounds__(256) void k_reformat_i16(const short2 *raw, float2 *virt, long total, int num_rx, int num_tx, int S,
                                                       int loops) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int l = (int)(gid % loops);
    const int s = (int)((gid / loops) % S);
    const int va = (int)((gid / ((long)loops * S)) % (num_rx * num_tx));
    const long f = gid / ((long)loops * S * num_rx * num_tx);
    const int t = va / num_rx, r = va % num_rx;
    const short2 v = raw[((f * num_rx + r) * S + s) * ((long)num_tx * loops) + (long)l * num_tx + t];
    virt[gid] = make_float2((float)v.x, (float)v.y);
}

__global__ __launch_bounds__(256) void k_abs_c64(const float2 *in, float *out, size_t n) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid < n) {
        const float2 v = in[gid];
        out[gid] = hypotf(v.x, v.y);
    }
}

// float32 -> float64, element by element (complex arrays as pairs): the reference hands back complex128 / float64 arrays,
// and widening 16 MB on one host core (numpy astype, ~5 ms) costs more than moving twice the bytes over PCIe
__global__ __launch_bounds__(256) void k_widen_f32_f64(const float *__restrict__ in, double *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (double)in[i];
}

// out[f][s] = mean_v |spec[f][v][s]|   (processors/range_resp.py:55-57; np.mean over axis 0 adds the
// antenna rows in order, which the loop reproduces)
template <typename T>
__global__ __launch_bounds__(256) void k_mean_abs_over_v(const cplx<T> *spec, T *out, int F, int V, int S) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)F * S) return;
    const int s = (int)(gid % S);
    const long f = gid / S;
    T acc = (T)0;
    for (int v = 0; v < V; ++v) {
        const cplx<T> x = spec[(f * V + v) * S + s];
        if constexpr (sizeof(T) == 8) acc += hypot(x.x, x.y);
        else acc += hypotf(x.x, x.y);
    }
    out[gid] = acc / (T)V;
}

// ------------------------------------------------------------------ in-situ HBM ceiling (diagnostics)
// mode 0: dst = src (16 B/lane copy)   1: dst = const (write only)   2: read only (sum kept live)
// mode 3: write only, non-temporal   4: write only, 4 stores in flight per lane   5: copy, 4 in flight
typedef float diag_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_diag_membw(const diag_f4 *src, diag_f4 *dst, size_t n_vec, int mode) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    diag_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const diag_f4 k = {1.f, 2.f, 3.f, 4.f};
    if (mode == 6 || mode == 7) {
        // the angle kernel's store pattern without its arithmetic: every thread owns 16 B of a 256-KiB "plane"
        // and writes the same offset of 64 consecutive planes (mode 7: 4 planes per pass, 16 passes)
        const size_t plane_vec = 16384, frame_vec = 64 * plane_vec;
        const size_t frames = n_vec / frame_vec;
        for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < frames * plane_vec; w += stride) {
            const size_t f = w / plane_vec, o = w % plane_vec;
            diag_f4 *base = dst + f * frame_vec + o;
#pragma unroll 8
            for (int a = 0; a < 64; ++a) __builtin_nontemporal_store(k, base + (size_t)a * plane_vec);
        }
        return;
    }
    if (mode == 4 || mode == 5) {
        for (size_t i = i0; i + 3 * stride < n_vec; i += 4 * stride) {
            if (mode == 4) {
                dst[i] = k; dst[i + stride] = k; dst[i + 2 * stride] = k; dst[i + 3 * stride] = k;
            } else {
                const diag_f4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
                dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
            }
        }
        return;
    }
    for (size_t i = i0; i < n_vec; i += stride) {
        if (mode == 0) dst[i] = src[i];
        else if (mode == 1) dst[i] = k;
        else if (mode == 3) __builtin_nontemporal_store(k, dst + i);
        else acc += src[i];
    }
    if (mode == 2 && acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[0] = acc;
}

// out[f][c][a] = mean over range rows s in [s_lo, s_hi) of mag[f][a][s][c]
// (DopplerAzimuthProcessor.process: np.mean(resp, axis=0), processors/doppler_azimuth_resp.py:489)
__global__ __launch_bounds__(256) void k_mean_over_range(const float *mag, float *out, int F, int A, int S, int C,
                                                          int s_lo, int s_hi) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)F * A * C) return;
    const int c = (int)(gid % C);
    const int a = (int)((gid / C) % A);
    const long f = gid / ((long)C * A);
    const float *src = mag + ((f * A + a) * S) * (long)C + c;
    float acc = 0.f;
    for (int s = s_lo; s < s_hi; ++s) acc += src[(long)s * C];
    out[(f * C + c) * A + a] = acc / (float)(s_hi - s_lo);
}

// out[f][s][i] = mag[f][ang_idx[i]][s][vel_idx[i]]   (perform_dbs_sharpen,
// processors/range_angle_resp_dbs_enhanced.py:216-263: one [angle bin, :, Doppler bin] column per output angle)
__global__ __launch_bounds__(256) void k_dbs_gather(const float *mag, const int *ang_idx, const int *vel_idx, float *out,
                                                     int F, int A, int S, int C, int n_out) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)F * S * n_out) return;
    const int i = (int)(gid % n_out);
    const int s = (int)((gid / n_out) % S);
    const long f = gid / ((long)n_out * S);
    out[gid] = mag[((f * A + ang_idx[i]) * S + s) * (long)C + vel_idx[i]];
}

// Zoom DFT: out[row][k] = sum_n win[n] x[row][n] exp(-j 2 pi n (f0 + k df)), frequencies in cycles/sample.
// This is what scipy.signal.ZoomFFT evaluates (by Bluestein) for RangeProcessor.zoom_fft
// (processors/range_resp.py:59-102); n and m are a few hundred, so the direct sum in float64 phase is enough.
// One workgroup per row, the windowed row staged in LDS, thread k owns output bins k, k+256, ...
__global__ __launch_bounds__(256) void k_zoom_dft(const float2 *x, long row_stride, long elem_stride, const float *win,
                                                   float2 *out, int n, int m, double f0, double df) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    const float2 *src = x + (long)blockIdx.x * row_stride;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float2 v = src[(long)i * elem_stride];
        const float w = win ? win[i] : 1.f;
        xs[i] = make_float2(v.x * w, v.y * w);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += 256) {
        const double f = f0 + (double)k * df;
        double re = 0.0, im = 0.0;
        for (int i = 0; i < n; ++i) {
            double turns = f * (double)i;
            turns -= rint(turns);
            double sn, cs;
            sincospi(-2.0 * turns, &sn, &cs);
            re += (double)xs[i].x * cs - (double)xs[i].y * sn;
            im += (double)xs[i].x * sn + (double)xs[i].y * cs;
        }
        out[(long)blockIdx.x * m + k] = make_float2((float)re, (float)im);
    }
}

// Doppler zoom transform of DopplerAzimuthProcessor.zoom_fft (processors/doppler_azimuth_resp.py:130-163):
// k_zoom_table  Z[i][k] = exp(-j 2 pi i f[k]) with the phase reduced in float64 (f in cycles per chirp; NaN marks
//               a bin the reference fills with zeros, :267/:285);
// k_zoom_rows   out[row][k] = sum_{i<n} x[row][i] Z[i][k] for the rows (frame*antenna, kept range bin) of the
//               range-FFT cube: RB rows staged in LDS per workgroup, thread k owns one zoom bin.
__global__ __launch_bounds__(256) void k_zoom_table(const double *freq, float2 *Z, int n, int m) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)n * m) return;
    const int k = (int)(gid % m), i = (int)(gid / m);
    const double f = freq[k];
    float2 z = make_float2(0.f, 0.f);
    if (f == f) {
        double turns = f * (double)i;
        turns -= rint(turns);
        double sn, cs;
        sincospi(-2.0 * turns, &sn, &cs);
        z = make_float2((float)cs, (float)sn);
    }
    Z[gid] = z;
}

template <int RB>
__global__ __launch_bounds__(256) void k_zoom_rows(const float2 *x, const float2 *Z, float2 *out, int S, int C, int s_lo,
                                                    int s_keep, int n, int m, long total_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);           // [RB][n]
    const long row0 = (long)blockIdx.x * RB;
    for (int e = threadIdx.x; e < RB * n; e += 256) {
        const int r = e / n, i = e - r * n;
        const long row = row0 + r;
        float2 v = make_float2(0.f, 0.f);
        if (row < total_rows) {
            const long fv = row / s_keep;
            const int s = s_lo + (int)(row - fv * s_keep);
            v = x[(fv * S + s) * (long)C + i];
        }
        xs[e] = v;
    }
    __syncthreads();
    const int k = blockIdx.y * 256 + threadIdx.x;
    if (k >= m) return;
    float2 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = make_float2(0.f, 0.f);
    for (int i = 0; i < n; ++i) {
        const float2 z = Z[(long)i * m + k];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float2 v = xs[r * n + i];
            acc[r].x = fmaf(v.x, z.x, fmaf(-v.y, z.y, acc[r].x));
            acc[r].y = fmaf(v.x, z.y, fmaf(v.y, z.x, acc[r].y));
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
        if (row0 + r < total_rows) out[(row0 + r) * m + k] = acc[r];
}

constexpr int MAX_ANT = 32;
struct AntList {
    int n;
    int idx[MAX_ANT];
};

// np.argmax order on magnitudes: a NaN beats everything that is not a NaN, the FIRST maximum wins
template <typename T> __device__ __forceinline__ bool mag_gt(T a, T b) {
    if (a != a) return !(b != b);
    if (b != b) return false;
    return a > b;
}
template <typename T> __device__ __forceinline__ bool mag_better(T a, int ia, T b, int ib) {   // (a, ia) ahead of (b, ib)
    return mag_gt(a, b) || (!mag_gt(b, a) && ia < ib);
}

// l1[plane] = sum over the plane of hann(S)[s] hann(C)[c] (|re| + |im|)  >=  sum |w x|: the scale of the rounding-error
// bound of the float32 range-Doppler cell values (see k_angle_argmax).  One workgroup per plane.
__global__ __launch_bounds__(256) void k_plane_l1(const float2 *__restrict__ in, float *__restrict__ l1, int S, int C,
                                                   const float *__restrict__ ws, const float *__restrict__ wc) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float2 *src = in + (long)blockIdx.x * S * C;
    float acc = 0.f;
    if ((C & 1) == 0) {             // one wave per row, 16-B loads
        const f4 *src4 = reinterpret_cast<const f4 *>(src);
        const int C2 = C >> 1;
        for (int s = wave; s < S; s += 4) {
            float row = 0.f;
            for (int p = lane; p < C2; p += 64) {
                const f4 v = __builtin_nontemporal_load(src4 + (long)s * C2 + p);
                row += wc[2 * p] * (fabsf(v.x) + fabsf(v.y)) + wc[2 * p + 1] * (fabsf(v.z) + fabsf(v.w));
            }
            acc += ws[s] * row;
        }
    } else {
        for (int s = wave; s < S; s += 4) {
            float row = 0.f;
            for (int c = lane; c < C; c += 64) {
                const float2 v = src[(long)s * C + c];
                row += wc[c] * (fabsf(v.x) + fabsf(v.y));
            }
            acc += ws[s] * row;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) l1[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// Detections whose float32 argmax is not provably the float64 one are appended here and re-evaluated in float64.
struct ArgmaxRefine {
    const float *l1;        // [F][V] from k_plane_l1; nullptr: no refinement (plain float32 argmax)
    int *n_flag;            // number of flagged detections (may exceed list_cap)
    int *list;              // f * cap + det of the flagged detections
    int list_cap;
    float k_fft, k_ang;     // error-bound constants (see below)
    int *flagpos;           // optional [F][cap]: 1 + list position of a flagged detection (the dense refinement finds a frame's
    int dense_cap;          // flagged detections through it: mmw_cells64.h); positions from dense_cap on are not recorded
    int tag;                // ORed into the list entries (REFINE_SECOND: the second list of a call that refines two at once)
    int *n_tagged;          // optional: counts the flagged evaluations of a tagged list
};

// One wave per detection: gather rd[f][ant[i]][r][v], lane k evaluates angle bins k, k+64, ... of the
// zero-padded A-point DFT, first-max argmax over the (optionally fftshifted) response.
// (processors/point_cloud_generator.py:168-214; np.argmax returns the FIRST maximum.)
// The reference does this in complex128 on a complex128 range-Doppler cube.  Here the cube is float32, so the kernel also
// bounds how far its magnitudes can be from the float64 ones:
//   |rd32 - rd64| <= k_fft * l1(plane)   (each RD cell is a sum of S*C products w x W with <= log2(S) + log2(C) + 2
//                                          roundings on the way: (log2 S + log2 C + 2) eps sum|w x| <= 17 eps l1 for
//                                          256 x 128; k_fft = 32 eps leaves a factor ~2)
//   |m32[k] - m64[k]| <= B = sum_i k_fft l1_i + k_ang sum_i |x_i|   (n_ant-term float32 sum + hypotf)
// and flags the detection unless best - second > 2 B: the float64 argmax of an unflagged detection is the one found.
// NA: antennas held in registers (the list is padded to it): 4 / 8 / 16 / 32 -- the gathers are cold, strided 8-byte
// loads, so the kernel lives on the number of waves in flight, i.e. on a small register footprint.
// gather of one detection's cells: x[i] = rd[f][ants.idx[i]][r][v] (zero beyond the list), sum_abs = sum |re| + |im|
template <int NA>
__device__ __forceinline__ void argmax_gather(const float2 *rd, long f, int V, int S, int C, int r, int v, const AntList &ants,
                                              float2 (&x)[NA], float &sum_abs) {
    sum_abs = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        // unconditional (entries of the list past n are antenna 0): a guarded load would be followed by its own wait
        x[i] = rd[((f * V + ants.idx[i]) * S + r) * C + v];
        if (i >= ants.n) x[i] = make_float2(0.f, 0.f);
        if (i < ants.n) sum_abs += fabsf(x[i].x) + fabsf(x[i].y);
    }
}

// one wave: zero-padded A-point DFT of the n cells, |.|, wave-wide FIRST maximum (wb at bin wi) and the largest
// magnitude among all other bins (ws2)
template <int NA>
__device__ __forceinline__ void argmax_eval(const float2 (&x)[NA], int n, int A, int shift, const float2 *twA, int lane,
                                            float &wb, int &wi, float &ws2, float &lane_re, float &lane_im, int &lane_idx) {
    const float NEG = -__builtin_huge_valf();
    float best = NEG, second = NEG;
    int best_idx = 0x7fffffff;
    lane_re = lane_im = 0.f;
    for (int k = lane; k < A; k += 64) {
        float re = 0.f, im = 0.f;
        int t = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i < n) {
                const float2 w = twA[t];
                re += x[i].x * w.x - x[i].y * w.y;
                im += x[i].x * w.y + x[i].y * w.x;
                t += k;
                if (t >= A) t -= A;
            }
        }
        const float m = hypotf(re, im);
        const int kk = shift ? (k + A / 2) % A : k;
        if (best_idx == 0x7fffffff || mag_better(m, kk, best, best_idx)) {
            if (best_idx != 0x7fffffff) second = best;
            best = m;
            best_idx = kk;
            lane_re = re;
            lane_im = im;
        } else if (mag_gt(m, second)) second = m;
    }
    lane_idx = best_idx;
    // wave-wide first maximum, then the largest magnitude among everything else
    wb = best;
    wi = best_idx;
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(wb, d, 64);
        const int oi = __shfl_xor(wi, d, 64);
        if (oi != 0x7fffffff && (wi == 0x7fffffff || mag_better(ob, oi, wb, wi))) {
            wb = ob;
            wi = oi;
        }
    }
    ws2 = (best_idx == wi) ? second : best;
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(ws2, d, 64);
        if (mag_gt(o, ws2)) ws2 = o;
    }
}

// Second pass of the certainty test (lists of at most 8 antennas; same form as detect_argmax_list in mmw_detect.h): the
// errors of two bins of one detection are the SAME per-antenna cell errors e_i seen through two steering vectors, so
//   |(m1 - mk)_32 - (m1 - mk)_64| <= sum_i e_i |conj(u1) W^(i k1) - conj(uk) W^(i k)| + Be^2 / (2 min(m1, mk)) + 2 c_ang,
// far below the independent-errors 2 (Be + c_ang) for neighbouring bins.  Every bin other than the winner is tested;
// returns true when some bin is not provably below the winner.
template <int NA>
__device__ __forceinline__ bool argmax_pairwise(const float2 (&x)[NA], const float (&e)[NA], int A, int shift, const float2 *twA,
                                                int lane, int wi, float m1, float re1, float im1, float be, float c_ang) {
    const float inv1 = 1.f / m1, u1x = re1 * inv1, u1y = im1 * inv1, two_b = 2.f * (be + c_ang);
    const int k1 = shift ? (wi + A - A / 2) % A : wi;
    float2 t1[NA];                      // conj(u_1) W^(i k_1)
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float2 w = twA[t];
            t1[i] = make_float2(u1x * w.x + u1y * w.y, u1x * w.y - u1y * w.x);
            t += k1;
            if (t >= A) t -= A;
        }
    }
    bool bad = false;
    for (int k = lane; k < A; k += 64) {
        const int kk = shift ? (k + A / 2) % A : k;
        if (kk == wi) continue;
        float2 w[NA];
        int t = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            w[i] = twA[t];
            t += k;
            if (t >= A) t -= A;
        }
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i) {      // (cells past the list are zero: same sums as the first pass)
            re += x[i].x * w[i].x - x[i].y * w[i].y;
            im += x[i].x * w[i].y + x[i].y * w[i].x;
        }
        const float m = hypotf(re, im), margin = m1 - m;
        bool ok = margin > two_b;
        if (!ok && m > 0.f) {
            const float inv = 1.f / m, ux = re * inv, uy = im * inv;
            float lin = 0.f;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const float dx = t1[i].x - (ux * w[i].x + uy * w[i].y), dy = t1[i].y - (ux * w[i].y - uy * w[i].x);
                lin += e[i] * __fsqrt_rn(fmaf(dx, dx, dy * dy));
            }
            ok = margin > 1.001f * (lin + be * be / (2.f * fminf(m1, m))) + 2.f * c_ang;
        }
        bad |= !ok;
    }
    return __ballot(bad) != 0ull;
}

template <int NA>
__global__ __launch_bounds__(256) void k_angle_argmax(const float2 *rd, const int32_t *dets, const int32_t *counts,
                                                       int32_t *out_idx, int V, int S, int C, int cap,
                                                       AntList ants, int A, int shift, const float2 *twA, ArgmaxRefine rf) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.y;
    int n_det = counts[f];
    if (n_det > cap) n_det = cap;
    constexpr bool PAIR = NA <= 8;      // longer lists keep the independent-errors test (more of them are refined)
    float sum_l1 = 0.f, e[PAIR ? NA : 1];
    if (rf.l1) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float l = i < ants.n ? rf.l1[(long)f * V + ants.idx[i]] : 0.f;
            sum_l1 += l;
            if constexpr (PAIR) e[i] = rf.k_fft * l;
        }
    }
    // grid.x covers the usual detection counts in one pass; a wave walks on for frames with more
    for (int det = blockIdx.x * 4 + wave; det < n_det; det += gridDim.x * 4) {
        const int r = dets[((long)f * cap + det) * 2], v = dets[((long)f * cap + det) * 2 + 1];
        float2 x[NA];
        float sum_abs, wb, ws2, lre, lim;
        int wi, lidx;
        argmax_gather<NA>(rd, f, V, S, C, r, v, ants, x, sum_abs);
        argmax_eval<NA>(x, ants.n, A, shift, twA, lane, wb, wi, ws2, lre, lim, lidx);
        bool flag = false;
        if (rf.l1) {
            const float be = rf.k_fft * sum_l1, c_ang = rf.k_ang * sum_abs;
            flag = !(wb - ws2 > 2.f * (be + c_ang));             // (also for NaN / inf magnitudes)
            if constexpr (PAIR) {
                if (flag && wb == wb && wb > 0.f && wb < __builtin_huge_valf()) {
                    const int l1 = __ffsll((long long)__ballot(lidx == wi)) - 1;
                    const float re1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lre), l1));
                    const float im1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lim), l1));
                    flag = argmax_pairwise<NA>(x, e, A, shift, twA, lane, wi, wb, re1, im1, be, c_ang);
                }
            }
        }
        if (lane == 0) {
            out_idx[(long)f * cap + det] = wi;
            if (flag) {
                const int pos = atomicAdd(rf.n_flag, 1);
                if (pos < rf.list_cap) rf.list[pos] = f * cap + det;
                if (rf.flagpos && pos < rf.dense_cap) rf.flagpos[(long)f * cap + det] = pos + 1;
            }
        }
    }
}

// float64 first-max argmax of the zero-padded A-point DFT of n_ant complex128 cells per row (np.fft.fft + np.abs +
// np.argmax of the reference, point_cloud_generator.py:187-206).  One wave per row.
__device__ __forceinline__ int argmax64_wave(const cplx<double> *X, int n_ant, int A, int shift, const cplx<double> *twA,
                                             int lane) {
    double best = 0.0;
    int best_idx = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        double re = 0.0, im = 0.0;
        int t = 0;
        for (int i = 0; i < n_ant; ++i) {
            const cplx<double> w = twA[t], xi = X[i];
            re += xi.x * w.x - xi.y * w.y;
            im += xi.x * w.y + xi.y * w.x;
            t += k;
            if (t >= A) t -= A;
        }
        const double m = hypot(re, im);
        const int kk = shift ? (k + A / 2) % A : k;
        if (best_idx == 0x7fffffff || mag_better(m, kk, best, best_idx)) {
            best = m;
            best_idx = kk;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const double ob = __shfl_xor(best, d, 64);
        const int oi = __shfl_xor(best_idx, d, 64);
        if (oi != 0x7fffffff && (best_idx == 0x7fffffff || mag_better(ob, oi, best, best_idx))) {
            best = ob;
            best_idx = oi;
        }
    }
    return best_idx;
}

__global__ __launch_bounds__(256) void k_argmax64_cells(const cplx<double> *cells, int32_t *out_idx, int n_rows, int n_ant,
                                                         int A, int shift, const cplx<double> *twA) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const int idx = argmax64_wave(cells + (long)row * n_ant, n_ant, A, shift, twA, lane);
    if (lane == 0) out_idx[row] = idx;
}

// Flagged detections: the range-Doppler cell of every listed antenna as the float64 2-D DFT coefficient of the windowed
// cube (direct S*C-term sum, table twiddles), then the float64 argmax.  The sum over a plane is cut into REFINE_PARTS
// slices, one workgroup each (a single workgroup per detection is latency-bound on its 32 trips to cold HBM lines):
//   k_argmax_refine_part   partial[e][part][i] for the first `n_split` flagged detections; a thread walks cells
//                          base + tid, + 256, ... keeping (s, c) and the two twiddle indices up to date incrementally and
//                          uses each cell's window * phase factor for all antennas of the list (phase tables: see
//                          refine_phases);
//   k_argmax_refine_finish adds the slices in a fixed order and runs the float64 argmax (one wave per detection);
//   k_argmax_refine_whole  whole planes in one workgroup: the (never yet seen) overflow beyond n_split detections.
// One flagged list may serve TWO antenna lists (mmw_detect_points: azimuth and elevation): an entry with bit 31 set
// belongs to the second (ants2 / shift2 / out_idx2).
#ifndef MMW_REFINE_PARTS
#define MMW_REFINE_PARTS 16
#endif
constexpr int REFINE_PARTS = MMW_REFINE_PARTS, REFINE_NA = 4;      // NA = 8 took 228 VGPRs: two workgroups per CU
constexpr int REFINE_SECOND = (int)0x80000000;

struct RefineArgs {
    const float2 *cubes;
    const int32_t *dets;
    const int *n_flag, *list;
    int list_cap;
    int32_t *out_idx, *out_idx2;
    int V, S, C, cap;
    AntList ants, ants2;
    int A, shift, shift2;
    const double *ws, *wc;
    const cplx<double> *twS, *twC, *twA;
    cplx<double> *partial;      // [n_split][parts][max(ants.n, ants2.n)]
    int n_split;
    int parts;                  // slices a plane sum is cut into (<= REFINE_PARTS): few when many detections are expected
    int dense_min, dense_cap;   // dense_min > 0: with *n_flag >= dense_min the first dense_cap entries belong to k_cells64
                                // (mmw_cells64.h) and these kernels take only what lies beyond them
};
// the entries [lo, n) of the flagged list the direct kernels own
__device__ __forceinline__ int refine_first(const RefineArgs &a, int n) {
    return a.dense_min > 0 && n >= a.dense_min ? (a.dense_cap < n ? a.dense_cap : n) : 0;
}

// Per flagged detection the factor of cell (s, c) splits into a range part and a Doppler part,
//   w_s(s) W_S^(r s)  *  w_c(c) W_C^(kd c),
// built once per task into two LDS tables (phS[S], phC[C]); the plane loop then reads phS[s] (one address per wave
// row: broadcast) and phC[c] (consecutive lanes, consecutive entries).  Gathering W_C^(kd c mod C) per cell instead put
// up to 64 lanes on one LDS bank (kd a multiple of 32) and the slowest task set the kernel's time.
inline size_t refine_tabs_lds(int S, int C) { return ((size_t)S + C) * 16 + 2 * MAX_ANT * sizeof(int); }

struct RefineTabs {
    cplx<double> *phS, *phC;
    int *ant;                   // ant[0..32) first list, ant[32..64) second
};
__device__ __forceinline__ RefineTabs refine_tabs(const RefineArgs &a, char *smem, int tid) {
    RefineTabs t;
    t.phS = reinterpret_cast<cplx<double> *>(smem);
    t.phC = t.phS + a.S;
    t.ant = reinterpret_cast<int *>(t.phC + a.C);
    if (tid == 0) {
        static_for<MAX_ANT>([&](auto I) {
            constexpr int i = decltype(I)::value;
            t.ant[i] = a.ants.idx[i];
            t.ant[MAX_ANT + i] = a.ants2.idx[i];
        });
    }
    return t;
}
// (every thread of the workgroup; ends with a barrier)
__device__ __forceinline__ void refine_phases(const RefineArgs &a, const RefineTabs &t, int r, int kd, int tid, int nt) {
    __syncthreads();            // the previous task's readers are done
    for (int s = tid; s < a.S; s += nt) t.phS[s] = a.twS[(int)(((long)r * s) % a.S)] * a.ws[s];
    for (int c = tid; c < a.C; c += nt) t.phC[c] = a.twC[(int)(((long)kd * c) % a.C)] * a.wc[c];
    __syncthreads();
}

// sum over cells [cell_lo, cell_hi) of plane (f, ant[a0 + i]) * window * phase, i < NA (antennas beyond n_ant: zero)
template <int NA>
__device__ __forceinline__ void refine_accumulate(const RefineArgs &a, const RefineTabs &t, const int *ant, int n_ant, int f, int a0,
                                                  long cell_lo, long cell_hi, int tid, cplx<double> (&acc)[NA]) {
    const int S = a.S, C = a.C;
    const long plane_cells = (long)S * C;
    const int ds = 256 / C, dc = 256 - ds * C;         // advancing a cell index by 256: c += dc (carry into s), s += ds
    const float2 *plane[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        acc[i] = cplx<double>{0.0, 0.0};
        plane[i] = a.cubes + ((long)f * a.V + (a0 + i < n_ant ? ant[a0 + i] : 0)) * plane_cells;
    }
    const long first = cell_lo + tid;
    int s = (int)(first / C), c = (int)(first - (long)s * C);
    if (s >= S) s = 0;
    constexpr int U = 4;        // four cells per trip: all of their (cold) samples are requested before any is used
    for (long cell0 = first; cell0 < cell_hi; cell0 += 256 * U) {
        float2 xv[U][NA];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                // unconditional, clamped (a guarded load is followed by its own wait: every one of these 16 loads cost a
                // full trip to HBM); cells past the end are not used, antennas past the list read plane 0 and are dropped
                const long cell = cell0 + 256 * u;
                xv[u][i] = plane[i][cell < cell_hi ? cell : cell_hi - 1];
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (cell0 + 256 * u < cell_hi) {
                const cplx<double> ph = cmul(t.phS[s], t.phC[c]);
#pragma unroll
                for (int i = 0; i < NA; ++i) acc[i] = acc[i] + cmul(cplx<double>{(double)xv[u][i].x, (double)xv[u][i].y}, ph);
            }
            c += dc;
            if (c >= C) {
                c -= C;
                ++s;
            }
            s += ds;
            if (s >= S) s = 0;          // past the plane (guarded above): keep the table index in range
        }
    }
}

// block-wide sum of acc[i] in a fixed order -> out[i] (thread i writes), `red` = [4][NA] LDS words
template <int NA>
__device__ __forceinline__ void refine_reduce(cplx<double> (&acc)[NA], cplx<double> (*red)[NA], int tid, cplx<double> *out, int n_out) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        for (int d = 32; d >= 1; d >>= 1) {
            acc[i].x += __shfl_xor(acc[i].x, d, 64);
            acc[i].y += __shfl_xor(acc[i].y, d, 64);
        }
        if (lane == 0) red[wave][i] = acc[i];
    }
    __syncthreads();
    if (tid < NA && tid < n_out) out[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
}

struct RefineEntry {
    int f, det, r, kd, which, n_ant;
};
__device__ __forceinline__ RefineEntry refine_entry(const RefineArgs &a, int e) {
    RefineEntry q;
    int id = a.list[e];
    q.which = id < 0 ? 1 : 0;
    id &= 0x7fffffff;
    q.n_ant = q.which ? a.ants2.n : a.ants.n;
    q.f = id / a.cap;
    q.det = id - q.f * a.cap;
    q.r = a.dets[((long)q.f * a.cap + q.det) * 2];
    int k = a.dets[((long)q.f * a.cap + q.det) * 2 + 1] - a.C / 2;     // FFT bin behind the fftshifted Doppler index
    if (k < 0) k += a.C;
    q.kd = k;
    return q;
}
__device__ __forceinline__ int refine_stride(const RefineArgs &a) { return a.ants.n > a.ants2.n ? a.ants.n : a.ants2.n; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_argmax_refine_part(RefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ cplx<double> red[4][REFINE_NA];
    int n = *a.n_flag;
    if (n > a.list_cap) n = a.list_cap;
    const int first = refine_first(a, n);
    if (n > a.n_split) n = a.n_split;
    if (first + (int)blockIdx.y >= n) return;
    const int part = blockIdx.x, tid = threadIdx.x;
    const RefineTabs t = refine_tabs(a, smem, tid);
    const long plane_cells = (long)a.S * a.C;
    const long per = ((plane_cells + a.parts - 1) / a.parts + 255) / 256 * 256;
    const long lo = (long)part * per, hi = lo + per < plane_cells ? lo + per : plane_cells;
    const int stride = refine_stride(a);
    for (int e = first + blockIdx.y; e < n; e += gridDim.y) {
        const RefineEntry q = refine_entry(a, e);
        refine_phases(a, t, q.r, q.kd, tid, 256);
        const int *al = t.ant + (q.which ? MAX_ANT : 0);
        for (int a0 = 0; a0 < q.n_ant; a0 += REFINE_NA) {
            cplx<double> acc[REFINE_NA];
            refine_accumulate<REFINE_NA>(a, t, al, q.n_ant, q.f, a0, lo < hi ? lo : hi, hi, tid, acc);
            refine_reduce<REFINE_NA>(acc, red, tid, a.partial + ((long)e * a.parts + part) * stride + a0, q.n_ant - a0);
        }
    }
}

__global__ __launch_bounds__(256) void k_argmax_refine_finish(RefineArgs a) {
    __shared__ cplx<double> X[4][MAX_ANT];
    int n = *a.n_flag;
    if (n > a.list_cap) n = a.list_cap;
    const int first = refine_first(a, n);
    if (n > a.n_split) n = a.n_split;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = refine_stride(a);
    for (int e0 = first + blockIdx.x * 4; e0 < n; e0 += gridDim.x * 4) {
        const int e = e0 + wave;
        RefineEntry q{};
        if (e < n) {
            q = refine_entry(a, e);
            for (int i = lane; i < q.n_ant; i += 64) {
                cplx<double> s = cplx<double>{0.0, 0.0};
                for (int p = 0; p < a.parts; ++p) s = s + a.partial[((long)e * a.parts + p) * stride + i];
                X[wave][i] = s;
            }
        }
        __syncthreads();
        if (e < n) {
            const int idx = argmax64_wave(X[wave], q.n_ant, a.A, q.which ? a.shift2 : a.shift, a.twA, lane);
            if (lane == 0) (q.which ? a.out_idx2 : a.out_idx)[(long)q.f * a.cap + q.det] = idx;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_argmax_refine_whole(RefineArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ cplx<double> red[4][REFINE_NA];
    __shared__ cplx<double> X[MAX_ANT];
    int n = *a.n_flag;
    if (n > a.list_cap) n = a.list_cap;
    const int tid = threadIdx.x;
    const int begin = a.n_split > refine_first(a, n) ? a.n_split : refine_first(a, n);
    if (begin + (int)blockIdx.x >= n) return;
    const RefineTabs t = refine_tabs(a, smem, tid);
    for (int e = begin + blockIdx.x; e < n; e += gridDim.x) {
        const RefineEntry q = refine_entry(a, e);
        refine_phases(a, t, q.r, q.kd, tid, 256);
        const int *al = t.ant + (q.which ? MAX_ANT : 0);
        for (int a0 = 0; a0 < q.n_ant; a0 += REFINE_NA) {
            cplx<double> acc[REFINE_NA];
            refine_accumulate<REFINE_NA>(a, t, al, q.n_ant, q.f, a0, 0, (long)a.S * a.C, tid, acc);
            refine_reduce<REFINE_NA>(acc, red, tid, X + a0, q.n_ant - a0);
        }
        if (tid < 64) {
            const int idx = argmax64_wave(X, q.n_ant, a.A, q.which ? a.shift2 : a.shift, a.twA, tid);
            if (tid == 0) (q.which ? a.out_idx2 : a.out_idx)[(long)q.f * a.cap + q.det] = idx;
        }
        __syncthreads();
    }
}

}  // namespace mmw
