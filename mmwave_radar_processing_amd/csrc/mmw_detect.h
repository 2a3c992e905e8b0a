// Fused per-frame detection stage of the batch pipeline (BASELINE configs[2]): everything behind the range-Doppler
// kernel in ONE launch, one 512-thread workgroup per frame, whatever the plane's size: the plane streams through the LDS in
// bands of rows (detect_screen_frame).  (Round 3 held the whole plane's magnitudes in the LDS and cut larger planes into row
// tiles with a last-finisher protocol; both are gone.)
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
//   d = X32 - T':   d >  band  => detection for certain,   d <= -band => certainly none,   band = (1 + alpha) B + 3e-6 (X32 + T')
//
// (the 3e-6 term covers the float32 evaluation of |.| = sqrt(fma(re, re, im * im)) -- three roundings, under 2e-7 relative for
// planes whose L1 norm is within [1e-10, 1e18], anything else is handed back -- and the float32 window sums: they add
// non-negative numbers only -- training rows of every window column plus guard rows of the columns outside the guard --
// so at most 33 roundings of 2^-24 relative each).  Cells inside the band -- one frame in ~13 of
// the synthetic workload has one -- are decided EXACTLY: k_cfar_cell_exact evaluates the (2 hr + 1) x (2 hd + 1) window
// of float64 range-Doppler magnitudes around the cell as direct float64 DFT sums of the input cube and applies the
// reference's float64 rule.  Frames with such cells skip compaction in the screening kernel and are finished by
// k_detect_finish once their bit masks are complete.  A frame whose input is not finite, or that has more undecided cells
// than the list holds, gets counts[f] = -1: the caller runs it through the float64 path (mmw_detect_batch).
//
// Compaction is ordered (np.where order, base.py:229-230); the per-detection angle argmax (azimuth and elevation lists,
// float32 with the error bound of k_angle_argmax, flagged near-ties re-evaluated in float64 by the k_argmax_refine_* kernels)
// runs in the same workgroup -- or, with 64 angle bins (the default: "late argmax"), as launches of their own in the call's
// tail over the cells this workgroup copied into the context's record list (DetectArgs::rec_cells, k_angle_argmax_recs).
#pragma once
#include "mmw_ctx.h"
#include "mmw_misc.h"
#include "mmw_fft_fused.h"

namespace mmw {

constexpr int DET_NT = 512;              // threads of the per-frame workgroup: TWO workgroups per CU (70 KB of LDS each at 256 x 128, 125
                                         // VGPRs): one's barriers and load waits are the other's issue slots -- 8 % faster than one of 1024
constexpr int DET_MAX_ANT = 8;           // antennas per list inside the fused kernels (cells, twiddles and error scales of a
                                         // list live in registers; lists of 9+ antennas take mmw_angle_argmax_exact)
constexpr int DET_LATE_MAX_ANT = 16;     // ... per list with the late argmax (64 angle bins: k_angle_argmax_recs, one lane per detection,
                                         // 64 / N register FFTs of length N = 4, 8 or 16)
constexpr int DET_LIST2 = 16;            // first lane of the second list's cells
constexpr int DET_SPEC = 32;             // undecided cells a frame may carry speculatively (more: the frame is handed back)
enum { DST_UNDECIDED = 1, DST_OVERFLOW = 2, DST_DEGENERATE = 4 };
enum { DCTL_FLAG_FRAMES = 0, DCTL_CELLS = 1, DCTL_FALLBACK = 2, DCTL_RECS = 3, DCTL_ARGMAX = 16, DCTL_EL = 32, DCTL_WORDS = 64 };   // ARGMAX: flagged
                                                    // evaluations of both lists (the refinement list's length), EL: those of the second

struct DetAnt {            // antenna list of one angle estimate (n == 0: not wanted)
    int n;
    int idx[DET_LATE_MAX_ANT];
};

struct DetectArgs {
    const float2 *rd;          // [F][V][S][C] float32 range-Doppler cube
    const float *l1;           // [F][V] plane L1 norms (error-bound scale)
    float *mag32;              // optional [F][S][C]: |RD| of antenna 0
    int32_t *dets, *counts;    // [F][cap][2], [F]
    int32_t *az_idx, *el_idx;  // [F][cap] each (nullptr with an empty list)
    int *ctl;                  // DCTL_* counters
    int *flag_frames;          // [F] frames with undecided cells
    int *spec;                 // [F][2 * DET_SPEC + 1]: n, then per undecided cell its index r * C + c and (from k_cfar_cell_exact) the decision
    int *cells;                // [cell_cap][2] undecided cells: (frame, r * C + c)
    int cell_cap;
    int V, S, C, cap, words, band_rows, band_pitch;   // band buffers: band_rows x band_pitch floats each
    int kind, tr, td, gr, gd, n_train, k_rank;
    double scale;
    float k_fft;               // ulps * 2^-24 of the RD kernel that ran
    DetAnt az, el;
    int A, shift_az, shift_el;
    const float2 *twA;
    ArgmaxRefine rf_az, rf_el;
    long long *clk;            // diagnostics (MMW_PHASE_CLOCKS=1): s_memtime at the phase boundaries of a mid-batch workgroup
    // overlapped schedule (k_detect_screen<..., true>): persistent workgroups draw frames from ctl[CTL_ANG_TICKET] and wait
    // until the producer (k_rd_fused_256x128_persist<DET>) has published all V planes of the frame in frame_cnt[f]
    unsigned *sy_ctl, *sy_frame_cnt;
    unsigned long long sy_timeout;      // s_memrealtime ticks a wait may last
    int sy_naps, n_frames;
    // late argmax (rec_cells != nullptr): instead of estimating the angles itself the frame's workgroup appends one RECORD per
    // detection (certain and speculative) to a flat list -- rec_slot[rec] = f * cap + slot, rec_cells[rec][n_az + n_el] = its
    // range-Doppler cells, azimuth list first -- and copies the frame's plane norms into l1_copy; k_angle_argmax_recs works
    // from those (ctl[DCTL_RECS] = records so far)
    float2 *rec_cells;
    int32_t *rec_slot;
    float *l1_copy;
    int rec_cap;
};

__device__ __forceinline__ void det_mark(const DetectArgs &a, int i) {
    if (a.clk && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) a.clk[i] = (long long)__builtin_amdgcn_s_memtime();
}
// persistent form: phase i of the frame in hand ends now -- clk[i - 1] accumulates its clocks (workgroup 0; clk[6] = last mark)
// (diagnostics) clocks since the previous mark into clk[slot] (slots 8..15: inside the band loop)
__device__ __forceinline__ void det_lap(const DetectArgs &a, int slot) {
    if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long now = (long long)__builtin_amdgcn_s_memtime();
        a.clk[slot] += now - a.clk[6];
        a.clk[6] = now;
    }
}
__device__ __forceinline__ void det_mark_acc(const DetectArgs &a, int i) {
    if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long now = (long long)__builtin_amdgcn_s_memtime();
        if (i > 0) a.clk[i - 1] += now - a.clk[6];
        a.clk[6] = now;
    }
}

// exclusive prefix of v over the workgroup (thread order), total in *total; ws: 40 ints of LDS
__device__ __forceinline__ int block_excl_scan(int v, int *ws, int *total, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
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
        static_for<DET_LATE_MAX_ANT>([&](auto I) {
            constexpr int i = decltype(I)::value;
            tab[i] = a.az.idx[i];
            tab[DET_LIST2 + i] = a.el.idx[i];
        });
    }
}

// Wave-wide reductions on the DPP network (row rotations inside the rows of 16 lanes, then row_bcast:15 / row_bcast:31
// into the following rows: the total arrives in lane 63) -- a few clocks per step where a __shfl_xor butterfly is a
// dependent chain of ds_bpermute round trips (24 of them per detection and list were half of the argmax phase).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1, 0xf>(v, v));       // quad_perm [1,0,3,2]
    v = max(v, dpp_u32<0x4E, 0xf>(v, v));       // quad_perm [2,3,0,1]
    v = max(v, dpp_u32<0x124, 0xf>(v, v));      // row_ror:4
    v = max(v, dpp_u32<0x128, 0xf>(v, v));      // row_ror:8
    v = max(v, dpp_u32<0x142, 0xa>(v, v));      // row_bcast:15 into rows 1, 3
    v = max(v, dpp_u32<0x143, 0xc>(v, v));      // row_bcast:31 into rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {     // the minimum of the wave, in every lane's return value
    v = min(v, dpp_u32<0xB1, 0xf>(v, v));
    v = min(v, dpp_u32<0x4E, 0xf>(v, v));
    v = min(v, dpp_u32<0x124, 0xf>(v, v));
    v = min(v, dpp_u32<0x128, 0xf>(v, v));
    v = min(v, dpp_u32<0x142, 0xa>(v, v));
    v = min(v, dpp_u32<0x143, 0xc>(v, v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_sum_f32(float x) {
    auto f = [](unsigned u) { return __builtin_bit_cast(float, u); };
    auto u = [](float v) { return __builtin_bit_cast(unsigned, v); };
    x += f(dpp_u32<0xB1, 0xf>(0u, u(x)));
    x += f(dpp_u32<0x4E, 0xf>(0u, u(x)));
    x += f(dpp_u32<0x124, 0xf>(0u, u(x)));
    x += f(dpp_u32<0x128, 0xf>(0u, u(x)));
    x += f(dpp_u32<0x142, 0xa>(0u, u(x)));
    x += f(dpp_u32<0x143, 0xc>(0u, u(x)));
    return f((unsigned)__builtin_amdgcn_readlane((int)u(x), 63));
}
// order-preserving key of a magnitude (>= 0, or NaN: every NaN maps to the top key, as np.argmax treats them alike)
__device__ __forceinline__ unsigned mag_key(float m) { return m != m ? 0xffffffffu : __builtin_bit_cast(unsigned, m); }
__device__ __forceinline__ float key_mag(unsigned k) {
    return k == 0xffffffffu ? __builtin_nanf("") : __builtin_bit_cast(float, k);
}

// One wave, one detection, one antenna list whose cells sit in lanes [base, base + n) of xl (lanes up to base + NMAX hold
// zeros, l1v the L1 norm of the lane's antenna plane): zero-padded A-point DFT, |.|, first maximum -- k_angle_argmax's
// arithmetic with the cells read out of the lanes (v_readlane: wave-uniform values) instead of held n-fold in every
// lane's registers; tw = W_A^m table (LDS).
//
// Is the float32 winner the float64 one?  The cells carry errors |dx_i| <= e_i = k_fft l1_i (range-Doppler kernel), the
// float32 evaluation of a bin's sum S_k another |eta| <= c = k_ang sum |x_i|.  k_angle_argmax asks for
// m_1 - m_k > 2 (sum e_i + c), which treats the errors of the two bins as independent; they are not -- both come from
// the SAME dx_i.  With u = S / |S| and convexity of |.|:   |S + d| - |S| = Re(u* d) + r,  0 <= r <= |d|^2 / (2 |S|), so
//   (m_1 - m_k)(x + dx) - (m_1 - m_k)(x)  >=  - sum_i e_i |u_1* W^(i k_1) - u_k* W^(i k)|  -  Be^2 / (2 min(m_1, m_k)),   Be = sum e_i.
// For the neighbours of the peak of a plane wave the bracket is ~0.1 instead of 2, so a detection is flagged (and read
// again, whole planes, by the float64 kernels) an order of magnitude less often at the same e_i.  A bin passes if
// either form of the bound clears its margin; any bin that does not flags the detection.
// wk (one_bin: A == 64, the lane's only bin is k = lane): W_A^(i lane), i < DET_MAX_ANT, held in registers by the caller for
// the whole frame -- the table walk (an LDS read and three index instructions per antenna) was a third of an evaluation.
template <int NMAX>
__device__ __forceinline__ void detect_argmax_list(int A, int *ctl_el, const float2 *tw, float2 xl, float l1v, int base, int n,
                                                   int shift, int32_t *out_idx, const ArgmaxRefine &rf, long slot, int lane, int tag,
                                                   const float2 (&wk)[DET_MAX_ANT], bool one_bin) {
    float xr[NMAX], xi[NMAX], e[NMAX], be = 0.f, sum_abs = 0.f;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xl.x), base + i));
        xi[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xl.y), base + i));
        e[i] = rf.k_fft * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l1v), base + i));
        be += e[i];
        sum_abs += fabsf(xr[i]) + fabsf(xi[i]);
    }
    // the bin's sum S_k and the twiddles it used
    auto bin = [&](int k, float2 (&w)[NMAX], float &re, float &im) {
        if (one_bin) {
#pragma unroll
            for (int i = 0; i < NMAX; ++i) w[i] = wk[i];
        } else {
            int t = 0;
#pragma unroll
            for (int i = 0; i < NMAX; ++i) {        // W_A^(i k mod A); the table reads are issued before the first use
                w[i] = tw[t];
                t += k;
                if (t >= A) t -= A;
            }
        }
        re = im = 0.f;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            re += xr[i] * w[i].x - xi[i] * w[i].y;
            im += xr[i] * w[i].y + xi[i] * w[i].x;
        }
    };
    const float NEG = -__builtin_huge_valf();
    float best = NEG, second = NEG, best_re = 0.f, best_im = 0.f;
    int best_idx = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        float2 w[NMAX];
        float re, im;
        bin(k, w, re, im);
        const float m = __fsqrt_rn(fmaf(re, re, im * im));       // (magnitudes near the float32 range end in the float64 pass)
        const int kk = shift ? (k + A / 2) % A : k;
        if (best_idx == 0x7fffffff || mag_better(m, kk, best, best_idx)) {
            if (best_idx != 0x7fffffff) second = best;
            best = m;
            best_idx = kk;
            best_re = re;
            best_im = im;
        } else if (mag_gt(m, second)) second = m;
    }
    // first maximum over the wave: largest key, smallest bin index among its holders; then the largest other magnitude
    const bool has = best_idx != 0x7fffffff;
    const unsigned kb = has ? mag_key(best) : 0u;
    const unsigned top = wave_max_u32(kb);
    const int wi = (int)~wave_max_u32((has && kb == top) ? ~(unsigned)best_idx : 0u);
    const float m1 = key_mag(top);
    const float m2 = key_mag(wave_max_u32(!has ? 0u : (best_idx == wi ? (second == NEG ? 0u : mag_key(second)) : kb)));
    const float c_ang = rf.k_ang * sum_abs, two_b = 2.f * (be + c_ang);
    bool bad = !(m1 - m2 > two_b);                               // (also for a NaN winner)
    if (bad && m1 == m1) {
        // the runner-up is inside the independent-errors bound (2 % of the detections): every other bin against the
        // winner with the correlated form
        const int l1 = __ffsll((long long)__ballot(has && best_idx == wi)) - 1;
        const float re1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, best_re), l1));
        const float im1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, best_im), l1));
        const float inv1 = 1.f / m1, u1x = re1 * inv1, u1y = im1 * inv1;
        const int k1 = shift ? (wi + A - A / 2) % A : wi;
        float2 t1[NMAX];                    // conj(u_1) W^(i k_1)
        {
            int t = 0;
#pragma unroll
            for (int i = 0; i < NMAX; ++i) {
                const float2 w = tw[t];
                t1[i] = make_float2(u1x * w.x + u1y * w.y, u1x * w.y - u1y * w.x);
                t += k1;
                if (t >= A) t -= A;
            }
        }
        bad = false;
        for (int k = lane; k < A; k += 64) {
            const int kk = shift ? (k + A / 2) % A : k;
            if (kk == wi) continue;
            float2 w[NMAX];
            float re, im;
            bin(k, w, re, im);
            const float m = __fsqrt_rn(fmaf(re, re, im * im)), margin = m1 - m;
            bool ok = margin > two_b;
            if (!ok && m > 0.f) {
                const float inv = 1.f / m, ux = re * inv, uy = im * inv;
                float lin = 0.f;
#pragma unroll
                for (int i = 0; i < NMAX; ++i) {
                    const float dx = t1[i].x - (ux * w[i].x + uy * w[i].y), dy = t1[i].y - (ux * w[i].y - uy * w[i].x);
                    lin += e[i] * __fsqrt_rn(fmaf(dx, dx, dy * dy));
                }
                ok = margin > 1.001f * (lin + be * be / (2.f * fminf(m1, m))) + 2.f * c_ang;
            }
            bad |= !ok;
        }
    }
    const bool flag = __ballot(bad) != 0ull;
    if (lane == 0) {
        out_idx[slot] = wi;
        if (flag) {                                 // both lists share one refinement list
            const int pos = atomicAdd(rf.n_flag, 1);
            if (pos < rf.list_cap) rf.list[pos] = (int)slot | tag;
            if (rf.flagpos && pos < rf.dense_cap) rf.flagpos[slot] = pos + 1;
            if (tag) atomicAdd(ctl_el, 1);
        }
    }
}

__device__ __forceinline__ void detect_argmax_lanes(const DetectArgs &a, const float2 *tw, float2 xl, float l1v, int base, int n,
                                                    int shift, int32_t *out_idx, const ArgmaxRefine &rf, long slot, int lane,
                                                    int tag, const float2 (&wk)[DET_MAX_ANT], bool one_bin) {
    if (n <= 4) detect_argmax_list<4>(a.A, a.ctl + DCTL_EL, tw, xl, l1v, base, n, shift, out_idx, rf, slot, lane, tag, wk, one_bin);
    else detect_argmax_list<DET_MAX_ANT>(a.A, a.ctl + DCTL_EL, tw, xl, l1v, base, n, shift, out_idx, rf, slot, lane, tag, wk, one_bin);
}

// The stand-alone float32 argmax of mmw_angle_argmax_exact for lists of up to DET_MAX_ANT antennas, on the routine above (one
// wave per detection, its cells in lanes, DPP reductions, W_A^m in the LDS, per-lane twiddles in registers, PD detections of
// look-ahead).  k_angle_argmax (mmw_misc.h: cells n-fold in every lane's registers, global twiddle reads, shuffle butterflies)
// took 1.16 us per 256 x 128 frame of ~470 OS-CFAR detections for the azimuth list alone.
template <int NMAX>
__global__ __launch_bounds__(256) void k_angle_argmax_lanes(const float2 *rd, const int32_t *dets, const int32_t *counts, int32_t *out_idx,
                                                             int V, int S, int C, int cap, AntList ants, int A, int shift, const float2 *twA,
                                                             ArgmaxRefine rf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *tw = reinterpret_cast<float2 *>(smem);          // [A]
    __shared__ int tab[DET_MAX_ANT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < A; i += 256) tw[i] = twA[i];
    if (tid == 0) {
        static_for<DET_MAX_ANT>([&](auto I) {
            constexpr int i = decltype(I)::value;
            tab[i] = ants.idx[i];
        });
    }
    __syncthreads();
    const long f = blockIdx.y;
    int n_det = counts[f];
    if (n_det > cap) n_det = cap;
    const int n = ants.n;
    const bool mine = lane < n;
    const long ant = mine ? tab[lane] : 0;
    const float l1v = (mine && rf.l1) ? rf.l1[f * V + ant] : 0.f;
    const float2 *plane = rd + (f * V + ant) * (long)S * C;
    const bool one_bin = A == 64;
    float2 wk[DET_MAX_ANT];
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < DET_MAX_ANT; ++i) {
            wk[i] = one_bin ? tw[t] : make_float2(0.f, 0.f);
            t = (t + lane) & 63;
        }
    }
    const int stride = gridDim.x * 4;
    auto fetch = [&](int det) {
        float2 v = make_float2(0.f, 0.f);
        if (det < n_det && mine) {
            const long slot = f * cap + det;
            v = plane[(long)dets[slot * 2] * C + dets[slot * 2 + 1]];
        }
        return v;
    };
    constexpr int PD = 4;
    float2 q[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) q[i] = fetch(blockIdx.x * 4 + wave + i * stride);
    for (int det = blockIdx.x * 4 + wave; det < n_det; det += stride) {
        const float2 xl = q[0];
#pragma unroll
        for (int i = 0; i + 1 < PD; ++i) q[i] = q[i + 1];
        q[PD - 1] = fetch(det + PD * stride);
        detect_argmax_list<NMAX>(A, nullptr, tw, xl, l1v, 0, n, shift, out_idx, rf, f * cap + det, lane, 0, wk, one_bin);
    }
}

// Lane-per-detection form of the same float32 argmax + certainty test, for the 64-bin angle axis of the reference's point cloud
// (point_cloud_generator.py:143-214: az_el_fft_size 64) and lists of up to 8 antennas.  The wave-per-detection routine above
// spends ~130 wave instructions per evaluation (24 v_readlane broadcasts, three wave-wide reductions, the bookkeeping) on 32
// useful FMAs per lane; with ~470 noise-level OS-CFAR detections per frame and list that was a quarter of the OS pipeline.
// Here a lane owns a detection:
//   * zero-padded 64-point DFT of N cells as G = 64 / N register FFTs of length N:  S[b + G a] = FFT_N( x_i W_64^(i b) )[a]
//     (compile-time twiddles, packed float32 math): ~11 instructions per bin and detection, a quarter of the direct sums;
//   * the order is decided on p = re^2 + im^2 (no square root per bin), first maximum by a scan in output order; the two
//     square roots of the certainty test  m1 - m2 > 2 (Be + c_ang)  come last.  Rounding: one twiddle product, log2 N
//     butterfly levels, one (1 +- j) / sqrt 2 rotation, p and its root: under 12 eps sum |x_i|, inside c_ang = 4 (n + 4) eps sum |x_i|;
//   * lanes whose runner-up is inside the independent-errors bound run the correlated (pairwise) form of the test -- the one of
//     detect_argmax_list, same arithmetic: direct sums with table twiddles -- over their CANDIDATE bins only (those the generic
//     test does not clear: p_k >= (m1 - 2B)^2 less a rounding allowance); a wave loops as long as its busiest lane.
// Flagged evaluations join rf.list / rf.flagpos with one atomic per wave.  Every index it reports unflagged is the float64
// one by the same proof as before; what it flags is re-evaluated in float64 by the caller.
// the evaluation of one lane's detection: cells x (zeros past the list), error scales e (be = their sum), sum_abs = sum |re| + |im|
template <int N, bool SHIFT>
__device__ __forceinline__ void argmax_lane_eval(const cplx<float> (&x)[N], const float (&e)[N], float be, float sum_abs, const float2 *tw,
                                                 const ArgmaxRefine &rf, bool active, long slot, int32_t *__restrict__ out_idx, int lane) {
    constexpr int A = 64, G = A / N;
    // ---- all 64 bins: p[k] = |S_k|^2
    unsigned key[A];
    static_for<G>([&](auto B) {
        constexpr int b = decltype(B)::value;
        cplx<float> y[N];
        static_for<N>([&](auto I) {
            constexpr int i = decltype(I)::value;
            y[i] = mul_w<64, i * b, float>(x[i]);
        });
        RegFFT<N, float>::run(y);
        static_for<N>([&](auto Aa) {
            constexpr int a = decltype(Aa)::value;
            const cplx<float> sk = y[bitrev<N>(a)];
            key[b + G * a] = mag_key(fmaf(sk.x, sk.x, sk.y * sk.y));
        });
    });
    // ---- first maximum in output order (np.argmax), largest other value
    unsigned k1key = 0u, k2key = 0u;
    int wi = 0;
    static_for<A>([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        constexpr int k = SHIFT ? (kk + A / 2) % A : kk;
        const unsigned v = key[k];
        if constexpr (kk == 0) {
            k1key = v;
        } else {
            const bool up = v > k1key;
            k2key = up ? k1key : max(k2key, v);
            k1key = up ? v : k1key;
            wi = up ? kk : wi;
        }
    });
    const float m1p = __fsqrt_rn(key_mag(k1key)), m2p = __fsqrt_rn(key_mag(k2key));
    const float c_ang = rf.k_ang * sum_abs, two_b = 2.f * (be + c_ang);
    bool bad = !(m1p - m2p > two_b);                            // (also for a NaN winner)
    if (bad && m1p == m1p) {
        // candidate bins: everything the generic test does not clear against the winner
        const float thr = m1p - two_b, t2 = thr > 0.f ? thr * thr * (1.f - 2e-6f) : 0.f;
        unsigned long long cand = 0ull;
        static_for<A>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            constexpr int k = SHIFT ? (kk + A / 2) % A : kk;
            if (!(key_mag(key[k]) < t2)) cand |= 1ull << k;
        });
        const int k1 = SHIFT ? (wi + A - A / 2) % A : wi;
        cand &= ~(1ull << k1);
        // the winner by direct sums (the arithmetic of detect_argmax_list), conj(u_1) W^(i k_1)
        float2 t1[N];
        float re1 = 0.f, im1 = 0.f;
        {
            int t = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float2 w = tw[t];
                t1[i] = w;
                re1 += x[i].x * w.x - x[i].y * w.y;
                im1 += x[i].x * w.y + x[i].y * w.x;
                t = (t + k1) & (A - 1);
            }
        }
        const float m1 = __fsqrt_rn(fmaf(re1, re1, im1 * im1));
        const float inv1 = 1.f / m1, u1x = re1 * inv1, u1y = im1 * inv1;
#pragma unroll
        for (int i = 0; i < N; ++i) t1[i] = make_float2(u1x * t1[i].x + u1y * t1[i].y, u1x * t1[i].y - u1y * t1[i].x);
        bad = !(m1 > 0.f);                                      // (a zero or non-finite winner stays flagged)
        while (__ballot(cand != 0ull) != 0ull) {
            if (cand != 0ull) {
                const int k = __ffsll((long long)cand) - 1;
                cand &= cand - 1ull;
                float2 w[N];
                float re = 0.f, im = 0.f;
                int t = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    w[i] = tw[t];
                    t = (t + k) & (A - 1);
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    re += x[i].x * w[i].x - x[i].y * w[i].y;
                    im += x[i].x * w[i].y + x[i].y * w[i].x;
                }
                const float m = __fsqrt_rn(fmaf(re, re, im * im)), margin = m1 - m;
                bool ok = margin > two_b;
                if (!ok && m > 0.f) {
                    const float inv = 1.f / m, ux = re * inv, uy = im * inv;
                    float lin = 0.f;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const float dx = t1[i].x - (ux * w[i].x + uy * w[i].y), dy = t1[i].y - (ux * w[i].y - uy * w[i].x);
                        lin += e[i] * __fsqrt_rn(fmaf(dx, dx, dy * dy));
                    }
                    ok = margin > 1.001f * (lin + be * be / (2.f * fminf(m1, m))) + 2.f * c_ang;
                }
                bad |= !ok;
            }
        }
    }
    const bool flag = active && bad;
    if (active) out_idx[slot] = wi;
    const unsigned long long fm = __ballot(flag);
    if (fm != 0ull) {
        int base = 0;
        if (lane == 0) {
            base = atomicAdd(rf.n_flag, __popcll(fm));
            if (rf.n_tagged) atomicAdd(rf.n_tagged, __popcll(fm));
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (flag) {
            const int pos = base + __popcll(fm & ((1ull << lane) - 1ull));
            if (pos < rf.list_cap) rf.list[pos] = (int)slot | rf.tag;
            if (rf.flagpos && pos < rf.dense_cap) rf.flagpos[slot] = pos + 1;
        }
    }
}

template <int N, bool SHIFT>
__global__ __launch_bounds__(256) void k_angle_argmax_dets(const float2 *__restrict__ rd, const int32_t *__restrict__ dets,
                                                            const int32_t *__restrict__ counts, int32_t *__restrict__ out_idx, int V, int S,
                                                            int C, int cap, AntList ants, const float2 *__restrict__ twA, ArgmaxRefine rf) {
    constexpr int A = 64;
    static_assert(N == 4 || N == 8 || N == 16, "lists of up to 4 / 8 / 16 antennas");
    __shared__ float2 tw[A];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < A) tw[tid] = twA[tid];
    __syncthreads();
    const long f = blockIdx.y;
    int n_det = counts[f];
    if (n_det > cap) n_det = cap;
    const int n = ants.n;
    float e[N], be = 0.f;
    long poff[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int ant = ants.idx[i < n ? i : 0];
        e[i] = i < n ? rf.k_fft * rf.l1[f * V + ant] : 0.f;
        be += e[i];
        poff[i] = (f * V + ant) * (long)S * C;
    }
    for (int det0 = (blockIdx.x * 4 + wave) * 64; det0 < n_det; det0 += gridDim.x * 256) {
        const int det = det0 + lane;
        const bool active = det < n_det;
        const long slot = f * cap + (active ? det : det0);
        const long cell = (long)dets[slot * 2] * C + dets[slot * 2 + 1];
        cplx<float> x[N];
        float sum_abs = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float2 v = rd[poff[i] + cell];                    // unconditional (a valid slot); unused antennas read as zeros
            x[i] = i < n ? cplx<float>{v.x, v.y} : cplx<float>{0.f, 0.f};
            sum_abs += fabsf(x[i].x) + fabsf(x[i].y);
        }
        argmax_lane_eval<N, SHIFT>(x, e, be, sum_abs, tw, rf, active, slot, out_idx, lane);
    }
}

// The same evaluation over a flat list of RECORDS (the late argmax of mmw_detect_points): the screening workgroup of a frame
// copied the cells of its detections -- rec_cells[rec][NV], list 1 at [off, off + n) -- and the frame's plane norms (l1c) into
// scratch of the context, so this launch reads nothing the caller's next range-Doppler call overwrites; rec_slot[rec] = f * cap
// + slot.  Every lane of every wave has work (a frame's ~76 CA-CFAR detections fill 1.2 waves of the per-frame form).
template <int N, bool SHIFT>
__global__ __launch_bounds__(256) void k_angle_argmax_recs(const float2 *__restrict__ rec_cells, const int32_t *__restrict__ rec_slot,
                                                            const int *__restrict__ n_recs, int rec_cap, int NV, int off,
                                                            const float *__restrict__ l1c, int32_t *__restrict__ out_idx, int V, int cap,
                                                            AntList ants, const float2 *__restrict__ twA, ArgmaxRefine rf) {
    constexpr int A = 64;
    __shared__ float2 tw[A];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < A) tw[tid] = twA[tid];
    __syncthreads();
    int total = *n_recs;
    if (total > rec_cap) total = rec_cap;
    const int n = ants.n;
    for (int r0 = (blockIdx.x * 4 + wave) * 64; r0 < total; r0 += gridDim.x * 256) {
        const int rec = r0 + lane;
        const bool active = rec < total;
        const long q = active ? rec : r0;
        const long slot = rec_slot[q];
        const long f = slot / cap;
        cplx<float> x[N];
        float e[N], be = 0.f, sum_abs = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float2 v = rec_cells[q * NV + off + (i < n ? i : 0)];
            const float l = l1c[f * V + ants.idx[i < n ? i : 0]];
            x[i] = i < n ? cplx<float>{v.x, v.y} : cplx<float>{0.f, 0.f};
            e[i] = i < n ? rf.k_fft * l : 0.f;
            be += e[i];
            sum_abs += fabsf(x[i].x) + fabsf(x[i].y);
        }
        argmax_lane_eval<N, SHIFT>(x, e, be, sum_abs, tw, rf, active, slot, out_idx, lane);
    }
}

// Ordered compaction of the frame's bit mask (bit r * C + c, LDS) into dets / counts, then the angle argmax of every
// detection.  ws: 96 ints of LDS ([48, 80) = antenna table), tw: W_A^m in LDS.  Ends with every thread past its last
// use of bits / ws.
// n_spec undecided cells (ws[80 ..]) ride along SPECULATIVELY: their angle indices are computed as if they were detections, in
// the slots cap - 1 - u at the end of the frame's list, so that nothing but k_detect_insert (a list insertion) depends on
// k_cfar_cell_exact's decision -- in round 3 the flagged frames' compaction and argmax, and the refinement of THEIR flagged
// evaluations, waited for it: three latency chains one after the other behind the screening kernel.
// false: the certain detections and the speculative slots do not fit cap together (the caller hands the frame back).
template <bool SYNC = false>
__device__ __forceinline__ bool detect_finish(const DetectArgs &a, long f, const unsigned *bits, int *ws, const float2 *tw, int tid, int n_spec) {
    const int C = a.C;
    int base = 0;
    for (int w0 = 0; w0 < a.words; w0 += DET_NT) {
        const int w = w0 + tid;
        unsigned word = w < a.words ? bits[w] : 0u;
        int total;
        int pos = base + block_excl_scan(__popc(word), ws, &total, tid);
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
    if (n_spec > 0 && base + n_spec > a.cap) return false;
    if (tid == 0) a.counts[f] = base;           // exact even beyond cap (MMW_ERR_TRUNCATED is the caller's check)
    if (tid < n_spec) {
        const int cell = ws[80 + tid], r = cell / C;
        const long slot = f * a.cap + (a.cap - 1 - tid);
        a.dets[slot * 2] = r;
        a.dets[slot * 2 + 1] = cell - r * C;
        a.spec[f * (2 * DET_SPEC + 1) + 1 + tid] = cell;
        a.spec[f * (2 * DET_SPEC + 1) + 1 + DET_SPEC + tid] = 0;       // the decision (k_cfar_cell_exact)
    }
    if (tid == 0 && n_spec > 0) a.spec[f * (2 * DET_SPEC + 1)] = n_spec;
    if constexpr (SYNC) det_mark_acc(a, 3); else det_mark(a, 3);
    const int n_az = a.az.n, n_el = a.el.n;
    if (n_az == 0 && n_el == 0) return true;
    __syncthreads();                            // the workgroup's own dets are visible to all of its waves
    const int n_cert = base < a.cap ? base : a.cap, n_det = n_cert + n_spec;
    // evaluation j of the frame -> its slot: the certain detections in order, then the speculative ones from the end
    auto slot_of = [&](int j) { return j < n_cert ? j : a.cap - 1 - (j - n_cert); };
    const int lane = tid & 63, wave = tid >> 6;
    const int *tab = ws + 48;
    // lanes [0, n_az): azimuth list, lanes [DET_LIST2, DET_LIST2 + n_el): elevation list
    const bool mine = (lane < n_az) || (lane >= DET_LIST2 && lane < DET_LIST2 + n_el);
    const long ant = mine ? tab[lane] : 0;
    // SYNC: the cube and the norms were written by the producer kernel that is still running on other XCDs: coherent
    // (sc1) loads, never the XCD-local L2's copy of a line
    float l1v = 0.f;
    if (mine) l1v = SYNC ? __hip_atomic_load(a.l1 + f * a.V + ant, MMW_RLX_AGENT) : a.l1[f * a.V + ant];
    const float2 *plane = a.rd + (f * a.V + ant) * (long)a.S * C;
    [[maybe_unused]] const auto frame_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.rd + f * a.V * (long)a.S * C), 0,
                                                                              (int)((unsigned)a.V * (unsigned)a.S * (unsigned)C * 8u), 0x00020000);
    constexpr int NW = DET_NT / 64;
    // the cells of the wave's next detection travel while it works on this one
    auto fetch = [&](int det) {
        float2 v = make_float2(0.f, 0.f);
        if (det < n_det && mine) {
            const long slot = f * a.cap + slot_of(det);
            const long cell = (long)a.dets[slot * 2] * C + a.dets[slot * 2 + 1];
            if constexpr (SYNC)
                v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(frame_rs, (unsigned)((ant * a.S * C + cell) * 8), 0, 16));
            else v = plane[cell];
        }
        return v;
    };
    if (a.rec_cells) {
        // late argmax: the cells of every evaluation into the context's record list (one cold strided read per cell, all of a
        // frame's in flight together: no arithmetic follows them here)
        const int NV = n_az + n_el;
        if (tid == 0) ws[41] = atomicAdd(a.ctl + DCTL_RECS, n_det);
        for (int i = tid; i < a.V; i += DET_NT) a.l1_copy[f * a.V + i] = SYNC ? __hip_atomic_load(a.l1 + f * a.V + i, MMW_RLX_AGENT) : a.l1[f * a.V + i];
        __syncthreads();
        const int rec0 = ws[41];
        for (int j = tid; j < n_det * NV; j += DET_NT) {
            const int det = j / NV, i = j - det * NV, rec = rec0 + det;
            if (rec >= a.rec_cap) continue;                 // (cannot happen: rec_cap = frames x cap)
            const long slot = f * a.cap + slot_of(det);
            const long cell = (long)a.dets[slot * 2] * C + a.dets[slot * 2 + 1];
            const long av = tab[i < n_az ? i : DET_LIST2 + i - n_az];
            float2 v;
            if constexpr (SYNC)
                v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(frame_rs, (unsigned)((av * a.S * C + cell) * 8), 0, 16));
            else v = a.rd[(f * a.V + av) * (long)a.S * C + cell];
            a.rec_cells[(long)rec * NV + i] = v;
            if (i == 0) a.rec_slot[rec] = (int32_t)slot;
        }
        return true;
    }
    // The cells of the wave's next PD detections travel while it works on this one (each fetch is a cold, strided read --
    // 2-4 k clocks from the memory side under load, against ~1 k clocks of arithmetic per detection: with one detection of
    // look-ahead the phase was latency bound, 33 k clocks for ~5 detections per wave).
    const bool one_bin = a.A == 64;
    float2 wk[DET_MAX_ANT];
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < DET_MAX_ANT; ++i) {
            wk[i] = one_bin ? tw[t] : make_float2(0.f, 0.f);
            t = (t + lane) & 63;
        }
    }
    constexpr int PD = 4;
    float2 q[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) q[i] = fetch(wave + i * NW);
    for (int det = wave; det < n_det; det += NW) {
        const float2 xl = q[0];
#pragma unroll
        for (int i = 0; i + 1 < PD; ++i) q[i] = q[i + 1];
        q[PD - 1] = fetch(det + PD * NW);
        const long slot = f * a.cap + slot_of(det);
        if (n_az) detect_argmax_lanes(a, tw, xl, l1v, 0, n_az, a.shift_az, a.az_idx, a.rf_az, slot, lane, 0, wk, one_bin);
        if (n_el) detect_argmax_lanes(a, tw, xl, l1v, DET_LIST2, n_el, a.shift_el, a.el_idx, a.rf_el, slot, lane, REFINE_SECOND, wk, one_bin);
    }
    return true;
}

// One band of a frame: cells under test in plane rows [r0, r0 + nb), their magnitudes and the halo rows [r0 - hr, r0 + nb + hr)
// in the LDS (Xs row 0 = plane row r0 - hr); the detection bit of cell (r, c) is bit r * C + c of the frame mask.
// An undecided cell: slot u of the frame (LDS list ws[80 + u], ws[47] = count) and an entry of the global work list of
// k_cfar_cell_exact; too many for either: ws[45], the frame is handed back.
__device__ __forceinline__ void det_undecided(const DetectArgs &a, long f, int cell, int *ws) {
    const int u = atomicAdd(&ws[47], 1);
    if (u >= DET_SPEC) {
        ws[45] = 1;
        return;
    }
    ws[80 + u] = cell;
    const int pos = atomicAdd(a.ctl + DCTL_CELLS, 1);
    if (pos < a.cell_cap) {
        a.cells[2 * pos] = (int)f;
        a.cells[2 * pos + 1] = cell | (u << 24);
        ws[44] = 1;
    } else
        ws[45] = 1;
}

// CFAR decision of every cell of the band (run-time window, one cell per thread and pass): column sums over the training /
// guard rows of the window into Vt / Vg, then the row-wise combination, the band test, detection bits into `bits` and
// undecided cells into the global list (ws[44] / ws[45] = has undecided cells / list overflow).
__device__ __forceinline__ void cfar_band_rt(const DetectArgs &a, long f, int r0, int nb, const float *Xs, float *Vt, float *Vg,
                                             unsigned *bits, int *ws, double Bf, int tid) {
    const int C = a.C, lane = tid & 63;
    const int tr = a.tr, td = a.td, gr = a.gr, gd = a.gd, hd = td + gd, hr = tr + gr;
    const float inv_n = (float)(1.0 / (double)a.n_train);
    const double alpha = a.scale, band0 = (1.0 + fabs(alpha)) * Bf;
    // cell i = tid, tid + NT, ... of a band as (row rr, column c) without a division per cell
    const int rr_t = tid / C, c_t = tid - rr_t * C, dq = DET_NT / C, dm = DET_NT - dq * C;
    const int g_lo = tr, g_hi = tr + 2 * gr;                // guard rows / columns inside the window
    const int Wr = 2 * hr + 1, Wd = 2 * hd + 1;
    const int gc_lo = td, gc_hi = td + 2 * gd;
    const int cells = nb * C;
    {
        int rr = rr_t, c = c_t;
        for (int i = tid; i < cells; i += DET_NT) {
            const float *col = Xs + rr * C + c;             // window rows rr .. rr + Wr - 1 of the band buffer
            float t = 0.f, g = 0.f;
            for (int dr = 0; dr < g_lo; ++dr) t += col[dr * C];
            for (int dr = g_lo; dr <= g_hi; ++dr) g += col[dr * C];
            for (int dr = g_hi + 1; dr < Wr; ++dr) t += col[dr * C];
            Vt[i] = t;
            Vg[i] = g;
            c += dm;
            rr += dq;
            if (c >= C) {
                c -= C;
                ++rr;
            }
        }
    }
    __syncthreads();
    int rr = rr_t, c = c_t;
    for (int i0 = tid - lane; i0 < cells; i0 += DET_NT) {        // wave-uniform trip count (ballot below)
        const int i = i0 + lane;
        bool det = false, unc = false;
        const int r = r0 + rr;
        if (i < cells && c >= hd && c < C - hd) {
            // training cells = training rows of every window column + guard rows of the columns outside the guard
            const float *pt = Vt + i - hd, *pg = Vg + i - hd;
            float tot = 0.f;
            for (int dc = 0; dc < gc_lo; ++dc) tot += pt[dc] + pg[dc];
            for (int dc = gc_lo; dc <= gc_hi; ++dc) tot += pt[dc];
            for (int dc = gc_hi + 1; dc < Wd; ++dc) tot += pt[dc] + pg[dc];
            const double X = (double)Xs[(rr + hr) * C + c], T = alpha * (double)(tot * inv_n);
            const double d = X - T, band = band0 + 3.0e-6 * (X + fabs(T));
            det = d > band;
            unc = !det && !(d <= -band);
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
        if (unc) det_undecided(a, f, r * C + c, ws);
        c += dm;
        rr += dq;
        if (c >= C) {
            c -= C;
            ++rr;
        }
    }
}

// The same with the window a compile-time constant -- and a pre-screen that leaves almost nothing to sum.
//
// Round 3 summed the window of EVERY cell (a thread: four rows of a column, then four columns of a row; ~75 vector
// instructions per cell, 53 k of a 256 x 128 frame's 112 k clocks -- the phase was bound by vector-instruction issue, not by the
// LDS).  But a cell can only be a detection, or undecided, if X is not far below its threshold, and every term of the
// training sum is non-negative, so ANY partial sum bounds the threshold from below:
//   tot(r, c) = sum over the 2 HD + 1 window columns of Vt[r][c + dc] + (guard rows of the outer columns)
//             >= (2 HD + 1) * min over the row of Vt[r][.]            Vt[r][c] = training rows of window column c
// Pass 1 computes Vt only (a thread: four rows of a column, WR + 3 reads) and the row minima (wave reduction + one LDS
// atomic per wave and row); pass 2 compares each cell with theta_r = alpha / N * (2 HD + 1) * min_r and appends the few
// that are not CERTAINLY below their threshold to a list (noise cells of the synthetic workload: ~2 %); pass 3 sums the
// windows of the listed cells only, one per thread (2 HD + 1 reads of Vt + the guard rows of the outer columns out of Xs),
// and applies the band test.  A cell is dropped by pass 2 only if   X (1 + 3e-6) + band0 <= theta (1 - 4e-6):   theta is a
// float32 lower bound of the float32 threshold T' the full sum would give (its 21-term float32 sum is >= (1 - 21 * 2^-24)
// times the exact sum of the same float32 terms >= their 13-term minimum bound; the factor covers that and theta's own three
// roundings), so d = X - T' <= -(band0 + 3e-6 (X + T')): the very condition under which the full test says "certainly none".
// alpha <= 0 (a pfa >= 1): no cell is dropped.
constexpr int det_band_pitch(int C, int hd) { return (C + 2 * hd + 4 + 4 + 3) / 4 * 4; }
template <int TR, int TD, int GR, int GD>
__device__ __forceinline__ void cfar_band_ct(const DetectArgs &a, long f, int r0, int nb, int xs_rows, const float *Xs, float *Vt, float *Vg,
                                             unsigned *bits, int *ws, double Bf, int tid) {
    constexpr int HR = TR + GR, HD = TD + GD, WR = 2 * HR + 1, WD = 2 * HD + 1, R = 4, OFF = HD + 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int C = a.C, P = a.band_pitch, cg_n = (C + R - 1) / R;
    // the test itself in float32: alpha / N and the band are rounded once (and the band widened by 1e-6 for it), the
    // product and the difference add two more roundings of T -- all inside the 3e-6 (X + |T|) term
    const float alpha_n = (float)(a.scale / (double)a.n_train), band0 = (float)((1.0 + fabs(a.scale)) * Bf * 1.000001);
    int *rowmin = reinterpret_cast<int *>(Vg);              // [64] bit patterns of the row minima of Vt (non-negative floats order as ints)
    int *cand = rowmin + 64;                                // band cells (rr * C + c) that need their window summed
    const int cand_cap = a.band_rows * P - 64;              // >= band_rows * C
    if (tid < 64) rowmin[tid] = 0x7f800000;
    if (tid == 0) ws[46] = 0;
    __syncthreads();
    // ---- pass 1: training-row sums of every window column (a thread: four columns of one row, the 2 TR training rows as 16-byte
    //      reads), row minima (the two halves of a wave hold a row each when C % 128 == 0: one DPP reduction serves both)
    const bool vec = (C & 3) == 0, halves_uniform = (cg_n & 31) == 0;
    for (int i0 = tid & ~63; i0 < nb * cg_n; i0 += DET_NT) {
        const int i = i0 + (tid & 63);
        const bool live = i < nb * cg_n;
        const int ii = live ? i : 0, rr = ii / cg_n, c0 = R * (ii - rr * cg_n);
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dr = 0; dr < WR; ++dr) {
            if (dr >= TR && dr <= TR + 2 * GR) continue;
            const float *row = Xs + (rr + dr) * C + c0;
            if (vec) acc += *reinterpret_cast<const f4 *>(row);
            else acc += f4{row[0], c0 + 1 < C ? row[1] : 0.f, c0 + 2 < C ? row[2] : 0.f, c0 + 3 < C ? row[3] : 0.f};
        }
        if (live) {
            float *dst = Vt + rr * P + OFF + c0;            // (float offset 4 g + OFF: 8-byte aligned when OFF is even)
            if constexpr (OFF % 2 == 0) {
                *reinterpret_cast<float2 *>(dst) = make_float2(acc.x, acc.y);
                *reinterpret_cast<float2 *>(dst + 2) = make_float2(acc.z, acc.w);
            } else {
                dst[0] = acc.x;
                dst[1] = acc.y;
                dst[2] = acc.z;
                dst[3] = acc.w;
            }
        }
        float m = acc.x;
        if (c0 + 1 < C) m = fminf(m, acc.y);
        if (c0 + 2 < C) m = fminf(m, acc.z);
        if (c0 + 3 < C) m = fminf(m, acc.w);
        int key = live ? __builtin_bit_cast(int, m) : 0x7f800000;
        if (key < 0 || key > 0x7f800000) key = 0;           // NaN (a degenerate plane is not screened; belt and braces): no drop
        if (halves_uniform) {               // (DPP network: a __shfl_xor butterfly is six dependent LDS round trips)
            unsigned v = (unsigned)key;
            v = min(v, dpp_u32<0xB1, 0xf>(v, v));
            v = min(v, dpp_u32<0x4E, 0xf>(v, v));
            v = min(v, dpp_u32<0x124, 0xf>(v, v));
            v = min(v, dpp_u32<0x128, 0xf>(v, v));
            v = min(v, dpp_u32<0x142, 0xa>(v, v));          // row_bcast:15: lanes 31 / 63 hold the minima of lanes 0..31 / 32..63
            if ((tid & 31) == 31 && live && rr < 64) atomicMin(&rowmin[rr], (int)v);
        } else if (live && rr < 64)
            atomicMin(&rowmin[rr], key);
    }
    __syncthreads();
    det_lap(a, 11);
    // ---- pass 2: cells that may reach their threshold
    for (int i = tid; i < nb * cg_n; i += DET_NT) {
        const int rr = i / cg_n, c0 = R * (i - rr * cg_n);
        const float theta = a.scale > 0.0 && rr < 64 ? alpha_n * (float)WD * __builtin_bit_cast(float, rowmin[rr]) * (1.f - 4e-6f) : 0.f;
        const float *xrow = Xs + (rr + HR) * C + c0;
        float x[R];
        if ((C & 3) == 0) {                 // one aligned 16-byte read (four 4-byte reads at this lane stride conflict)
            const f4 q = *reinterpret_cast<const f4 *>(xrow);
#pragma unroll
            for (int j = 0; j < R; ++j) x[j] = q[j];
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) x[j] = c0 + j < C ? xrow[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int c = c0 + j;
            if (c >= HD && c < C - HD && !(fmaf(x[j], 1.f + 3e-6f, band0) <= theta)) {
                const int pos = atomicAdd(&ws[46], 1);
                if (pos < cand_cap) cand[pos] = rr * C + c;
            }
        }
    }
    __syncthreads();
    det_lap(a, 12);
    // ---- pass 3: the listed cells, one per thread
    const int n_cand = min(ws[46], cand_cap);
    if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) a.clk[14] += n_cand;
    for (int i = tid; i < n_cand; i += DET_NT) {
        const int cell = cand[i], rr = cell / C, c = cell - rr * C, r = r0 + rr;
        // training cells = training rows of every window column + guard rows of the columns outside the guard
        const float *pt = Vt + rr * P + OFF + c - HD;
        float tot = 0.f;
#pragma unroll
        for (int dc = 0; dc < WD; ++dc) tot += pt[dc];
        const float *px = Xs + (rr + TR) * C + c - HD;      // first guard row of the window, first window column
#pragma unroll
        for (int dc = 0; dc < WD; ++dc) {
            if (dc >= TD && dc <= TD + 2 * GD) continue;
            float g = 0.f;
#pragma unroll
            for (int dr = 0; dr <= 2 * GR; ++dr) g += px[dr * C + dc];
            tot += g;
        }
        const float X = Xs[(rr + HR) * C + c], T = alpha_n * tot;
        const float d = X - T, band = band0 + 3.0e-6f * (X + fabsf(T));
        if (d > band) atomicOr(&bits[((long)r * C + c) >> 5], 1u << (((long)r * C + c) & 31));
        else if (!(d <= -band)) det_undecided(a, f, r * C + c, ws);
    }
    det_lap(a, 13);
}

// LDS of k_detect_screen: two band buffers of float32 magnitudes (band rows + halo), two float32 column-sum buffers, the frame's
// bit mask, 96 ints, W_A^m table
inline size_t detect_tail_lds(int words, int A) { return (((size_t)words * 4 + 15) & ~(size_t)15) + 128 * 4 + (size_t)A * 8; }
inline size_t detect_screen_lds(int xs_rows, int C, int band_rows, int band_pitch, int words, int A) {
    return 2 * (((size_t)xs_rows * C * 4 + 15) & ~(size_t)15) + 2 * (((size_t)band_rows * band_pitch * 4 + 15) & ~(size_t)15) +
           detect_tail_lds(words, A);
}
constexpr int DET_LOADS = 5;        // 16-byte loads a thread has in flight for the next band (two cells each)

// TR, TD, GR, GD >= 0: the window is a compile-time constant (every loop over it unrolls: all of a cell's LDS reads are
// issued before the first add); -1: taken from the arguments at run time.
//
// The frame's plane of antenna 0 STREAMS through the LDS in bands of a.band_rows rows: while the workgroup works on band b
// (magnitudes in one of two band buffers), the complex cells of band b + 1 -- its rows and the 2 hr halo rows around them,
// so halo rows are read twice, from cache -- are in flight into registers (at most DET_LOADS 16-byte loads per thread).
// Round 3 held the whole plane's magnitudes (128 KB for 256 x 128) and loaded it in one exposed phase (18 k of a frame's
// 112 k clocks), which also left room for 20-row bands only: 640 of the 1024 threads had an item per pass.
//
// SYNC: the consumer of the overlapped schedule -- a persistent grid on its own CU set; a workgroup draws a frame ticket,
// waits for the producer to publish the frame's V planes, and screens it while the producer is still transforming later
// frames on the other CUs.  A wait that times out (the two launches did not run side by side) raises the abort word: the
// frame in hand and every ticket still to be drawn get counts[f] = -1, i.e. they are handed back to the caller's float64
// path like any other frame the screening cannot decide.
template <int TR, int TD, int GR, int GD, bool SYNC>
__device__ __forceinline__ void detect_screen_frame(const DetectArgs &a, const long f, char *smem) {
    constexpr bool CT = TR >= 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    int tid_ = threadIdx.x;
    // (persistent form: re-derived behind an opaque asm every frame, else hipcc hoists every frame-invariant index and
    //  address out of the frame loop and spills them around it)
    if constexpr (SYNC) asm volatile("" : "+v"(tid_));
    const int S = a.S, C = a.C, tid = tid_;
    const int hr = CT ? TR + GR : a.tr + a.gr, hd = CT ? TD + GD : a.td + a.gd;
    const bool window_fits = S > 2 * hr && C > 2 * hd;
    const int xs_rows = a.band_rows + 2 * hr;
    const size_t xs_bytes = ((size_t)xs_rows * C * 4 + 15) & ~(size_t)15;
    const size_t band_bytes = ((size_t)a.band_rows * a.band_pitch * 4 + 15) & ~(size_t)15;
    float *Xs0 = reinterpret_cast<float *>(smem);
    float *Vt = reinterpret_cast<float *>(smem + 2 * xs_bytes);         // column sums over the training rows of the window
    float *Vg = reinterpret_cast<float *>(smem + 2 * xs_bytes + band_bytes);      // ... over its guard rows
    size_t off = 2 * xs_bytes + 2 * band_bytes;
    unsigned *bits = reinterpret_cast<unsigned *>(smem + off);          // a.words words: the frame's detection mask
    off += ((size_t)a.words * 4 + 15) & ~(size_t)15;
    int *ws = reinterpret_cast<int *>(smem + off);          // [0, 40) scan, 44 undecided, 45 overflow, 46 candidates, 47 undecided cells, [48, 80) antennas, [80, 112) those cells
    float2 *tw = reinterpret_cast<float2 *>(smem + off + 128 * 4);
    if constexpr (SYNC) det_mark_acc(a, 0); else det_mark(a, 0);
    if (tid < 4) ws[44 + tid] = 0;
    for (int w = tid; w < a.words; w += DET_NT) bits[w] = 0u;

    // error band scale; 1.0001 covers the float32 summation of the L1 norm itself
    const float l1v = SYNC ? __hip_atomic_load(a.l1 + f * a.V, MMW_RLX_AGENT) : a.l1[f * a.V];
    const double Bf = (double)a.k_fft * (double)l1v * 1.0001;
    // NaN / inf samples in antenna 0, or a scale at which float32 squares over- / underflow (an all-zero plane is fine)
    const bool degenerate = !(l1v == 0.f || (l1v >= 1e-10f && l1v <= 1e18f));
    const long n_plane = (long)S * C;
    const float2 *p = a.rd + f * a.V * n_plane;             // antenna 0
    float *mg = a.mag32 ? a.mag32 + f * n_plane : nullptr;
    [[maybe_unused]] const auto prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(p), 0, (int)(n_plane * 8), 0x00020000);
    // The cells of plane rows [row_lo, row_hi) -- one contiguous run of the plane -- as 16-byte loads of two consecutive cells,
    // DET_LOADS per thread, UNCONDITIONAL and clamped (a guarded load is followed by its own s_waitcnt, and so is a load under a
    // run-time branch: the first version chose between 16- and 8-byte loads by the parity of C and every load waited for the one
    // before).  A run that starts at an odd cell is 8-byte aligned only: fine for global / buffer loads (dword alignment).
    auto issue = [&](int row_lo, int row_hi, f4 (&v)[DET_LOADS]) {
        const long base = (long)row_lo * C;
        const int n2 = ((row_hi - row_lo) * C + 1) / 2;
#pragma unroll
        for (int u = 0; u < DET_LOADS; ++u) {
            const int i = min(tid + u * DET_NT, n2 - 1);
            if constexpr (SYNC) v[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(prs, (unsigned)((base + 2 * i) * 8), 0, 16));
            else v[u] = *reinterpret_cast<const f4 *>(p + base + 2 * i);
        }
    };
    // |.| of the loaded cells into band buffer Xs (row 0 = plane row row_lo), and out to mag32 for the rows [own_lo, own_hi)
    auto land = [&](int row_lo, int row_hi, int own_lo, int own_hi, const f4 (&v)[DET_LOADS], float *Xs) {
        const long base = (long)row_lo * C;
        const int n = (row_hi - row_lo) * C;
        const int keep_lo = (own_lo - row_lo) * C, keep_hi = (own_hi - row_lo) * C;
#pragma unroll
        for (int u = 0; u < DET_LOADS; ++u) {
            const int i = tid + u * DET_NT;
            if (2 * i < n) {
                const float m0 = __fsqrt_rn(fmaf(v[u].x, v[u].x, v[u].y * v[u].y));
                const float m1 = __fsqrt_rn(fmaf(v[u].z, v[u].z, v[u].w * v[u].w));
                const bool two = 2 * i + 1 < n;
                if (Xs) {
                    if (two) *reinterpret_cast<float2 *>(Xs + 2 * i) = make_float2(m0, m1);
                    else Xs[2 * i] = m0;
                }
                if (mg) {
                    // (written once, read by the caller later: non-temporal; a pair never straddles the kept rows when C is even)
                    const bool k0 = 2 * i >= keep_lo && 2 * i < keep_hi, k1 = two && 2 * i + 1 >= keep_lo && 2 * i + 1 < keep_hi;
                    if (k0 && k1 && ((base + 2 * i) & 1) == 0)
                        __builtin_nontemporal_store(f32x2{m0, m1}, reinterpret_cast<f32x2 *>(mg + base + 2 * i));
                    else {
                        if (k0) mg[base + 2 * i] = m0;
                        if (k1) mg[base + 2 * i + 1] = m1;
                    }
                }
            }
        }
    };
    if (!window_fits) {
        // no cell under test (the reference returns no detections: ca_cfar.py:99-102); the magnitudes are still wanted
        if (mg) {
            const int rows_per = max(1, 2 * DET_LOADS * DET_NT / C);
            for (int r0 = 0; r0 < S; r0 += rows_per) {
                f4 v[DET_LOADS];
                const int r1 = min(S, r0 + rows_per);
                issue(r0, r1, v);
                land(r0, r1, r0, r1, v, nullptr);
            }
        }
    } else {
        const int r_lo = hr, r_hi = S - hr;                 // rows with cells under test
        f4 v[DET_LOADS];
        issue(r_lo - hr, min(r_lo + a.band_rows, r_hi) + hr, v);
        int b = 0;
        for (int r0 = r_lo; r0 < r_hi; r0 += a.band_rows, ++b) {
            const int nb = min(a.band_rows, r_hi - r0);
            float *Xs = Xs0 + (b & 1) * (xs_bytes / 4);
            if constexpr (SYNC) det_lap(a, 8);
            land(r0 - hr, r0 + nb + hr, r0 == r_lo ? 0 : r0, r0 + nb == r_hi ? S : r0 + nb, v, Xs);
            if constexpr (SYNC) det_lap(a, 9);
            const int r1 = r0 + a.band_rows;
            if (r1 < r_hi) issue(r1 - hr, min(r1 + a.band_rows, r_hi) + hr, v);        // the next band travels during this one
            __syncthreads();            // band b's magnitudes are in; everybody is past band b - 1 (Vt / Vg, the other Xs)
            if constexpr (SYNC) det_lap(a, 10);
            if (!degenerate) {
                if constexpr (CT) cfar_band_ct<TR, TD, GR, GD>(a, f, r0, nb, nb + 2 * hr, Xs, Vt, Vg, bits, ws, Bf, tid);
                else cfar_band_rt(a, f, r0, nb, Xs, Vt, Vg, bits, ws, Bf, tid);
            }
        }
    }
    __syncthreads();
    if constexpr (SYNC) det_mark_acc(a, 2); else det_mark(a, 2);
    const int st = (degenerate ? DST_DEGENERATE : 0) | (ws[45] ? DST_OVERFLOW : 0) | (ws[44] ? DST_UNDECIDED : 0);
    const int n_spec = min(ws[47], DET_SPEC);
    bool ok = !(st & (DST_DEGENERATE | DST_OVERFLOW));
    if (ok) {
        detect_ant_table(a, ws + 48);
        for (int i = tid; i < a.A; i += DET_NT) tw[i] = a.twA[i];
        __syncthreads();
        ok = detect_finish<SYNC>(a, f, bits, ws, tw, tid, n_spec);
    }
    if (!ok) {                                              // the float64 path decides this frame
        if (tid == 0) {
            a.counts[f] = -1;
            atomicAdd(a.ctl + DCTL_FALLBACK, 1);
        }
    } else if (n_spec > 0 && tid == 0)                      // k_detect_insert adds the cells k_cfar_cell_exact decides positive
        a.flag_frames[atomicAdd(a.ctl + DCTL_FLAG_FRAMES, 1)] = (int)f;
    if constexpr (SYNC) det_mark_acc(a, 4); else det_mark(a, 4);
}

template <int TR, int TD, int GR, int GD>
__global__ __launch_bounds__(DET_NT, 2048 / DET_NT) void k_detect_screen(DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    detect_screen_frame<TR, TD, GR, GD, false>(a, blockIdx.x, smem);
}

template <int TR, int TD, int GR, int GD>
__global__ __launch_bounds__(DET_NT, 2048 / DET_NT) void k_detect_screen_sync(DetectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int sy[2];       // ticket, abort
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) {
            const long long t_wait = a.clk ? (long long)__builtin_amdgcn_s_memtime() : 0;
            const int f = (int)__hip_atomic_fetch_add(a.sy_ctl + CTL_ANG_TICKET, 1u, MMW_RLX_AGENT);
            bool dead = false;
            if (f < a.n_frames) {
                unsigned *cnt = a.sy_frame_cnt + f;
                dead = __hip_atomic_load(a.sy_ctl + CTL_ABORT, MMW_RLX_AGENT) != 0u ||
                       !chain_wait(cnt, (unsigned)a.V, __hip_atomic_load(cnt, MMW_RLX_AGENT), a.sy_ctl, a.sy_timeout, a.sy_naps);
                if (dead) {                     // handed back: the caller's float64 path decides this frame
                    a.counts[f] = -1;
                    atomicAdd(a.ctl + DCTL_FALLBACK, 1);
                }
            }
            sy[0] = f;
            sy[1] = dead ? 1 : 0;
            if (a.clk && blockIdx.x == 0 && f < a.n_frames) {
                a.clk[5] += (long long)__builtin_amdgcn_s_memtime() - t_wait;
                a.clk[7] += 1;
            }
        }
        __syncthreads();        // between the poll and EVERY load of the published bytes; also: the previous frame's LDS is dead
        const int f = __builtin_amdgcn_readfirstlane(sy[0]);
        const bool dead = __builtin_amdgcn_readfirstlane(sy[1]) != 0;
        if (f >= a.n_frames) return;
        if (!dead) detect_screen_frame<TR, TD, GR, GD, true>(a, f, smem);
        __syncthreads();
    }
}

// Frames with undecided cells, once k_cfar_cell_exact has decided them (and the refinement kernels are done with the speculative
// slots): every cell decided positive is INSERTED into the frame's list -- detections, azimuth and elevation indices -- at its
// place in np.where order.  One 256-thread workgroup per flagged frame
// (insertions shift the tail up).
constexpr int INS_NT = 256;
__global__ __launch_bounds__(INS_NT) void k_detect_insert(DetectArgs a) {
    __shared__ int pk[DET_SPEC], paz[DET_SPEC], pel[DET_SPEC], ppos[DET_SPEC], n_pos;
    const int n_flagged = a.ctl[DCTL_FLAG_FRAMES], tid = threadIdx.x, C = a.C;
    for (int e = blockIdx.x; e < n_flagged; e += gridDim.x) {
        const long f = a.flag_frames[e];
        const int *sp = a.spec + f * (2 * DET_SPEC + 1);
        const int n_spec = sp[0], n = a.counts[f];
        const int n_list = n < a.cap ? n : a.cap;           // (n + n_spec <= cap was checked when the slots were taken)
        if (tid == 0) {                                     // the positive cells, sorted by cell index (at most DET_SPEC: insertion sort)
            int m = 0;
            for (int u = 0; u < n_spec; ++u)
                if (sp[1 + DET_SPEC + u]) {
                    const int key = sp[1 + u];
                    const long slot = f * a.cap + (a.cap - 1 - u);
                    const int az = a.az_idx ? a.az_idx[slot] : 0, el = a.el_idx ? a.el_idx[slot] : 0;
                    int q = m++;
                    while (q > 0 && pk[q - 1] > key) {
                        pk[q] = pk[q - 1];
                        paz[q] = paz[q - 1];
                        pel[q] = pel[q - 1];
                        --q;
                    }
                    pk[q] = key;
                    paz[q] = az;
                    pel[q] = el;
                }
            n_pos = m;
        }
        __syncthreads();
        const int m = n_pos;
        if (m > 0 && n >= 0) {
            if (tid < m) {                                  // certain detections in front of positive cell tid (the list is sorted)
                int lo = 0, hi = n_list;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const long slot = f * a.cap + mid;
                    if (a.dets[slot * 2] * C + a.dets[slot * 2 + 1] < pk[tid]) lo = mid + 1;
                    else hi = mid;
                }
                ppos[tid] = lo;
            }
            __syncthreads();
            // the certain entries move up by the number of positive cells in front of them: chunks of INS_NT from the END of the
            // list, every entry of a chunk in a register before any is written (an entry lands inside its own chunk or above)
            for (int hi = n_list; hi > 0; hi -= INS_NT) {
                const int i = hi - INS_NT + tid;
                int r = 0, c = 0, az = 0, el = 0;
                if (i >= 0) {
                    const long slot = f * a.cap + i;
                    r = a.dets[slot * 2];
                    c = a.dets[slot * 2 + 1];
                    if (a.az_idx) az = a.az_idx[slot];
                    if (a.el_idx) el = a.el_idx[slot];
                }
                __syncthreads();
                if (i >= 0) {
                    int up = 0;
                    for (int q = 0; q < m; ++q) up += pk[q] < r * C + c ? 1 : 0;
                    if (up) {
                        const long slot = f * a.cap + i + up;
                        a.dets[slot * 2] = r;
                        a.dets[slot * 2 + 1] = c;
                        if (a.az_idx) a.az_idx[slot] = az;
                        if (a.el_idx) a.el_idx[slot] = el;
                    }
                }
                __syncthreads();
            }
            if (tid < m) {
                const long slot = f * a.cap + ppos[tid] + tid;
                a.dets[slot * 2] = pk[tid] / C;
                a.dets[slot * 2 + 1] = pk[tid] - (pk[tid] / C) * C;
                if (a.az_idx) a.az_idx[slot] = paz[tid];
                if (a.el_idx) a.el_idx[slot] = pel[tid];
            }
            if (tid == 0) a.counts[f] = n + m;
        }
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
    int *spec;                 // [F][2 * DET_SPEC + 1] (DetectArgs::spec): the decision of cell u of frame f goes to [1 + DET_SPEC + u]
    int V, S, C;
    int kind, tr, td, gr, gd, n_train, k_rank;
    double scale;
    const double *ws, *wc;
    const cplx<double> *twS, *twC;
    long long *clk;            // diagnostics (MMW_PHASE_CLOCKS=1)
};

constexpr int CE_NT = 1024, CE_RPT = 7, CE_SG = 8;     // threads per cell; range rows per pass; sample groups per chirp
inline size_t cell_exact_lds(int S, int C, int Wr, int Wd) {
    return ((size_t)Wr * C + S + C) * 16 + ((size_t)S + C + (size_t)Wr * Wd + 8) * 8;
}

// One workgroup per undecided cell; everything on its critical path is spread over the 1024 threads (180 such cells per
// 1250-frame batch run side by side, so the kernel lasts as long as ONE cell takes):
//   step A  lane = (chirp, sample group): a thread asks for its 1 / CE_SG share of the chirp's samples in one go (cold HBM
//           lines), accumulates CE_RPT range rows at a time, the sample groups are added up by lane shuffles (fixed order);
//   step B  four lanes per window cell share its Doppler sum;  the CA sum / OS rank run on the whole workgroup.
__global__ __launch_bounds__(CE_NT) void k_cfar_cell_exact(CellExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = a.S, C = a.C, hr = a.tr + a.gr, hd = a.td + a.gd, Wr = 2 * hr + 1, Wd = 2 * hd + 1, tid = threadIdx.x;
    cplx<double> *Y = reinterpret_cast<cplx<double> *>(smem);        // [Wr][C]
    cplx<double> *twS = Y + (size_t)Wr * C, *twC = twS + S;          // W_S^m, W_C^m
    double *wsl = reinterpret_cast<double *>(twC + C), *wcl = wsl + S, *M = wcl + C, *res = M + Wr * Wd;
    auto mark = [&](int i) {
        if (a.clk && blockIdx.x == 0 && tid == 0) a.clk[i] = (long long)__builtin_amdgcn_s_memtime();
    };
    mark(0);
    int n = *a.n_cells;
    if (n > a.cell_cap) n = a.cell_cap;
    if ((int)blockIdx.x >= n) return;
    for (int i = tid; i < S; i += CE_NT) {
        twS[i] = a.twS[i];
        wsl[i] = a.ws[i];
    }
    for (int i = tid; i < C; i += CE_NT) {
        twC[i] = a.twC[i];
        wcl[i] = a.wc[i];
    }
    __syncthreads();
    mark(1);
    // sample group sg of a chirp takes the samples s = sg, sg + CE_SG, ...: neighbouring lanes then read W_S^(k s) entries
    // k apart (spread over the LDS banks for most k; with contiguous sample blocks all eight sat in one bank and the reads
    // were 290 k of this kernel's 320 k clocks)
    const int sg = tid & (CE_SG - 1);
    auto guard = [&](int o) {
        const int row = o / Wd, j = o - row * Wd;
        return row >= a.tr && row <= a.tr + 2 * a.gr && j >= a.td && j <= a.td + 2 * a.gd;
    };
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        const long f = a.cells[2 * e];
        const int packed = a.cells[2 * e + 1], cell = packed & 0xffffff, u = packed >> 24, r = cell / C, c = cell - r * C;
        const float2 *x = a.cubes + f * a.V * S * C;                 // antenna 0
        for (int ch0 = 0; ch0 < C; ch0 += CE_NT / CE_SG) {           // (uniform trip count: shuffles inside)
            const int ch = ch0 + tid / CE_SG;
            const bool live = ch < C;
            for (int row0 = 0; row0 < Wr; row0 += CE_RPT) {
                // Horner over the lane's samples s = sg + CE_SG u, last one first:
                //   sum_u x_u W^(k (sg + CE_SG u)) = W^(k sg) ( ... (x_last z + x_(last-1)) z + ... + x_0 ),   z = W^(k CE_SG)
                // -- one complex multiply-add (4 FMAs) per sample and range bin, no table read in the loop.  (The form before
                // rotated a twiddle from bin to bin: 8 FMAs per sample and bin, 118 k of the kernel's ~200 k clocks.)
                cplx<double> acc[CE_RPT], z[CE_RPT];
                const int k0 = r - hr + row0;                         // first range bin (inside the plane: valid region)
#pragma unroll
                for (int j = 0; j < CE_RPT; ++j) {
                    acc[j] = cplx<double>{0.0, 0.0};
                    z[j] = twS[(int)(((long)(k0 + j) * CE_SG) % S)];
                }
                constexpr int U = 8;             // samples requested before the first is used (16 spill beside 2 x 7 accumulators)
                const int rounds = ((S + CE_SG - 1) / CE_SG + U - 1) / U;
                for (int rb = rounds - 1; rb >= 0; --rb) {
                    const int s0 = sg + rb * U * CE_SG;
                    float2 xv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u)       // (unconditional, clamped: a guarded load is followed by its own wait)
                        xv[u] = x[(long)min(s0 + u * CE_SG, S - 1) * C + (live ? ch : 0)];
#pragma unroll
                    for (int u = U - 1; u >= 0; --u) {
                        const int sx = s0 + u * CE_SG;
                        if (live && sx < S) {
                            const double w = wsl[sx], xr = (double)xv[u].x * w, xi = (double)xv[u].y * w;
#pragma unroll
                            for (int j = 0; j < CE_RPT; ++j) {
                                const double nr = fma(acc[j].x, z[j].x, fma(-acc[j].y, z[j].y, xr));
                                acc[j].y = fma(acc[j].x, z[j].y, fma(acc[j].y, z[j].x, xi));
                                acc[j].x = nr;
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < CE_RPT; ++j) acc[j] = cmul(acc[j], twS[(int)(((long)(k0 + j) * sg) % S)]);
#pragma unroll
                for (int j = 0; j < CE_RPT; ++j) {
                    for (int d = 1; d < CE_SG; d <<= 1) {             // the sample groups of a chirp sit in adjacent lanes
                        acc[j].x += __shfl_xor(acc[j].x, d, 64);
                        acc[j].y += __shfl_xor(acc[j].y, d, 64);
                    }
                    if (live && sg == 0 && row0 + j < Wr) Y[(row0 + j) * C + ch] = acc[j] * wcl[ch];
                }
            }
        }
        __syncthreads();
        mark(2);
        for (int o0 = 0; o0 < Wr * Wd; o0 += CE_NT / 4) {            // four lanes per window cell
            const int o = o0 + tid / 4, q = tid & 3;
            cplx<double> acc = cplx<double>{0.0, 0.0};
            if (o < Wr * Wd) {
                const int row = o / Wd, j = o - row * Wd;
                int kd = c - hd + j - C / 2;                         // np.fft.fftshift: out[i] = X[(i - C//2) mod C]
                if (kd < 0) kd += C;
                const int per_c = (C + 3) / 4, c_lo = q * per_c, c_hi = min(C, c_lo + per_c);
                int idx = (int)(((long)kd * c_lo) % C);
                for (int ch = c_lo; ch < c_hi; ++ch) {
                    acc = acc + cmul(Y[row * C + ch], twC[idx]);
                    idx += kd;
                    if (idx >= C) idx -= C;
                }
            }
            for (int d = 1; d < 4; d <<= 1) {
                acc.x += __shfl_xor(acc.x, d, 64);
                acc.y += __shfl_xor(acc.y, d, 64);
            }
            if (o < Wr * Wd && q == 0) M[o] = hypot(acc.x, acc.y);
        }
        __syncthreads();
        mark(3);
        const double X = M[hr * Wd + hd];
        if (a.kind == MMW_CFAR_CA) {
            if (tid < 64) {                     // sum of the training cells (any order: float64, ~1e-16 of the threshold)
                double tot = 0.0;
                for (int o = tid; o < Wr * Wd; o += 64) tot += guard(o) ? 0.0 : M[o];
                for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d, 64);
                if (tid == 0) res[0] = tot / (double)a.n_train;
            }
        } else {
            // k-th smallest training cell by rank counting (ties broken by position, as a stable sort would)
            for (int o = tid; o < Wr * Wd; o += CE_NT) {
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
        }
        __syncthreads();
        if (tid == 0) a.spec[f * (2 * DET_SPEC + 1) + 1 + DET_SPEC + u] = X > a.scale * res[0] ? 1 : 0;
        __syncthreads();
        mark(4);
    }
}

}  // namespace mmw
