// CFAR detectors in float64 and ordered compaction of the detection mask.
//
// Thresholds reproduce the reference's arithmetic bit for bit when given the same float64 input:
// the training-cell sums are accumulated in the ORDER NumPy uses for the reference's expressions
// (probed against numpy 2.2.6 in the build container; the pairwise kernel is unchanged since 1.x):
//   CaCFAR2D (detectors/ca_cfar.py:134-136)  np.sum(windows*mask, axis=(2,3)):
//       per window row an 8-accumulator pairwise sum over Wd entries (masked entries are +0.0),
//       rows added sequentially, then / N.
//   CaCFAR1D (ca_cfar.py:51-54)  np.mean(windows[:, mask], axis=1): the fancy-indexed copy is
//       F-ordered, so the reduction is a plain left-to-right sum over [left cells, right cells], / N.
//   Go/SoCFAR1D (go_so_cfar.py:43-55)  np.mean over contiguous views: pairwise sum per side, / num_train.
//   OsCFAR (os_cfar.py:68-73,176-177)  np.partition(...)[k-1] == the k-th smallest: exact selection.
// Decision rule X > T strict; outside the valid region T = +inf, noise = 0 (ca_cfar.py:96-97,144-153).
#pragma once
#include "mmw_ctx.h"

namespace mmw {

// NumPy's pairwise_sum (numpy/core/src/umath/loops_utils.h.src) for n <= 128 * 2^DEPTH.
template <int DEPTH, typename G> __device__ double np_pairwise(G get, int lo, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += get(lo + i);
        return res;
    }
    if (n <= 128 || DEPTH == 0) {
        double r0 = get(lo + 0), r1 = get(lo + 1), r2 = get(lo + 2), r3 = get(lo + 3);
        double r4 = get(lo + 4), r5 = get(lo + 5), r6 = get(lo + 6), r7 = get(lo + 7);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
            r0 += get(lo + i + 0); r1 += get(lo + i + 1); r2 += get(lo + i + 2); r3 += get(lo + i + 3);
            r4 += get(lo + i + 4); r5 += get(lo + i + 5); r6 += get(lo + i + 6); r7 += get(lo + i + 7);
        }
        double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; ++i) res += get(lo + i);
        return res;
    }
    if constexpr (DEPTH > 0) {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise<DEPTH - 1>(get, lo, n2) + np_pairwise<DEPTH - 1>(get, lo + n2, n - n2);
    }
    return 0.0;
}

// order-preserving key of a double (total order, negatives included)
__device__ __forceinline__ unsigned long long f64_key(double v) {
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unkey(unsigned long long k) {
    unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// k-th smallest (1-based) of the values get(0..n) by bitwise bisection on the ordered key.
template <typename G> __device__ double kth_smallest(G get, int n, int k) {
    unsigned long long prefix = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const unsigned long long cand = prefix | (1ull << bit);
        int below = 0;
        for (int i = 0; i < n; ++i) below += (f64_key(get(i)) < cand) ? 1 : 0;
        if (below < k) prefix = cand;
    }
    return f64_unkey(prefix);
}

// Bitonic sort of npad = 256 * EPT (key, position) pairs by key, 256 threads, element e = EPT * t + q held by thread t
// in registers.  Exchange distance j < EPT: inside the thread; EPT <= j < 64 EPT: the partner is lane ^ (j / EPT) of the
// same wave (cross-lane shuffles, no LDS round trip); larger j (3 of the 55 stages at npad = 1024): through the LDS
// arrays with workgroup barriers.  Equal keys may end in either order (positions travel with their keys), which does
// not change any order statistic.  On return lds_key / lds_pos hold the sorted sequence.
template <int EPT>
__device__ __forceinline__ void bitonic_sort_256(unsigned long long (&key)[EPT], unsigned (&pos)[EPT],
                                                 unsigned long long *lds_key, unsigned short *lds_pos) {
    const int t = threadIdx.x, npad = 256 * EPT;
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * EPT) {                                   // partner in another wave
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    lds_key[EPT * t + q] = key[q];
                    lds_pos[EPT * t + q] = (unsigned short)pos[q];
                }
                __syncthreads();
                unsigned long long ok[EPT];
                unsigned op[EPT];
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    ok[q] = lds_key[(EPT * t + q) ^ j];
                    op[q] = lds_pos[(EPT * t + q) ^ j];
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int e = EPT * t + q;
                    const bool take_min = ((e & j) == 0) == ((e & k) == 0);
                    const bool other_less = ok[q] < key[q];
                    if (other_less == take_min && ok[q] != key[q]) {
                        key[q] = ok[q];
                        pos[q] = op[q];
                    }
                }
            } else if (j >= EPT) {                                 // partner in another lane of this wave
                const int lane_mask = j / EPT;
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const unsigned long long ok = __shfl_xor(key[q], lane_mask);
                    const unsigned op = __shfl_xor(pos[q], lane_mask);
                    const int e = EPT * t + q;
                    const bool take_min = ((e & j) == 0) == ((e & k) == 0);
                    const bool other_less = ok < key[q];
                    if (other_less == take_min && ok != key[q]) {
                        key[q] = ok;
                        pos[q] = op;
                    }
                }
            } else {                                               // both elements in this thread (static register indices)
                static_for<2>([&](auto JB) {
                    constexpr int J = 1 << decltype(JB)::value;   // 1, 2
                    if constexpr (J < EPT) {
                        if (j == J) {
                            static_for<EPT>([&](auto Q) {
                                constexpr int q = decltype(Q)::value, r = q ^ J;
                                if constexpr (r > q) {
                                    const int e = EPT * t + q;
                                    const bool up = (e & k) == 0;
                                    if ((key[q] > key[r]) == up) {
                                        const unsigned long long tk = key[q];
                                        key[q] = key[r];
                                        key[r] = tk;
                                        const unsigned tp = pos[q];
                                        pos[q] = pos[r];
                                        pos[r] = tp;
                                    }
                                }
                            });
                        }
                    }
                });
            }
        }
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        lds_key[EPT * t + q] = key[q];
        lds_pos[EPT * t + q] = (unsigned short)pos[q];
    }
    __syncthreads();
}

struct Cfar2dArgs {
    const double *X;
    double *thr, *noise;
    uint8_t *mask;
    int R, D;
    int kind;
    int tr, td, gr, gd;
    double scale;
    int k_rank;
    int os_fast;   // OS only: LDS holds the 15 coarse integral images (tile <= 1024 cells)
};

constexpr int CFAR_TR = 16, CFAR_TC = 16;
constexpr int OS_COARSE = 16;   // coarse rank buckets; the bucket width npad/16 <= 64 fits one 64-bit mask

__global__ __launch_bounds__(CFAR_TR *CFAR_TC) void k_cfar2d(Cfar2dArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tile = reinterpret_cast<double *>(smem);
    const int hr = p.tr + p.gr, hd = p.td + p.gd;
    const int TW = CFAR_TC + 2 * hd, TH = CFAR_TR + 2 * hr;
    double *rs_full = tile + TW * TH;                 // [TH][CFAR_TC] pairwise row sums of the full window row
    double *rs_mask = rs_full + TH * CFAR_TC;         // same with the guard columns replaced by +0.0
    const long plane = (long)p.R * p.D;
    const double *X = p.X + (long)blockIdx.z * plane;
    const int r0 = blockIdx.y * CFAR_TR, c0 = blockIdx.x * CFAR_TC;
    {   // tile + halo, 2-D thread mapping (no per-element division)
        const int ty = threadIdx.x / CFAR_TC, tx = threadIdx.x % CFAR_TC;
        for (int y = ty; y < TH; y += CFAR_TR) {
            const int rr = r0 - hr + y;
            const bool row_ok = rr >= 0 && rr < p.R;
            for (int x = tx; x < TW; x += CFAR_TC) {
                const int cc = c0 - hd + x;
                tile[y * TW + x] = (row_ok && cc >= 0 && cc < p.D) ? X[(long)rr * p.D + cc] : 0.0;
            }
        }
    }
    __syncthreads();
    const int Wr = 2 * hr + 1, Wd = 2 * hd + 1;
    if (p.kind == MMW_CFAR_CA) {
        // Every CUT of a column shares the per-row sums of its window rows: compute each (tile row, window
        // start column) pair once, in NumPy's pairwise order, then add Wr of them in row order per CUT.
        const int z0 = p.td, z1 = p.td + 2 * p.gd;        // guard columns of the window
        for (int row = threadIdx.x / CFAR_TC; row < TH; row += CFAR_TR) {
            const int col = threadIdx.x % CFAR_TC, t = row * CFAR_TC + col;
            const double *src = tile + row * TW + col;
            double sf, sm;
            if (Wd > 128) {                               // rare: generic pairwise recursion
                sf = np_pairwise<2>([&](int wd) { return src[wd]; }, 0, Wd);
                sm = np_pairwise<2>([&](int wd) { return (wd >= z0 && wd <= z1) ? 0.0 : src[wd]; }, 0, Wd);
            } else if (Wd < 8) {
                sf = sm = 0.0;
                for (int i = 0; i < Wd; ++i) {
                    const double v = src[i];
                    sf += v;
                    sm += (i >= z0 && i <= z1) ? 0.0 : v;
                }
            } else {                                      // NumPy's 8-accumulator block, both sums in one sweep
                double f[8], m[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double v = src[j];
                    f[j] = v;
                    m[j] = (j >= z0 && j <= z1) ? 0.0 : v;
                }
                int i = 8;
                for (; i < Wd - (Wd % 8); i += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const double v = src[i + j];
                        f[j] += v;
                        m[j] += (i + j >= z0 && i + j <= z1) ? 0.0 : v;
                    }
                }
                sf = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
                sm = ((m[0] + m[1]) + (m[2] + m[3])) + ((m[4] + m[5]) + (m[6] + m[7]));
                for (; i < Wd; ++i) {
                    const double v = src[i];
                    sf += v;
                    sm += (i >= z0 && i <= z1) ? 0.0 : v;
                }
            }
            rs_full[t] = sf;
            rs_mask[t] = sm;
        }
        __syncthreads();
    }
    // OS: rank every tile element once per workgroup (bitonic sort of (ordered key, position) pairs in LDS)
    unsigned long long *os_key = reinterpret_cast<unsigned long long *>(rs_full);
    int os_bits = 0;
    while ((1 << os_bits) < TW * TH) ++os_bits;
    const int npad = 1 << os_bits;
    unsigned short *os_pos = reinterpret_cast<unsigned short *>(os_key + npad);
    unsigned short *os_rank = os_pos + npad;
    unsigned short *os_integ = os_rank + ((TW * TH + 1) & ~1);
    if (p.kind == MMW_CFAR_OS) {
        const int NT = CFAR_TR * CFAR_TC;
        auto load_sort = [&](auto E) {
            constexpr int EPT = decltype(E)::value;
            unsigned long long key[EPT];
            unsigned pos[EPT];
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int e = EPT * threadIdx.x + q;
                key[q] = e < TW * TH ? f64_key(tile[e]) : ~0ull;
                pos[q] = (unsigned)e;
            }
            bitonic_sort_256<EPT>(key, pos, os_key, os_pos);
        };
        if (npad == 1024) load_sort(std::integral_constant<int, 4>{});
        else if (npad == 512) load_sort(std::integral_constant<int, 2>{});
        else if (npad == 256) load_sort(std::integral_constant<int, 1>{});
        else {
            for (int t = threadIdx.x; t < npad; t += NT) {
                os_key[t] = t < TW * TH ? f64_key(tile[t]) : ~0ull;
                os_pos[t] = (unsigned short)t;
            }
            __syncthreads();
            // Compare-exchange t touches elements i and i | j.  For j <= 64 the 64 exchanges of one wave iteration stay
            // inside one aligned 128-element block that no other wave touches in that stage, and a wave's LDS accesses
            // complete in program order, so those 45 of the 55 stages (npad = 1024) need no workgroup barrier -- only a
            // wavefront-scope fence to keep the compiler from carrying values across stages in registers.
            bool need_barrier = false;
            for (int k = 2; k <= npad; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    if (j > 64 || need_barrier) __syncthreads();
                    else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    need_barrier = j > 64;          // the stage after a cross-wave stage must see every wave's writes
                    for (int t = threadIdx.x; t < npad / 2; t += NT) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), q = i | j;
                        const unsigned long long ka = os_key[i], kb = os_key[q];
                        const unsigned short pa = os_pos[i], pb = os_pos[q];
                        const bool a_gt_b = ka > kb || (ka == kb && pa > pb);
                        if (a_gt_b == ((i & k) == 0)) {
                            os_key[i] = kb; os_key[q] = ka;
                            os_pos[i] = pb; os_pos[q] = pa;
                        }
                    }
                }
            }
            __syncthreads();
        }
        for (int t = threadIdx.x; t < npad; t += NT) {
            const int pos = os_pos[t];
            if (pos < TW * TH) os_rank[pos] = (unsigned short)t;
            // position of rank t as (row << 8 | column); padding entries can never fall inside a window
            os_pos[t] = pos < TW * TH ? (unsigned short)(((pos / TW) << 8) | (pos % TW)) : (unsigned short)0xFFFF;
        }
        __syncthreads();
        if (p.os_fast) {
            // integral images of [rank < j * npad/16], j = 1..15, with a zero border row / column
            const int IW = TW + 1, IH = TH + 1, isz = IW * IH, bucket = npad / OS_COARSE;
            for (int t = threadIdx.x; t < (OS_COARSE - 1) * IH; t += NT) {        // row prefix sums
                const int j = t / IH + 1, y = t % IH;
                unsigned short *row = os_integ + (j - 1) * isz + y * IW;
                row[0] = 0;
                unsigned short run = 0;
                for (int x = 1; x < IW; ++x) {
                    if (y > 0) run += os_rank[(y - 1) * TW + (x - 1)] < j * bucket ? 1 : 0;
                    row[x] = run;
                }
            }
            __syncthreads();
            for (int t = threadIdx.x; t < (OS_COARSE - 1) * TW; t += NT) {        // column prefix sums
                const int j = t / TW + 1, x = t % TW + 1;
                unsigned short *col = os_integ + (j - 1) * isz + x;
                unsigned short run = 0;
                for (int y = 1; y < IH; ++y) {
                    run += col[y * IW];
                    col[y * IW] = run;
                }
            }
            __syncthreads();
        }
    }
    const int lr = threadIdx.x / CFAR_TC, lc = threadIdx.x % CFAR_TC;
    const int r = r0 + lr, c = c0 + lc;
    if (r >= p.R || c >= p.D) return;
    const long o = (long)blockIdx.z * plane + (long)r * p.D + c;
    double thr = INFINITY, est = 0.0;
    const bool valid = r >= hr && r < p.R - hr && c >= hd && c < p.D - hd;
    if (valid) {
        const int ntrain = Wr * Wd - (2 * p.gr + 1) * (2 * p.gd + 1);
        if (p.kind == MMW_CFAR_CA) {
            double sum = 0.0;
            for (int wr = 0; wr < Wr; ++wr) {
                const bool guard_row = wr >= p.tr && wr <= p.tr + 2 * p.gr;
                sum += (guard_row ? rs_mask : rs_full)[(lr + wr) * CFAR_TC + lc];
            }
            est = sum / (double)ntrain;
        } else {
            // OS: the k-th smallest training cell == the training cell with the k-th smallest tile rank.
            int prefix = 0;
            if (p.os_fast) {
                // (1) coarse bucket from the integral images: 8 two-byte reads per probe, 4 probes
                const int IW = TW + 1, isz = IW * (TH + 1), bucket = npad / OS_COARSE;
                const int wy0 = lr, wx0 = lc, wy1 = lr + Wr, wx1 = lc + Wd;                       // window, exclusive end
                const int gy0 = lr + p.tr, gx0 = lc + p.td, gy1 = gy0 + 2 * p.gr + 1, gx1 = gx0 + 2 * p.gd + 1;
                auto count_below = [&](int j) {
                    const unsigned short *I = os_integ + (j - 1) * isz;
                    const int win = I[wy1 * IW + wx1] - I[wy0 * IW + wx1] - I[wy1 * IW + wx0] + I[wy0 * IW + wx0];
                    const int grd = I[gy1 * IW + gx1] - I[gy0 * IW + gx1] - I[gy1 * IW + gx0] + I[gy0 * IW + gx0];
                    return win - grd;
                };
                int j = 0, below = 0;
                for (int step = OS_COARSE / 2; step >= 1; step >>= 1) {
                    const int c = count_below(j + step);
                    if (c < p.k_rank) {
                        j += step;
                        below = c;
                    }
                }
                // (2) the answer is the (k - below)-th smallest rank inside [j*bucket, (j+1)*bucket).  The sorted order
                //     knows where each of those <= 64 ranks sits in the tile (os_pos, repacked as y << 8 | x), so walk
                //     them in rank order and stop at the q-th one that is a training cell of this window -- about a
                //     quarter of the tile is, so this takes ~4q steps instead of a sweep of the whole window.
                const int base = j * bucket;
                int q = p.k_rank - below;
                prefix = base;
                for (int d = 0; d < bucket; ++d) {
                    const unsigned yx = os_pos[base + d];
                    const unsigned y = yx >> 8, x = yx & 255u;
                    const bool in_win = (y - (unsigned)wy0) < (unsigned)Wr && (x - (unsigned)wx0) < (unsigned)Wd;
                    const bool in_grd = (y - (unsigned)gy0) < (unsigned)(2 * p.gr + 1) && (x - (unsigned)gx0) < (unsigned)(2 * p.gd + 1);
                    if (in_win && !in_grd && --q == 0) {
                        prefix = base + d;
                        break;
                    }
                }
            } else {
            // Bisection over the log2(npad) rank bits instead of 64 key bits, on 2-byte LDS reads.
            for (int bit = os_bits - 1; bit >= 0; --bit) {
                const int cand = prefix | (1 << bit);
                int below = 0;
                for (int wr = 0; wr < Wr; ++wr) {
                    const unsigned short *rrow = os_rank + (lr + wr) * TW + lc;
                    const bool guard_row = wr >= p.tr && wr <= p.tr + 2 * p.gr;
                    if (!guard_row) {
                        for (int wd = 0; wd < Wd; ++wd) below += rrow[wd] < cand ? 1 : 0;
                    } else {
                        for (int wd = 0; wd < p.td; ++wd) below += rrow[wd] < cand ? 1 : 0;
                        for (int wd = p.td + 2 * p.gd + 1; wd < Wd; ++wd) below += rrow[wd] < cand ? 1 : 0;
                    }
                }
                if (below < p.k_rank) prefix = cand;
            }
            }
            est = f64_unkey(os_key[prefix]);
        }
        thr = p.scale * est;
    }
    if (p.thr) p.thr[o] = thr;
    if (p.noise) p.noise[o] = est;
    if (p.mask) p.mask[o] = (tile[(lr + hr) * TW + lc + hd] > thr) ? 1 : 0;
}

// CA-CFAR 2-D on a larger tile.  k_cfar2d's 16 x 16 tile suits the OS path (the tile + halo must fit a 1024-element sort);
// for CA it means a 28 x 28 tile + halo for the reference's (4,4)/(2,2) window -- 3.06 input cells loaded per output -- and
// 128 short-lived workgroups per 256 x 128 plane.  Here a 256-thread workgroup owns CA_TR x CA_TC = 32 x 32 cells (halo
// factor 1.9, a quarter of the workgroups): phase 1 computes, for every (tile row, window start column), the row sum of
// the full window row and of the guard-masked one in NumPy's 8-accumulator pairwise order (each shared by the Wr cells
// under test of that column); phase 2 adds Wr of them per cell in row order, / N, x alpha.  Same arithmetic, same
// order, bit-identical thresholds.
constexpr int CA_TR = 32, CA_TC = 32;
__global__ __launch_bounds__(256) void k_cfar2d_ca(Cfar2dArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tile = reinterpret_cast<double *>(smem);
    const int hr = p.tr + p.gr, hd = p.td + p.gd;
    const int TW = CA_TC + 2 * hd, TH = CA_TR + 2 * hr;
    double *rs_full = tile + TW * TH;                 // [TH][CA_TC]
    double *rs_mask = rs_full + TH * CA_TC;
    const long plane = (long)p.R * p.D;
    const double *X = p.X + (long)blockIdx.z * plane;
    const int r0 = blockIdx.y * CA_TR, c0 = blockIdx.x * CA_TC;
    for (int e = threadIdx.x; e < TH * TW; e += 256) {
        const int y = e / TW, x = e - y * TW;
        const int rr = r0 - hr + y, cc = c0 - hd + x;
        tile[e] = (rr >= 0 && rr < p.R && cc >= 0 && cc < p.D) ? X[(long)rr * p.D + cc] : 0.0;
    }
    __syncthreads();
    const int Wr = 2 * hr + 1, Wd = 2 * hd + 1;
    const int z0 = p.td, z1 = p.td + 2 * p.gd;        // guard columns of the window
    for (int t = threadIdx.x; t < TH * CA_TC; t += 256) {
        const int row = t / CA_TC, col = t - row * CA_TC;
        const double *src = tile + row * TW + col;
        double sf, sm;
        if (Wd > 128) {                               // rare: generic pairwise recursion
            sf = np_pairwise<2>([&](int wd) { return src[wd]; }, 0, Wd);
            sm = np_pairwise<2>([&](int wd) { return (wd >= z0 && wd <= z1) ? 0.0 : src[wd]; }, 0, Wd);
        } else if (Wd < 8) {
            sf = sm = 0.0;
            for (int i = 0; i < Wd; ++i) {
                const double v = src[i];
                sf += v;
                sm += (i >= z0 && i <= z1) ? 0.0 : v;
            }
        } else {                                      // NumPy's 8-accumulator block, both sums in one sweep
            double f[8], m[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double v = src[j];
                f[j] = v;
                m[j] = (j >= z0 && j <= z1) ? 0.0 : v;
            }
            int i = 8;
            for (; i < Wd - (Wd % 8); i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double v = src[i + j];
                    f[j] += v;
                    m[j] += (i + j >= z0 && i + j <= z1) ? 0.0 : v;
                }
            }
            sf = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
            sm = ((m[0] + m[1]) + (m[2] + m[3])) + ((m[4] + m[5]) + (m[6] + m[7]));
            for (; i < Wd; ++i) {
                const double v = src[i];
                sf += v;
                sm += (i >= z0 && i <= z1) ? 0.0 : v;
            }
        }
        rs_full[t] = sf;
        rs_mask[t] = sm;
    }
    __syncthreads();
    const int ntrain = Wr * Wd - (2 * p.gr + 1) * (2 * p.gd + 1);
    for (int cell = threadIdx.x; cell < CA_TR * CA_TC; cell += 256) {
        const int lr = cell / CA_TC, lc = cell - lr * CA_TC;
        const int r = r0 + lr, c = c0 + lc;
        if (r >= p.R || c >= p.D) continue;
        const long o = (long)blockIdx.z * plane + (long)r * p.D + c;
        double thr = INFINITY, est = 0.0;
        if (r >= hr && r < p.R - hr && c >= hd && c < p.D - hd) {
            double sum = 0.0;
            for (int wr = 0; wr < Wr; ++wr) {
                const bool guard_row = wr >= p.tr && wr <= p.tr + 2 * p.gr;
                sum += (guard_row ? rs_mask : rs_full)[(lr + wr) * CA_TC + lc];
            }
            est = sum / (double)ntrain;
            thr = p.scale * est;
        }
        if (p.thr) p.thr[o] = thr;
        if (p.noise) p.noise[o] = est;
        if (p.mask) p.mask[o] = (tile[(lr + hr) * TW + lc + hd] > thr) ? 1 : 0;
    }
}

// OS-CFAR when only the detection mask is wanted (mmw_detect_batch, FramePipeline): no selection at all.
//   X > alpha * T*,  T* = k-th smallest training cell   <=>   #{ training cells t : alpha * t < X } >= k
// because t -> alpha * t (one float64 multiply, the very product the reference forms for T*) is monotone, so the k-th
// smallest of the products is alpha * T*, and "the k-th smallest is below X" means "at least k are below X".  NaNs
// compare false and so count as largest, as np.partition orders them (os_cfar.py:176-177); a NaN cell under test never
// fires.  Each cell under test therefore needs ONE count over its window with its own threshold: ~220 LDS reads and
// compares instead of a 1024-element sort shared by 256 cells, integral images and a rank walk (6.1 -> under 1.5 us per
// 256 x 128 frame for the GUI's (5,5)/(3,2) window).  The threshold / noise arrays the single-frame API also returns
// still come from k_cfar2d's exact selection.
constexpr int OSM_TR = 16, OSM_TC = 16;
__global__ __launch_bounds__(OSM_TR *OSM_TC) void k_cfar2d_os_mask(Cfar2dArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tile = reinterpret_cast<double *>(smem);          // alpha * X over the tile + halo, row pitch TWp
    const int hr = p.tr + p.gr, hd = p.td + p.gd;
    const int TW = OSM_TC + 2 * hd, TH = OSM_TR + 2 * hr;
    // row pitch = 16 (mod 32) doubles: the two rows a 32-lane group reads land on disjoint banks
    const int TWp = ((TW + 15) / 32) * 32 + 16;
    const long plane = (long)p.R * p.D;
    const double *X = p.X + (long)blockIdx.z * plane;
    const int r0 = blockIdx.y * OSM_TR, c0 = blockIdx.x * OSM_TC;
    const int ty = threadIdx.x / OSM_TC, tx = threadIdx.x % OSM_TC;
    for (int y = ty; y < TH; y += OSM_TR) {
        const int rr = r0 - hr + y;
        const bool row_ok = rr >= 0 && rr < p.R;
        for (int x = tx; x < TW; x += OSM_TC) {
            const int cc = c0 - hd + x;
            tile[y * TWp + x] = (row_ok && cc >= 0 && cc < p.D) ? p.scale * X[(long)rr * p.D + cc] : 0.0;
        }
    }
    __syncthreads();
    const int r = r0 + ty, c = c0 + tx;
    if (r >= p.R || c >= p.D) return;
    const bool valid = r >= hr && r < p.R - hr && c >= hd && c < p.D - hd;
    bool det = false;
    if (valid) {
        const double x = X[(long)r * p.D + c];
        const int Wr = 2 * hr + 1, Wd = 2 * hd + 1, g0 = p.td, g1 = p.td + 2 * p.gd + 1;
        int below = 0;
#pragma unroll 1
        for (int wr = 0; wr < Wr; ++wr) {
            const double *row = tile + (ty + wr) * TWp + tx;
            const bool guard_row = wr >= p.tr && wr <= p.tr + 2 * p.gr;
            if (!guard_row) {
#pragma unroll 5
                for (int wd = 0; wd < Wd; ++wd) below += row[wd] < x ? 1 : 0;
            } else {
                for (int wd = 0; wd < g0; ++wd) below += row[wd] < x ? 1 : 0;
                for (int wd = g1; wd < Wd; ++wd) below += row[wd] < x ? 1 : 0;
            }
        }
        det = below >= p.k_rank;
    }
    p.mask[(long)blockIdx.z * plane + (long)r * p.D + c] = det ? 1 : 0;
}

// The same count with FOUR vertically adjacent cells per thread and the window shape at compile time.  k_cfar2d_os_mask is
// bound by the LDS (220 eight-byte reads per cell: 14 of the CU's 16 doubles per clock); the windows of four cells in a column
// share all but three of their rows, so a thread reads (2 hr + 4) x (2 hd + 1) values once and compares each with the cells
// whose training set holds it: 75 reads per cell instead of 220, the kernel is then bound by its 2 x 220 compare / add-carry
// instructions per cell.  32 x 32 cells per 256-thread workgroup (a half-wave reads 32 consecutive doubles: conflict-free).
constexpr int OSV_T = 32, OSV_NV = 4;
template <int TR, int TD, int GR, int GD>
__global__ __launch_bounds__(256) void k_cfar2d_os_mask_v(Cfar2dArgs p) {
    constexpr int HR = TR + GR, HD = TD + GD, WR = 2 * HR + 1, WD = 2 * HD + 1;
    constexpr int TW = OSV_T + 2 * HD, TH = OSV_T + 2 * HR;
    __shared__ double tile[TH * TW];                            // alpha * X over the tile + halo
    const long plane = (long)p.R * p.D;
    const double *X = p.X + (long)blockIdx.z * plane;
    const int r0 = blockIdx.y * OSV_T, c0 = blockIdx.x * OSV_T;
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
        const int y = i / TW, x = i - y * TW, rr = r0 - HR + y, cc = c0 - HD + x;
        tile[i] = (rr >= 0 && rr < p.R && cc >= 0 && cc < p.D) ? p.scale * X[(long)rr * p.D + cc] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = c0 + tx, rb = r0 + ty * OSV_NV;
    if (c >= p.D || rb >= p.R) return;
    double x[OSV_NV];
    int below[OSV_NV];
#pragma unroll
    for (int j = 0; j < OSV_NV; ++j) {
        const int r = rb + j < p.R ? rb + j : p.R - 1;
        x[j] = X[(long)r * p.D + c];
        below[j] = 0;
    }
    const double *base = tile + (ty * OSV_NV) * TW + tx;
    // a run-time loop over the tile rows (fully unrolled, the compiler issues all 300 reads first and spills); which cells a
    // row belongs to, and whether as a guard row, are wave-uniform branches
#pragma unroll 1
    for (int y = 0; y < WR + OSV_NV - 1; ++y) {
        double v[WD];
#pragma unroll
        for (int wd = 0; wd < WD; ++wd) v[wd] = base[y * TW + wd];
        static_for<OSV_NV>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int wr = y - j;                               // window row of cell j
            if (wr >= 0 && wr < WR) {
                if (wr >= TR && wr <= TR + 2 * GR) {
                    static_for<WD>([&](auto Wd) {
                        constexpr int wd = decltype(Wd)::value;
                        if constexpr (wd < TD || wd > TD + 2 * GD) below[j] += v[wd] < x[j] ? 1 : 0;
                    });
                } else {
#pragma unroll
                    for (int wd = 0; wd < WD; ++wd) below[j] += v[wd] < x[j] ? 1 : 0;
                }
            }
        });
    }
#pragma unroll
    for (int j = 0; j < OSV_NV; ++j) {
        const int r = rb + j;
        if (r >= p.R) break;
        const bool valid = r >= HR && r < p.R - HR && c >= HD && c < p.D - HD;
        p.mask[(long)blockIdx.z * plane + (long)r * p.D + c] = (valid && below[j] >= p.k_rank) ? 1 : 0;
    }
}

struct Cfar1dArgs {
    const double *x;
    double *thr, *noise;
    uint8_t *mask;
    int n_rows, L;
    int kind, T, G;
    double scale;
    int k_rank;
};

__global__ __launch_bounds__(256) void k_cfar1d(Cfar1dArgs p) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)p.n_rows * p.L) return;
    const int i = (int)(gid % p.L);
    const double *x = p.x + (gid - i);
    const int half = p.T + p.G;
    double thr = INFINITY, est = 0.0;
    if (i >= half && i < p.L - half) {
        const double *w = x + (i - half);          // window start
        const int rs = p.T + 2 * p.G + 1;            // start of the right training cells
        if (p.kind == MMW_CFAR_CA) {
            double s = 0.0;
            for (int q = 0; q < p.T; ++q) s += w[q];
            for (int q = 0; q < p.T; ++q) s += w[rs + q];
            est = s / (double)(2 * p.T);
        } else if (p.kind == MMW_CFAR_OS) {
            est = kth_smallest([&](int q) { return q < p.T ? w[q] : w[rs + q - p.T]; }, 2 * p.T, p.k_rank);
        } else {
            const double ml = np_pairwise<2>([&](int q) { return w[q]; }, 0, p.T) / (double)p.T;
            const double mr = np_pairwise<2>([&](int q) { return w[rs + q]; }, 0, p.T) / (double)p.T;
            est = (p.kind == MMW_CFAR_GO) ? fmax(ml, mr) : fmin(ml, mr);
        }
        thr = p.scale * est;
    }
    if (p.thr) p.thr[gid] = thr;
    if (p.noise) p.noise[gid] = est;
    if (p.mask) p.mask[gid] = (x[i] > thr) ? 1 : 0;
}

// Ordered compaction: one workgroup per frame, each thread owns a contiguous run of cells so the
// emitted (row, col) list is in np.where's row-major order (detectors/base.py:229-230).
__global__ __launch_bounds__(1024) void k_compact2d(const uint8_t *mask, int32_t *dets, int32_t *counts,
                                                    int R, int D, int cap) {
    __shared__ int wave_tot[16];
    __shared__ int wave_off[16];
    const long cells = (long)R * D;
    const uint8_t *m = mask + (long)blockIdx.x * cells;
    const int per = (int)((cells + 1023) / 1024);
    const long lo = (long)threadIdx.x * per;
    const long hi = lo + per < cells ? lo + per : cells;
    int cnt = 0;
    for (long i = lo; i < hi; ++i) cnt += m[i] ? 1 : 0;
    // wave-level inclusive scan, then scan of the 16 wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < 16; ++w) {
            wave_off[w] = run;
            run += wave_tot[w];
        }
        counts[blockIdx.x] = run;
    }
    __syncthreads();
    int pos = wave_off[wave] + incl - cnt;
    int32_t *out = dets + (long)blockIdx.x * cap * 2;
    for (long i = lo; i < hi; ++i) {
        if (m[i]) {
            if (pos < cap) {
                out[2 * pos] = (int32_t)(i / D);
                out[2 * pos + 1] = (int32_t)(i % D);
            }
            ++pos;
        }
    }
}

}  // namespace mmw
