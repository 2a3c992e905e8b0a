// Fast paths for the headline shape (V<=16 virtual antennas, A=64 angle bins; S x C = 256 x 128).
//
// k_angle64: last stage of the 3-D chain (processors/range_angle_resp_dbs_enhanced.py:175-196).
//   HBM-write bound: reads V*8 B and writes 64*8 B per range-Doppler bin (16.8 MB of the
//   19.9 MB/frame the chain moves), so the kernel is built around full-width streaming stores:
//   one thread owns two adjacent chirp bins, lanes walk the contiguous chirp index, every load is a
//   16-B-per-lane 1-KiB wave access and every store a 1-KiB wave access.  No LDS.
//   The zero-padded 64-point DFT is evaluated as 8 x 8 with the zero rows pruned:
//     n = 8*n1 + n2 (only n1 in {0,1} can be non-zero for V <= 16), k = k1 + 8*k2
//     Y[k1][n2] = x[n2] + x[n2+8] * W8^k1,   X[k1+8*k2] = FFT8_{n2}( Y[k1][n2] * W64^(n2*k1) )
//   with every twiddle a compile-time constant.
#pragma once
#include "mmw_ctx.h"

namespace mmw {

struct AngleWin {
    float h[16];
};


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ device-side hand-off of the chain
// One range-Doppler launch and one angle launch per chain call run side by side on their CU-masked queues and
// synchronise through counters in device memory instead of one pair of launches + four event packets per chunk of
// frames (which left the angle queue idle ~15 us per launch).  The RD cube of a frame lives in slot g % ring of a ring
// of frames (g = frames since the layout was set up, monotone across calls); per slot two monotone counters:
//   rd_cnt[s]  += 1 by every RD workgroup after its plane is stored       (angle waits for (use + 1) * v_live)
//   ang_cnt[s] += 1 by every angle workgroup after its tile is loaded     (RD waits for use * tiles before overwriting)
// Work is handed out by tickets (atomic counters), so whatever order workgroups become resident in, the lowest
// unfinished item of either kind is always held by a running workgroup: no deadlock by construction.
// Visibility (MI355X_MICROARCH.md, inter-workgroup hand-off table, counter row): every store of the RD cube is an
// `sc1` (write-through) store, every storing wave drains `vmcnt(0)`, workgroup barrier, ONE lane adds to the counter
// (agent-scope atomic); the consumer polls with an `sc1` load by one lane, workgroup barrier, then EVERY load of the
// cube is an `sc1` load (L1 bypass).  Spins are bounded: a timeout sets ctl[CTL_ABORT] and everybody leaves.
constexpr int CTL_RD_TICKET = 0, CTL_ANG_TICKET = 1, CTL_ABORT = 2, CTL_CNT = 32, CTL_RING_MAX = 256;
constexpr int CTL_WORDS = CTL_CNT + 2 * CTL_RING_MAX;
struct ChainSync {
    unsigned *ctl;              // CTL_WORDS words: tickets, abort flag, rd_cnt[CTL_RING_MAX], ang_cnt[CTL_RING_MAX]
    unsigned rd_base, ang_base; // ticket counter values at the start of this call
    unsigned s0, u0;            // ring slot and use index of the call's frame 0
    int ring;                   // frames in the ring
    int V, vskip, v_live;       // planes per frame; vskip = V when the end planes are skipped, else 0; planes transformed
    int tiles;                  // angle work items per frame
    int n_frames;
    unsigned long long timeout; // s_memrealtime ticks (100 MHz) a spin may last
    int naps_rd, naps_ang;      // s_sleep(32) calls between two polls of a counter (RD runs a ring ahead: long naps are free)
    unsigned long long vmap;    // 16 x 4 bits: live-plane index within a frame -> virtual antenna (order of the RD work items);
                                // a packed word, not an array: indexing an array in the argument block spills it to scratch
    int ntx, nrx;               // ntx > 1: the RD input is the raw [F][nrx][S][ntx * C] cube
    int i16;                    // ... of int16 (I, Q) cells (256 x 128 producer only)
    unsigned *frame_cnt;        // DET producers: [n_frames] planes published per frame (the detection stage's hand-off, mmw_detect.h)
};
#define MMW_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
// one lane: wait until (int)(*cnt - target) >= 0; false on timeout / abort (and the abort flag is raised).
// Polls are sc1 loads that go to the memory side every time, so they are spaced out (s_sleep) -- hundreds of
// workgroups hammering one counter's memory channel slow the whole chip down.
__device__ __forceinline__ bool chain_wait(unsigned *cnt, unsigned target, unsigned first, unsigned *ctl, unsigned long long timeout,
                                           int naps) {
    unsigned c = first;
    if ((int)(c - target) >= 0) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 1;; ++spin) {
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(32);      // 32 x 64 cycles ~ 0.9 us each
        c = __hip_atomic_load(cnt, MMW_RLX_AGENT);
        if ((int)(c - target) >= 0) return true;
        if ((spin & 15) == 0) {
            if (__hip_atomic_load(ctl + CTL_ABORT, MMW_RLX_AGENT)) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {
                __hip_atomic_store(ctl + CTL_ABORT, 1u, MMW_RLX_AGENT);
                return false;
            }
        }
    }
}
// The eight pruned k1 passes + stores of one thread's two bins (shared by k_angle64 and k_angle64_sync).
// Output rows that do not start on 64-B boundaries: alignment unit in cells (complex64: 8, float32 magnitude: 16).
// Measured (tools/angle_shape.py, plain kernel): rows aligned to 128 B / 64 B / 32 B / 16 B give 5.1 / 5.4 / 4.7 / 4.1 TB/s
// for complex and 5.5 / 5.2 / 4.2 / 3.8 TB/s for magnitude output -- 64 B is enough.
constexpr int angle_rows_unit(bool mag) { return mag ? 16 : 8; }
// ROWS 1 (k_angle64_rows): the lane stores row a only if its pair lies in that row's 64-B aligned window of the wave.
// ROWS 2 (k_angle64_rows_odd): odd bin count -- `pairs_per_frame` is the BIN count, `pair` the lane's first cell c0, and
// rows that start at an odd cell offset store (second cell, next lane's first cell) so that every 16-B store is aligned.
template <int VIN, bool MAG, bool NT, int ROWS = 0>
__device__ __forceinline__ void angle64_passes(cplx<float> (&xa)[VIN], cplx<float> (&xb)[VIN], void *__restrict__ out, long f,
                                               long pairs_per_frame, long pair, int shift_off, int b16 = 0, int lane = 0,
                                               bool pair_ok = true, bool second_ok = true, bool next_ok = true) {
    typedef cplx<float> C;
    static_for<8>([&](auto K1) {
        constexpr int k1 = decltype(K1)::value;
        C za[8], zb[8];
        static_for<8>([&](auto N2) {
            constexpr int n2 = decltype(N2)::value;
            C ya = C{0.f, 0.f}, yb = C{0.f, 0.f};
            if constexpr (n2 < VIN) {
                ya = xa[n2];
                yb = xb[n2];
            }
            if constexpr (n2 + 8 < VIN) {
                ya = ya + mul_w<8, k1, float, C>(xa[n2 + 8]);
                yb = yb + mul_w<8, k1, float, C>(xb[n2 + 8]);
            }
            za[n2] = mul_w<64, n2 * k1, float, C>(ya);
            zb[n2] = mul_w<64, n2 * k1, float, C>(yb);
        });
        RegFFT<8, float, 8, 0, C>::run(za);
        RegFFT<8, float, 8, 0, C>::run(zb);
        static_for<8>([&](auto K2) {
            constexpr int k2 = decltype(K2)::value;
            const int a = (k1 + 8 * k2 + shift_off) & 63;   // fftshift over the angle axis (shift_off = 32)
            const C va = za[bitrev<8>(k2)], vb = zb[bitrev<8>(k2)];
            const long o = (f * 64 + a) * pairs_per_frame + pair;
            if constexpr (ROWS == 2) {
                static_assert(!MAG, "complex output only");
                const int m = ((k1 + 8 * k2) * b16) & 15;           // row a starts m cells past a line boundary
                cplx<float> *dst = reinterpret_cast<cplx<float> *>(out) + o;      // cell c0 of row a
                if constexpr (k1 % 2 == 0) {        // bins odd: m has the parity of a, i.e. of k1.  Even start: the lane's own pair
                    const int lo = 8 - (m >> 1);
                    if (lane >= lo && lane < lo + 56) {
                        if (pair_ok && second_ok) __builtin_nontemporal_store(f32x4{va.x, va.y, vb.x, vb.y}, reinterpret_cast<f32x4 *>(dst));
                        else if (pair_ok) __builtin_nontemporal_store(f32x2{va.x, va.y}, reinterpret_cast<f32x2 *>(dst));
                    }
                } else {                            // odd start: (own second cell, next lane's first cell)
                    const int nb = (lane + 1) * 4;
                    // (copies first: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0 for .y as well)
                    const float ax = va.x, ay = va.y;
                    const float nx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(nb, __builtin_bit_cast(int, ax)));
                    const float ny = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(nb, __builtin_bit_cast(int, ay)));
                    const int lo = (15 - m) >> 1;
                    if (lane >= lo && lane < lo + 56) {
                        if (second_ok && next_ok) __builtin_nontemporal_store(f32x4{vb.x, vb.y, nx, ny}, reinterpret_cast<f32x4 *>(dst + 1));
                        else if (second_ok) __builtin_nontemporal_store(f32x2{vb.x, vb.y}, reinterpret_cast<f32x2 *>(dst + 1));
                        else if (next_ok) __builtin_nontemporal_store(f32x2{nx, ny}, reinterpret_cast<f32x2 *>(dst + 2));
                    }
                }
                return;
            }
            if constexpr (ROWS == 1) {
                // row a starts m cells past a 64-B boundary (m = a * bins mod U, U = 64 B of output cells; the + 32 of the
                // fftshift drops out): this wave's aligned (128 - U)-cell window of the row begins m cells before its base
                constexpr int U = angle_rows_unit(MAG);
                const int lo = U / 2 - ((((k1 + 8 * k2) * b16) & (U - 1)) >> 1);
                if (!(pair_ok && lane >= lo && lane < lo + (128 - U) / 2)) return;
            }
            if constexpr (MAG) {
                // |.| as sqrt(x^2 + y^2): spectrum values are far from the float32 range limits, so hypotf's rescaling
                // (and its register appetite, which spilled the persistent kernel) buys nothing; error <= 1.5 ulp
                const f32x2 m = {__builtin_sqrtf(va.x * va.x + va.y * va.y), __builtin_sqrtf(vb.x * vb.x + vb.y * vb.y)};
                if constexpr (NT) __builtin_nontemporal_store(m, reinterpret_cast<f32x2 *>(out) + o);
                else reinterpret_cast<f32x2 *>(out)[o] = m;
            } else {
                const f32x4 q = {va.x, va.y, vb.x, vb.y};   // one 16-B store per lane: 1 KiB per wave instruction
                if constexpr (NT) __builtin_nontemporal_store(q, reinterpret_cast<f32x4 *>(out) + o);
                else reinterpret_cast<f32x4 *>(out)[o] = q;
            }
        });
    });
}

template <int VIN, bool MAG, bool NT, bool ZE>
__global__ __launch_bounds__(256) void k_angle64(const f32x4 *__restrict__ rd, void *__restrict__ out,
                                                  long pairs_per_frame, AngleWin win, int shift_off) {
    typedef cplx<float> C;
    // grid.x covers pairs of adjacent bins of one frame; grid.y = frame
    const long f = blockIdx.y, tile = blockIdx.x;
    const long pair = tile * 256 + threadIdx.x;
    if (pair >= pairs_per_frame) return;
    const f32x4 *src = rd + f * VIN * pairs_per_frame + pair;
    C xa[VIN], xb[VIN];
#pragma unroll
    for (int v = 0; v < VIN; ++v) {
        if (ZE && (v == 0 || v == VIN - 1)) {
            xa[v] = C{0.f, 0.f};
            xb[v] = C{0.f, 0.f};
            continue;
        }
        const f32x4 t = src[(long)v * pairs_per_frame];
        const float h = win.h[v];
        xa[v] = C{t.x * h, t.y * h};
        xb[v] = C{t.z * h, t.w * h};
    }
    angle64_passes<VIN, MAG, NT>(xa, xb, out, f, pairs_per_frame, pair, shift_off);
}

// k_angle64_rows: k_angle64 for planes whose bin count is not a multiple of 16, i.e. whose angle rows [bins] c64 do not
// start on 128-B lines (most shipped cfgs: 63 x 100, 63 x 70, 254 x 50 ...).  There a wave's 1 KB store instruction
// straddles 9 lines instead of covering 8 (12 write requests per instruction instead of 8) and the store stream drops
// from ~5.5 to ~4.3 TB/s (tools/angle_shape.py; running all tiles of a frame on one XCD, so that split lines meet in
// one L2, changed nothing -- it is the request count per instruction).  Here every WAVE computes 128 cells but stores, per row, the 112-cell
// (7-line) window of them that is line aligned for THAT row; consecutive waves overlap by 16 cells (12.5 % more loads
// and arithmetic, both far from their limits) and every store instruction writes whole lines.  Only the row ends share
// a line with the next row (one per 50 KB).
template <int VIN, bool ZE, bool MAG>
__global__ __launch_bounds__(256, (VIN <= 12 && ZE && !MAG) ? 3 : 2) void k_angle64_rows(const f32x4 *__restrict__ rd, void *__restrict__ out,
                                                       long pairs_per_frame, AngleWin win, int shift_off, int bu, int n_waves) {
    typedef cplx<float> C;
    constexpr int U = angle_rows_unit(MAG), WL = (128 - U) / 2;       // stored lanes per wave and row
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= n_waves) return;
    const long f = blockIdx.y;
    const long pair = wv * WL - U / 2 + lane;       // cells (128 - U) wv - U + 2 lane, + 1
    const bool ok = pair >= 0 && pair < pairs_per_frame;
    const f32x4 *src = rd + f * VIN * pairs_per_frame + (ok ? pair : 0);
    C xa[VIN], xb[VIN];
#pragma unroll
    for (int v = 0; v < VIN; ++v) {
        if (ZE && (v == 0 || v == VIN - 1)) {
            xa[v] = C{0.f, 0.f};
            xb[v] = C{0.f, 0.f};
            continue;
        }
        const f32x4 t = src[(long)v * pairs_per_frame];
        const float h = win.h[v];
        xa[v] = C{t.x * h, t.y * h};
        xb[v] = C{t.z * h, t.w * h};
    }
    angle64_passes<VIN, MAG, true, 1>(xa, xb, out, f, pairs_per_frame, pair, shift_off, bu, lane, ok);
}

// k_angle64_rows_odd: the same for an ODD bin count (63 x 127, 63 x 115): planes and rows then start at odd cell offsets,
// so the loads are per cell (8 B) and every second row stores pairs shifted by one cell (angle64_passes ROWS 2).  Before,
// such planes took the generic strided kernel (3.2 TB/s).
template <int VIN, bool ZE>
__global__ __launch_bounds__(256) void k_angle64_rows_odd(const cplx<float> *__restrict__ rd, void *__restrict__ out, long bins,
                                                           AngleWin win, int shift_off, int b16, int n_waves) {
    typedef cplx<float> C;
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= n_waves) return;
    const long f = blockIdx.y;
    const long c0 = wv * 112 - 16 + 2 * lane;
    const bool ok_a = c0 >= 0 && c0 < bins, ok_b = c0 + 1 >= 0 && c0 + 1 < bins, ok_n = c0 + 2 >= 0 && c0 + 2 < bins;
    const C *src = rd + f * VIN * bins;
    C xa[VIN], xb[VIN];
#pragma unroll
    for (int v = 0; v < VIN; ++v) {
        if (ZE && (v == 0 || v == VIN - 1)) {
            xa[v] = C{0.f, 0.f};
            xb[v] = C{0.f, 0.f};
            continue;
        }
        const float h = win.h[v];
        const C a = src[(long)v * bins + (ok_a ? c0 : 0)], b = src[(long)v * bins + (ok_b ? c0 + 1 : 0)];
        xa[v] = a * h;
        xb[v] = b * h;
    }
    angle64_passes<VIN, false, true, 2>(xa, xb, out, f, bins, c0, shift_off, b16, lane, ok_a, ok_b, ok_n);
}

// k_angle64_sync: the chain's device-synchronised angle stage (ChainSync above).  Persistent workgroups take
// (frame, tile of 512 bins) items from a ticket counter, wait until the frame's RD planes are published, read them
// from the ring with sc1 loads, release the slot and run the same passes + streaming stores as k_angle64.
// Register budget: 3 waves per SIMD (168 VGPRs) for the shapes that fit it -- the headline 12-antenna windowed case
// among them -- and 2 where the unrolled passes need more (a spill costs more than the lost wave).
// ROWS 1 / 2: the line-aligned row windows of k_angle64_rows / k_angle64_rows_odd (bins % 16 != 0): a tile is 4 waves x 112
// stored cells; ROWS 2 (odd bin count): `pairs_per_frame` is the bin count and the ring is read per cell.
template <int VIN, bool MAG, bool ZE, int ROWS = 0>
__global__ __launch_bounds__(256, (VIN == 12 && ZE && !MAG && !ROWS) || VIN <= 4 ? 3 : 2) void k_angle64_sync(const void *__restrict__ ring, void *__restrict__ out, long pairs_per_frame,
                                                       AngleWin win, int shift_off, ChainSync cs, int b16, int n_waves) {
    typedef cplx<float> C;
    __shared__ int sh[4];       // [0] ticket, [1] abort
    const int tid = threadIdx.x;
    // wave-uniform descriptor of the ring (sc1 buffer loads: aux = 16)
    const unsigned ring_bytes = (unsigned)((long)cs.ring * VIN * pairs_per_frame * (ROWS == 2 ? 8 : 16));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ring), 0, (int)ring_bytes, 0x00020000);
    const int n_items = cs.n_frames * cs.tiles;
    int prev_slot = -1;
    if (tid == 0) sh[1] = 0;
    for (;;) {
        if (tid == 0) sh[0] = (int)(__hip_atomic_fetch_add(cs.ctl + CTL_ANG_TICKET, 1u, MMW_RLX_AGENT) - cs.ang_base);
        __syncthreads();    // every wave is past the previous item's loads (their values were consumed)
        const int item = __builtin_amdgcn_readfirstlane(sh[0]);
        if (tid == 0 && prev_slot >= 0) __hip_atomic_fetch_add(cs.ctl + CTL_CNT + CTL_RING_MAX + prev_slot, 1u, MMW_RLX_AGENT);
        if (item >= n_items) return;
        const int f = item / cs.tiles, tile = item - f * cs.tiles;
        const unsigned g = cs.s0 + (unsigned)f;
        const int slot = (int)(g % (unsigned)cs.ring);
        if (tid == 0) {
            const unsigned target = (cs.u0 + g / (unsigned)cs.ring + 1u) * (unsigned)cs.v_live;
            unsigned *cnt = cs.ctl + CTL_CNT + slot;
            if (!chain_wait(cnt, target, __hip_atomic_load(cnt, MMW_RLX_AGENT), cs.ctl, cs.timeout, cs.naps_ang)) sh[1] = 1;
        }
        __syncthreads();    // between the poll and EVERY load of the published bytes
        if (__builtin_amdgcn_readfirstlane(sh[1])) return;
        prev_slot = slot;
        long pair = (long)tile * 256 + tid;
        bool ok = pair < pairs_per_frame, run = ok;
        [[maybe_unused]] bool ok_b = true, ok_n = true;
        if constexpr (ROWS == 1) {
            constexpr int U = angle_rows_unit(MAG);
            const long wv = (long)tile * 4 + (tid >> 6);
            pair = wv * ((128 - U) / 2) - U / 2 + (tid & 63);
            ok = pair >= 0 && pair < pairs_per_frame;
            run = wv < n_waves;
        }
        if constexpr (ROWS == 2) {
            const long wv = (long)tile * 4 + (tid >> 6);
            pair = wv * 112 - 16 + 2 * (tid & 63);      // first cell c0 of the lane
            ok = pair >= 0 && pair < pairs_per_frame;
            ok_b = pair + 1 >= 0 && pair + 1 < pairs_per_frame;
            ok_n = pair + 2 >= 0 && pair + 2 < pairs_per_frame;
            run = wv < n_waves;
        }
        if (run) {
            C xa[VIN], xb[VIN];
#pragma unroll
            for (int v = 0; v < VIN; ++v) {
                if (ZE && (v == 0 || v == VIN - 1)) {
                    xa[v] = C{0.f, 0.f};
                    xb[v] = C{0.f, 0.f};
                    continue;
                }
                if constexpr (ROWS == 2) {
                    const long plane = ((long)slot * VIN + v) * pairs_per_frame;
                    const C a = __builtin_bit_cast(C, __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)((plane + (ok ? pair : 0)) * 8), 0, 16));
                    const C b = __builtin_bit_cast(C, __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)((plane + (ok_b ? pair + 1 : 0)) * 8), 0, 16));
                    const float h = win.h[v];
                    xa[v] = a * h;
                    xb[v] = b * h;
                    continue;
                }
                const unsigned off = (unsigned)((((long)slot * VIN + v) * pairs_per_frame + (ok ? pair : 0)) * 16);
                const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
                const float h = win.h[v];
                xa[v] = C{t.x * h, t.y * h};
                xb[v] = C{t.z * h, t.w * h};
            }
            angle64_passes<VIN, MAG, true, ROWS>(xa, xb, out, f, pairs_per_frame, pair, shift_off, b16, tid & 63, ok, ok_b, ok_n);
        }
    }
}

// even bin counts whose rows are not 64-B aligned take the row-window kernels
inline bool angle_rows_needed(long bins, bool mag) {
    return bins % 2 == 0 && bins % angle_rows_unit(mag) != 0;
}
inline int angle_rows_waves(long bins, bool mag) {      // waves per frame: the last window must reach the row end for every m
    const int U = angle_rows_unit(mag);
    return (int)((bins + (U - 2) + (128 - U) - 1) / (128 - U));
}

template <int VIN>
int launch_angle64(mmw_ctx *ctx, const void *rd, void *out, int F, long bins, bool mag, const float *h, bool shift) {
    AngleWin w;
    for (int i = 0; i < 16; ++i) w.h[i] = i < VIN ? h[i] : 0.f;
    const long pairs = bins / 2;
    dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)F);
    const bool nt = true;
    const bool ze = VIN > 2 && h[0] == 0.f && h[VIN - 1] == 0.f && opt_int(ctx, "MMW_ANGLE_ZE", 1) != 0;
    if (bins % 2) {         // angle_fast_path admits odd bin counts only for complex output
        const int n_waves = (int)((bins + 15 + 111) / 112);
        dim3 g((unsigned)((n_waves + 3) / 4), (unsigned)F);
        if (ze) hipLaunchKernelGGL((k_angle64_rows_odd<VIN, true>), g, dim3(256), 0, ctx->stream, (const cplx<float> *)rd, out, bins, w, shift ? 32 : 0, (int)(bins & 15), n_waves);
        else hipLaunchKernelGGL((k_angle64_rows_odd<VIN, false>), g, dim3(256), 0, ctx->stream, (const cplx<float> *)rd, out, bins, w, shift ? 32 : 0, (int)(bins & 15), n_waves);
        return check_launch("angle64_rows_odd");
    }
    if (angle_rows_needed(bins, mag)) {
        const int n_waves = angle_rows_waves(bins, mag), bu = (int)(bins & (angle_rows_unit(mag) - 1));
        dim3 g((unsigned)((n_waves + 3) / 4), (unsigned)F);
#define MMW_ANGLE_ROWS_LAUNCH(ZEV, MAGV) \
    hipLaunchKernelGGL((k_angle64_rows<VIN, ZEV, MAGV>), g, dim3(256), 0, ctx->stream, (const f32x4 *)rd, out, pairs, w, shift ? 32 : 0, bu, n_waves)
        if (ze && mag) MMW_ANGLE_ROWS_LAUNCH(true, true);
        else if (ze) MMW_ANGLE_ROWS_LAUNCH(true, false);
        else if (mag) MMW_ANGLE_ROWS_LAUNCH(false, true);
        else MMW_ANGLE_ROWS_LAUNCH(false, false);
#undef MMW_ANGLE_ROWS_LAUNCH
        return check_launch("angle64_rows");
    }
#define MMW_ANGLE_LAUNCH(MAGV, NTV, ZEV) \
    hipLaunchKernelGGL((k_angle64<VIN, MAGV, NTV, ZEV>), grid, dim3(256), 0, ctx->stream, (const f32x4 *)rd, out, pairs, w, \
                       shift ? 32 : 0)
    if (ze) {
        if (mag && nt) MMW_ANGLE_LAUNCH(true, true, true);
        else if (mag) MMW_ANGLE_LAUNCH(true, false, true);
        else if (nt) MMW_ANGLE_LAUNCH(false, true, true);
        else MMW_ANGLE_LAUNCH(false, false, true);
    } else {
        if (mag && nt) MMW_ANGLE_LAUNCH(true, true, false);
        else if (mag) MMW_ANGLE_LAUNCH(true, false, false);
        else if (nt) MMW_ANGLE_LAUNCH(false, true, false);
        else MMW_ANGLE_LAUNCH(false, false, false);
    }
#undef MMW_ANGLE_LAUNCH
    return check_launch("angle64");
}

// angle work items per frame of the device-synchronised chain
inline bool angle_sync_rows(long bins, bool mag) { return angle_rows_needed(bins, mag); }
inline int angle_sync_tiles(long bins, bool mag) {
    if (bins % 2) return (int)(((bins + 15 + 111) / 112 + 3) / 4);
    if (angle_sync_rows(bins, mag)) return (angle_rows_waves(bins, mag) + 3) / 4;
    return (int)((bins / 2 + 255) / 256);
}

template <int VIN>
int launch_angle64_sync(mmw_ctx *ctx, const void *ring, void *out, long bins, bool mag, const float *h, bool shift,
                        ChainSync cs, int grid) {
    AngleWin w;
    for (int i = 0; i < 16; ++i) w.h[i] = i < VIN ? h[i] : 0.f;
    const long pairs = bins / 2;
    const bool ze = VIN > 2 && h[0] == 0.f && h[VIN - 1] == 0.f;
    if (bins % 2) {                         // odd bin count (complex output only): cs.tiles from angle_sync_tiles()
        const int n_waves = (int)((bins + 15 + 111) / 112);
        if (ze) hipLaunchKernelGGL((k_angle64_sync<VIN, false, true, 2>), dim3(grid), dim3(256), 0, ctx->stream, ring, out, bins, w, shift ? 32 : 0, cs, (int)(bins & 15), n_waves);
        else hipLaunchKernelGGL((k_angle64_sync<VIN, false, false, 2>), dim3(grid), dim3(256), 0, ctx->stream, ring, out, bins, w, shift ? 32 : 0, cs, (int)(bins & 15), n_waves);
        return check_launch("angle64_sync_rows_odd");
    }
    if (angle_sync_rows(bins, mag)) {       // cs.tiles was sized for this variant by angle_sync_tiles()
        const int n_waves = angle_rows_waves(bins, mag), bu = (int)(bins & (angle_rows_unit(mag) - 1));
#define MMW_ANGLE_SYNC_ROWS(MAGV, ZEV) \
    hipLaunchKernelGGL((k_angle64_sync<VIN, MAGV, ZEV, 1>), dim3(grid), dim3(256), 0, ctx->stream, ring, out, pairs, w, shift ? 32 : 0, cs, bu, n_waves)
        if (ze && mag) MMW_ANGLE_SYNC_ROWS(true, true);
        else if (ze) MMW_ANGLE_SYNC_ROWS(false, true);
        else if (mag) MMW_ANGLE_SYNC_ROWS(true, false);
        else MMW_ANGLE_SYNC_ROWS(false, false);
#undef MMW_ANGLE_SYNC_ROWS
        return check_launch("angle64_sync_rows");
    }
#define MMW_ANGLE_SYNC(MAGV, ZEV) \
    hipLaunchKernelGGL((k_angle64_sync<VIN, MAGV, ZEV>), dim3(grid), dim3(256), 0, ctx->stream, ring, out, pairs, w, \
                       shift ? 32 : 0, cs, 0, 0)
    if (ze) {
        if (mag) MMW_ANGLE_SYNC(true, true);
        else MMW_ANGLE_SYNC(false, true);
    } else {
        if (mag) MMW_ANGLE_SYNC(true, false);
        else MMW_ANGLE_SYNC(false, false);
    }
#undef MMW_ANGLE_SYNC
    return check_launch("angle64_sync");
}

// k_angle64_rmean: angle FFT + |.| + mean over a range window in one pass -- the tail of
// DopplerAzimuthProcessor.process (processors/doppler_azimuth_resp.py:320-332,486-489), which never needs the
// [A][S][C] magnitude cube itself.  Same pruned 8 x 8 DFT as k_angle64; thread = (column, row lane): for each
// of its rows of the range window it loads the V inputs once, runs the eight k1 passes and adds the 64 x 2 magnitudes
// to running sums held in registers.  Partial sums per (range partition, row lane) go to
// part[f][p][rl][a][c]; k_rmean_finish adds them in a fixed order (deterministic) and scales by 1 / rows, writing
// out[f][c][a].  HBM traffic: the RD rows once plus 1/(rows per partition) of the magnitude cube.
constexpr int RMEAN_RL = 4;     // row lanes per workgroup (256 threads = 64 columns x 4)
template <int VIN, bool ZE>
__global__ __launch_bounds__(256) void k_angle64_rmean(const cplx<float> *__restrict__ rd, float *__restrict__ part, int S,
                                                        int C, int s_lo, int s_hi, int rows_per_part, AngleWin win,
                                                        int shift_off) {
    typedef cplx<float> Cx;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    if (c >= C) return;
    const int p = blockIdx.y, P = gridDim.y;
    const long f = blockIdx.z;
    const long plane = (long)S * C;
    const Cx *src = rd + f * VIN * plane + c;
    const int r0 = s_lo + p * rows_per_part, r1 = min(s_hi, r0 + rows_per_part);
    float *dst = part + (((f * P + p) * RMEAN_RL + rl) * 64) * C + c;
    // the 64 running sums stay in registers (static indices), so every input row is read exactly once
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    for (int s = r0 + rl; s < r1; s += RMEAN_RL) {
        Cx x[VIN];
#pragma unroll
        for (int v = 0; v < VIN; ++v) {
            if (ZE && (v == 0 || v == VIN - 1)) {
                x[v] = Cx{0.f, 0.f};
                continue;
            }
            x[v] = __builtin_nontemporal_load(src + (long)v * plane + (long)s * C) * win.h[v];
        }
        static_for<8>([&](auto K1) {
            constexpr int k1 = decltype(K1)::value;
            Cx z[8];
            static_for<8>([&](auto N2) {
                constexpr int n2 = decltype(N2)::value;
                Cx y = Cx{0.f, 0.f};
                if constexpr (n2 < VIN) y = x[n2];
                if constexpr (n2 + 8 < VIN) y = y + mul_w<8, k1, float, Cx>(x[n2 + 8]);
                z[n2] = mul_w<64, n2 * k1, float, Cx>(y);
            });
            RegFFT<8, float, 8, 0, Cx>::run(z);
            static_for<8>([&](auto K2) {
                constexpr int k2 = decltype(K2)::value;
                const Cx v = z[bitrev<8>(k2)];
                // v_sqrt_f32 (1 ulp) instead of the refined sqrtf sequence
                acc[k1 + 8 * k2] += __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y);
            });
        });
    }
#pragma unroll
    for (int k = 0; k < 64; ++k) dst[(long)((k + shift_off) & 63) * C] = acc[k];
}

// out[f][c][a] = (sum over partitions p and row lanes rl, in that order, of part[f][p][rl][a][c]) / rows
static __global__ __launch_bounds__(256) void k_rmean_finish(const float *__restrict__ part, float *__restrict__ out, long F,
                                                       int P, int C, float inv_rows) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= F * 64 * C) return;
    const int c = (int)(gid % C), a = (int)((gid / C) % 64);
    const long f = gid / ((long)C * 64);
    float acc = 0.f;
    for (int q = 0; q < P * RMEAN_RL; ++q) acc += part[((f * P * RMEAN_RL + q) * 64 + a) * C + c];
    out[(f * C + c) * 64 + a] = acc * inv_rows;
}

// rd [F][VIN][S][C] c64 -> out [F][C][64] float32; part: scratch of F * P * RMEAN_RL * 64 * C floats
inline int rmean_partitions(int F, int S, int C, int rows) {
    const int tiles = (C + 63) / 64;
    int P = (2048 + F * tiles - 1) / (F * tiles);              // ~2048 workgroups in flight
    const int max_p = (rows + RMEAN_RL - 1) / RMEAN_RL;        // at least one row per thread
    if (P > max_p) P = max_p;
    return P < 1 ? 1 : P;
}

template <int VIN>
int launch_angle64_rmean(mmw_ctx *ctx, const void *rd, float *part, size_t part_bytes, float *out, int F, int S, int C,
                         int s_lo, int s_hi, const float *h, bool shift) {
    AngleWin w;
    for (int i = 0; i < 16; ++i) w.h[i] = i < VIN ? h[i] : 0.f;
    const int rows = s_hi - s_lo;
    int P = rmean_partitions(F, S, C, rows);
    const size_t per_p = (size_t)F * RMEAN_RL * 64 * C * sizeof(float);
    if ((size_t)P * per_p > part_bytes) P = (int)(part_bytes / per_p);     // a short last chunk may not use more scratch
    if (P < 1) return set_error(MMW_ERR_INVALID, "range-mean scratch too small");
    const int rpp = (rows + P - 1) / P;
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)P, (unsigned)F);
    const bool ze = VIN > 2 && h[0] == 0.f && h[VIN - 1] == 0.f;
    if (ze)
        hipLaunchKernelGGL((k_angle64_rmean<VIN, true>), grid, dim3(256), 0, ctx->stream, (const cplx<float> *)rd, part, S, C, s_lo,
                           s_hi, rpp, w, shift ? 32 : 0);
    else
        hipLaunchKernelGGL((k_angle64_rmean<VIN, false>), grid, dim3(256), 0, ctx->stream, (const cplx<float> *)rd, part, S, C,
                           s_lo, s_hi, rpp, w, shift ? 32 : 0);
    MMW_TRY(check_launch("angle64_rmean"));
    const long total = (long)F * 64 * C;
    hipLaunchKernelGGL(k_rmean_finish, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, part, out, (long)F, P,
                       C, 1.0f / (float)rows);
    return check_launch("rmean_finish");
}

// Raw DCA-style cube [F][num_rx][S][num_tx * C] read in place of the virtual-array cube [F][V][S][C]: virtual
// antenna v = tx * num_rx + rx takes every num_tx-th chirp starting at tx
// (VirtualArrayReformatter.process, processors/virtual_array_reformater.py:53-63).  Element (s, c) of plane
// p = frame * V + v sits at raw_plane(...)[(s * C + c) * ntx]; the three tx planes of one rx share cache lines.
struct RawView {
    int ntx, nrx;       // ntx <= 1: the input already is the virtual-array cube
    int vskip;          // V > 2: do not transform the planes of virtual antennas 0 and V - 1 of every frame (their
                        // Hann(V) weight is exactly 0 and the chain's angle kernel never loads them); 0 = all planes
    int i16;            // raw cube of int16 (I, Q) pairs instead of complex64 (same indexing, 4-byte cells); NO UPSTREAM
                        // ORACLE for this sample layout (mmw_virtual_array_reformat_i16)
};
// element `elem` of a raw cube: complex64, or an int16 (I, Q) pair converted in registers
template <bool I16> __device__ __forceinline__ cplx<float> raw_cell(const void *base, long elem) {
    if constexpr (I16) {
        const short2 v = reinterpret_cast<const short2 *>(base)[elem];
        return cplx<float>{(float)v.x, (float)v.y};
    } else
        return reinterpret_cast<const cplx<float> *>(base)[elem];
}
// Planes a launch really transforms, and workgroup -> plane for the non-raw kernels (raw kernels keep their XCD-grouped
// mapping and the workgroups of a skipped plane exit at once).
__host__ __device__ __forceinline__ long skip_planes(long planes, RawView rv) {
    return rv.vskip > 2 ? planes / rv.vskip * (rv.vskip - 2) : planes;
}
__device__ __forceinline__ long skip_block_plane(long b, RawView rv) {
    if (rv.vskip <= 2) return b;
    const long f = b / (rv.vskip - 2);
    return f * rv.vskip + 1 + (b - f * (rv.vskip - 2));
}
__device__ __forceinline__ bool skip_raw_plane(long plane, RawView rv) {
    if (rv.vskip <= 2) return false;
    const int v = (int)(plane % rv.vskip);
    return v == 0 || v == rv.vskip - 1;
}
// Workgroup -> plane for raw cubes.  The num_tx planes of one (frame, rx) pair read the same raw rows, so they are
// given to workgroups b, b + 8, b + 16 ...: hardware dispatch is round-robin over the 8 XCDs, which puts them on the
// SAME XCD at about the same time and the shared lines come out of that XCD's L2 instead of being fetched once per
// XCD.  Grid = raw_grid(planes, rv) workgroups; a workgroup past the last pair gets -1.
__host__ __device__ __forceinline__ long raw_grid(long planes, RawView rv) {
    const long pairs = planes / rv.ntx;              // (frame, rx) pairs
    return (pairs + 7) / 8 * 8 * rv.ntx;
}
__device__ __forceinline__ long raw_block_plane(long b, long planes, RawView rv) {
    const long slot = b >> 3;
    const int xcd = (int)(b & 7), tx = (int)(slot % rv.ntx);
    const long g = (slot / rv.ntx) * 8 + xcd;
    if (g >= planes / rv.ntx) return -1;
    const long f = g / rv.nrx;
    const int rx = (int)(g - f * rv.nrx);
    return f * (rv.nrx * rv.ntx) + tx * rv.nrx + rx;
}
__device__ __forceinline__ long raw_plane_off(long plane, int S, int C, RawView rv) {      // in cells
    const int V = rv.nrx * rv.ntx;
    const long f = plane / V;
    const int v = (int)(plane - f * V);
    const int tx = v / rv.nrx, rx = v - tx * rv.nrx;
    return ((f * rv.nrx + rx) * S) * (long)C * rv.ntx + tx;
}
__device__ __forceinline__ const cplx<float> *raw_plane(const cplx<float> *in, long plane, int S, int C, RawView rv) {
    return in + raw_plane_off(plane, S, C, rv);
}

// ------------------------------------------------------------------ fused range-Doppler, 256 x 128
// k_rd_fused_256x128: one workgroup (1024 threads = 16 waves) transforms one [256 samples][128 chirps]
// plane in a single pass over HBM (reads 256 KiB, writes 256 KiB; algorithmic 6.29 MB/frame for 12 planes).
// The plane does not fit LDS (256 KiB > 160 KiB): it lives in REGISTERS (32 complex per thread) and moves
// through LDS three times, half a plane (128 range rows) at a time:
//   step 0   lane <-> chirp pair, wave <-> sample phase j: 16 coalesced 1-KiB row loads, Hann(S) x Hann(C),
//            16-point range FFT over n1 (samples 16*n1 + j) in registers, twiddle W256^(j*k1) (wave-uniform).
//   X1       exchange over j  -> second 16-point range FFT: rows kr = k1 + 16*k2 of the half are complete.
//   X2       transpose to row-major (pitch 152 keeps the strided reads bank-conflict free).
//   Doppler  16-point FFT over chirps 8*n1 + j8, twiddle W128^(j8*k1d), X3 exchange in place inside the
//            wave's own 8 rows, 8-point FFT, fftshift folded into the store index (128-B store segments).
// Range FFT = processors/range_doppler_resp.py:99-101 axis -2, Doppler = axis -1, shift = :98,103.
constexpr int RD_S = 256, RD_C = 128, RD_PITCH = 152;
constexpr int RD_LDS_MAIN = 128 * RD_PITCH;                       // complex elements (>= 8*16*128 for X1)
constexpr int RD_LDS_BYTES = RD_LDS_MAIN * 8 + 128 * 8 + 16 + 64;      // + W128 table + ticket words + L1 partials

// RAW: the input is the raw [F][num_rx][256][num_tx * 128] cube (two 8-B loads per lane and row instead of one 16-B)
template <bool NTIN, int ABL = 0, bool RAW = false, bool I16 = false>   // ABL: timing-only ablations (1 no stores, 2 no loads, 3 neither)
__global__ __launch_bounds__(1024) void k_rd_fused_256x128(const f32x4 *__restrict__ in, cplx<float> *__restrict__ out,
                                                            int planes, const float *__restrict__ hann_s,
                                                            const float *__restrict__ hann_c,
                                                            const cplx<float> *__restrict__ tw256,
                                                            const cplx<float> *__restrict__ tw128, RawView rv, int nt_out) {
    // nt_out: non-temporal stores (a stand-alone launch: the cube is written once and is larger than the caches; inside the
    // event-schedule chain the output is a ring that is meant to stay cached: plain stores)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(smem);
    cplx<float> *tw128_l = lds + RD_LDS_MAIN;
    const int t = threadIdx.x;
    const int l = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    if (t < 128) tw128_l[t] = tw128[t];
    const float hc0 = hann_c[2 * l], hc1 = hann_c[2 * l + 1];

    {
        int plane = blockIdx.x;
        if constexpr (RAW) {
            plane = (int)raw_block_plane(blockIdx.x, planes, rv);
            if (plane < 0 || skip_raw_plane(plane, rv)) return;
        } else plane = (int)skip_block_plane(blockIdx.x, rv);
        const f32x4 *src = in + (long)plane * (RD_S * RD_C / 2);
        long roff = 0;
        if constexpr (RAW) roff = raw_plane_off(plane, RD_S, RD_C, rv);
        cplx<float> *dst = out + (long)plane * (RD_S * RD_C);
        // ---- step 0: load, window, range pass 1 (n = 16*n1 + w)
        cplx<float> y0[16], y1[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int n = 16 * n1 + w;
            f32x4 v;
            if constexpr (ABL & 2) v = f32x4{(float)(n + l), (float)(n - l), (float)(n ^ l), 1.0f};
            else if constexpr (RAW) {
                const long e = roff + (long)(n * RD_C + 2 * l) * rv.ntx;
                const cplx<float> a = raw_cell<I16>(in, e), b = raw_cell<I16>(in, e + rv.ntx);
                v = f32x4{a.x, a.y, b.x, b.y};
            } else v = NTIN ? __builtin_nontemporal_load(src + n * (RD_C / 2) + l) : src[n * (RD_C / 2) + l];
            const float hs = hann_s[n];
            y0[n1] = cplx<float>{v.x, v.y} * (hs * hc0);
            y1[n1] = cplx<float>{v.z, v.w} * (hs * hc1);
        }
        RegFFT<16, float>::run(y0);
        RegFFT<16, float>::run(y1);
        static_for<2>([&](auto H) {
            constexpr int h = decltype(H)::value;
            // ---- X1: [k1l][j = w][c], twiddle applied on the way out
            static_for<8>([&](auto K) {
                constexpr int k1l = decltype(K)::value;
                constexpr int k1 = 8 * h + k1l;
                const cplx<float> tw = tw256[w * k1];
                constexpr int br = bitrev<16>(k1);
                const cplx<float> a = cmul(y0[br], tw), b = cmul(y1[br], tw);
                *reinterpret_cast<f32x4 *>(&lds[(k1l * 16 + w) * 128 + 2 * l]) = f32x4{a.x, a.y, b.x, b.y};
            });
            __syncthreads();
            const int k1l = w & 7, c = 64 * (w >> 3) + l;
            cplx<float> b[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = lds[(k1l * 16 + j) * 128 + c];
            RegFFT<16, float>::run(b);
            __syncthreads();
            // ---- X2: row-major [rl = k1l + 8*k2][c], pitch 152
            static_for<16>([&](auto K) {
                constexpr int k2 = decltype(K)::value;
                lds[(k1l + 8 * k2) * RD_PITCH + c] = b[bitrev<16>(k2)];
            });
            __syncthreads();
            // ---- Doppler pass 1: thread (rl, j8), chirps 8*n1 + j8
            const int rl = t >> 3, j8 = t & 7;
            cplx<float> d[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) d[n1] = lds[rl * RD_PITCH + 8 * n1 + j8];
            RegFFT<16, float>::run(d);
            // X3 in place: only this wave reads/writes rows 8w .. 8w+7 from here on
            static_for<16>([&](auto K) {
                constexpr int k1d = decltype(K)::value;
                lds[rl * RD_PITCH + k1d * 9 + j8] = cmul(d[bitrev<16>(k1d)], tw128_l[j8 * k1d]);
            });
            __syncthreads();
            // ---- Doppler pass 2 + store (two (row, k1d) units per lane)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = l + 64 * u;
                const int k1d = idx & 15, row = 8 * w + (idx >> 4);
                cplx<float> e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) e[q] = lds[row * RD_PITCH + k1d * 9 + q];
                RegFFT<8, float>::run(e);
                const int kr = 8 * h + (row & 7) + 16 * (row >> 3);
                static_for<8>([&](auto K) {
                    constexpr int k2d = decltype(K)::value;
                    const int kk = (k1d + 16 * k2d) ^ 64;       // fftshift over the 128 Doppler bins
                    const cplx<float> val = e[bitrev<8>(k2d)];
                    if constexpr (ABL & 1) {
                        const float vx = val.x, vy = val.y;
                        asm volatile("" ::"v"(vx), "v"(vy));
                    } else if (nt_out) {
                        __builtin_nontemporal_store(val, &dst[kr * RD_C + kk]);
                    } else {
                        dst[kr * RD_C + kk] = val;
                    }
                });
            }
            __syncthreads();
        });
    }
}

// Persistent variant: each workgroup walks planes blockIdx.x, +gridDim.x, ... and issues the NEXT plane's 16 row
// loads as soon as the y registers die (after the second X1 write), so the load latency hides under the rest of
// the current plane.  PF = rows prefetched that early; the rest load at the loop top.  PF = 8 fits the register
// budget (126 VGPRs, no scratch) and is 12 % faster than one plane per workgroup; PF = 16 spilled and was slower.
// SYNC: the chain's device-synchronised form (ChainSync above): planes come from a ticket counter, `out` is the ring of
// RD frames, stores are sc1, and the workgroup waits for / signals the per-slot counters.
// L1N: also write l1[plane] = sum of |re| + |im| of the windowed plane (the error-bound scale of mmw_angle_argmax_exact;
// same quantity as k_plane_l1, here for free while the samples are in registers).
// RAWIN (SYNC only): the input is the raw cube; the de-interleave is folded into the row loads (two 8-B loads per lane).
// I16 (RAWIN only): the raw cube holds int16 (I, Q) cells: one 4-byte load per cell, converted when the plane is windowed.
// DET (SYNC + L1N): the producer of the device-synchronised DETECTION pipeline (mmw_detect.h: DetSync).  `out` is the caller's
// plain [F][V][256][128] cube, not a ring -- nothing is ever overwritten, so the producer never waits --, item = frame * V +
// antenna; every store (the plane AND its L1 norm) is sc1, and a published plane adds one to cs.frame_cnt[frame].
template <bool NTIN, int PF, bool SYNC = false, bool L1N = false, bool RAWIN = false, bool I16 = false, bool DET = false>
__global__ __launch_bounds__(1024) void k_rd_fused_256x128_persist(const f32x4 *__restrict__ in, cplx<float> *__restrict__ out,
                                                            int planes, const float *__restrict__ hann_s,
                                                            const float *__restrict__ hann_c,
                                                            const cplx<float> *__restrict__ tw256,
                                                            const cplx<float> *__restrict__ tw128, ChainSync cs,
                                                            float *__restrict__ l1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(smem);
    cplx<float> *tw128_l = lds + RD_LDS_MAIN;
    int *lds_ctl = reinterpret_cast<int *>(tw128_l + 128);      // SYNC: [0] / [1] tickets (double buffered), [2] abort
    float *lds_l1 = reinterpret_cast<float *>(lds_ctl + 4);     // L1N: one partial sum per wave
    if (threadIdx.x < 128) tw128_l[threadIdx.x] = tw128[threadIdx.x];
    const int t0 = threadIdx.x;
    const int l0 = t0 & 63;
    const int w0 = __builtin_amdgcn_readfirstlane(t0 >> 6);

    f32x4 nx[16];
    // work item -> input plane (SYNC: item = frame * v_live + live antenna)
    // live-plane index within a frame -> virtual antenna
    auto live_antenna = [&](int vi) {
        if constexpr (RAWIN) return (int)((cs.vmap >> (4 * vi)) & 15);
        else return cs.vskip > 2 ? vi + 1 : vi;
    };
    auto issue_loads = [&](int item, auto FIRST, auto LAST) {
        if constexpr (RAWIN) {
            // item -> (frame, virtual antenna v = tx * nrx + rx): every ntx-th chirp of raw row (frame, rx), starting at tx
            const int f = item / cs.v_live, v = live_antenna(item - f * cs.v_live);
            const int tx = v / cs.nrx, rx = v - tx * cs.nrx;
            // buffer loads: the (uniform) row base lives in the descriptor and the scalar offset, the lane part is one
            // 32-bit register for the whole plane -- 64-bit per-lane pointers for 32 loads do not fit the register budget
            constexpr unsigned CELL = I16 ? 4u : 8u;
            const char *rowbase = reinterpret_cast<const char *>(in) + ((((long)f * cs.nrx + rx) * RD_S) * (long)RD_C * cs.ntx + tx) * (long)CELL;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(rowbase), 0,
                                                              (int)((unsigned)(RD_S * RD_C) * (unsigned)cs.ntx * CELL), 0x00020000);
            const unsigned voff = (unsigned)(2 * l0 * cs.ntx) * CELL, step = (unsigned)cs.ntx * CELL;
#pragma unroll
            for (int n1 = decltype(FIRST)::value; n1 < decltype(LAST)::value; ++n1) {
                const unsigned soff = (unsigned)((16 * n1 + w0) * RD_C * cs.ntx) * CELL;
                if constexpr (I16) {        // the raw (I, Q) words travel as they are: converting here would wait for the load
                    const unsigned a = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
                    const unsigned b = __builtin_amdgcn_raw_buffer_load_b32(rs, voff + step, soff, 0);
                    nx[n1] = f32x4{__builtin_bit_cast(float, a), 0.f, __builtin_bit_cast(float, b), 0.f};
                } else {
                    const cplx<float> a = __builtin_bit_cast(cplx<float>, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
                    const cplx<float> b = __builtin_bit_cast(cplx<float>, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + step, soff, 0));
                    nx[n1] = f32x4{a.x, a.y, b.x, b.y};
                }
            }
        } else {
            long plane_in = item;
            if constexpr (SYNC && !DET) {
                const int f = item / cs.v_live;
                plane_in = (long)f * cs.V + live_antenna(item - f * cs.v_live);
            }
            const f32x4 *src = in + plane_in * (RD_S * RD_C / 2);
#pragma unroll
            for (int n1 = decltype(FIRST)::value; n1 < decltype(LAST)::value; ++n1) {
                const int n = 16 * n1 + w0;
                nx[n1] = NTIN ? __builtin_nontemporal_load(src + n * (RD_C / 2) + l0) : src[n * (RD_C / 2) + l0];
            }
        }
    };
    int first = blockIdx.x, iter = 0;
    [[maybe_unused]] int unpublished = -1;      // DET: the plane stored last, not yet published (see the end of the loop body)
    if constexpr (SYNC) {
        if (t0 == 0) {
            lds_ctl[0] = (int)(__hip_atomic_fetch_add(cs.ctl + CTL_RD_TICKET, 1u, MMW_RLX_AGENT) - cs.rd_base);
            lds_ctl[2] = 0;
        }
        __syncthreads();
        first = __builtin_amdgcn_readfirstlane(lds_ctl[0]);     // uniform: the slot arithmetic stays on the scalar unit
    }
    if (first < planes) issue_loads(first, std::integral_constant<int, 0>{}, std::integral_constant<int, PF>{});
    for (int plane = first; plane < planes; ++iter) {
        cplx<float> *dst = out + (long)plane * (RD_S * RD_C);
        // SYNC: slot bookkeeping and (thread 0) the early, un-waited poll of the slot's consumer counter + next ticket
        int slot = 0;
        unsigned free_target = 0, free_seen = 0, next_ticket = 0;
        if constexpr (DET) {
            // (asked for here, un-waited; it has arrived with the rows that step 0 waits for anyway, and goes to the LDS there)
            if (t0 == 0) next_ticket = __hip_atomic_fetch_add(cs.ctl + CTL_RD_TICKET, 1u, MMW_RLX_AGENT) - cs.rd_base;
        } else if constexpr (SYNC) {
            const int f = plane / cs.v_live, vi = plane - f * cs.v_live;
            const unsigned g = cs.s0 + (unsigned)f;
            slot = (int)(g % (unsigned)cs.ring);
            free_target = (cs.u0 + g / (unsigned)cs.ring) * (unsigned)cs.tiles;
            dst = out + ((long)slot * cs.V + live_antenna(vi)) * (RD_S * RD_C);
            if (t0 == 0) {
                free_seen = __hip_atomic_load(cs.ctl + CTL_CNT + CTL_RING_MAX + slot, MMW_RLX_AGENT);
                next_ticket = __hip_atomic_fetch_add(cs.ctl + CTL_RD_TICKET, 1u, MMW_RLX_AGENT) - cs.rd_base;
            }
        }
        // Re-derive the thread indices behind an opaque asm every plane: otherwise hipcc hoists every
        // lane-constant LDS / global address out of the plane loop and spills them around it.
        int t = t0;
        asm volatile("" : "+v"(t));
        const int l = t & 63;
        const int w = __builtin_amdgcn_readfirstlane(t >> 6);
        const float hc0 = hann_c[2 * l], hc1 = hann_c[2 * l + 1];
        if constexpr (PF < 16) issue_loads(plane, std::integral_constant<int, PF>{}, std::integral_constant<int, 16>{});
        // ---- step 0: window, range pass 1 (n = 16*n1 + w)
        cplx<float> y0[16], y1[16];
        [[maybe_unused]] float l1_acc = 0.f;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int n = 16 * n1 + w;
            f32x4 v = nx[n1];
            if constexpr (I16) {
                // (copies first: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0 for .z as well)
                const float wa = v.x, wb = v.z;
                const int a = __builtin_bit_cast(int, wa), b = __builtin_bit_cast(int, wb);
                v = f32x4{(float)(short)(a & 0xffff), (float)(a >> 16), (float)(short)(b & 0xffff), (float)(b >> 16)};
            }
            const float hs = hann_s[n];
            y0[n1] = cplx<float>{v.x, v.y} * (hs * hc0);
            y1[n1] = cplx<float>{v.z, v.w} * (hs * hc1);
            if constexpr (L1N) l1_acc += (fabsf(y0[n1].x) + fabsf(y0[n1].y)) + (fabsf(y1[n1].x) + fabsf(y1[n1].y));
        }
        if constexpr (DET) {
            if (t == 0) lds_ctl[(iter + 1) & 1] = (int)next_ticket;     // read behind this plane's barriers
        }
        if constexpr (L1N) {
            for (int d = 32; d >= 1; d >>= 1) l1_acc += __shfl_xor(l1_acc, d, 64);
            if (l == 0) lds_l1[w] = l1_acc;      // read by thread 0 after this plane's first barrier
        }
        RegFFT<16, float>::run(y0);
        RegFFT<16, float>::run(y1);
        bool dead = false;      // SYNC: hand-off timed out / aborted: no stores, leave after this plane's barriers
        unsigned ring_soff = 0;
        auto ring_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);      // SYNC: set once the slot is free
        if constexpr (SYNC && !DET) ring_soff = (unsigned)(dst - out) * 8u;
        // DET: one descriptor per plane (its base is wave-uniform; a cube of many frames exceeds 32-bit offsets)
        if constexpr (DET) ring_rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, RD_S * RD_C * 8, 0x00020000);
        static_for<2>([&](auto H) {
            constexpr int h = decltype(H)::value;
            // ---- X1: [k1l][j = w][c], twiddle applied on the way out
            static_for<8>([&](auto K) {
                constexpr int k1l = decltype(K)::value;
                constexpr int k1 = 8 * h + k1l;
                const cplx<float> tw = tw256[w * k1];
                constexpr int br = bitrev<16>(k1);
                const cplx<float> a = cmul(y0[br], tw), b = cmul(y1[br], tw);
                *reinterpret_cast<f32x4 *>(&lds[(k1l * 16 + w) * 128 + 2 * l]) = f32x4{a.x, a.y, b.x, b.y};
            });
            if constexpr (h == 1 && !SYNC) {     // y0 / y1 are dead from here on: their registers take the next plane
                const int next = plane + gridDim.x;
                if (next < planes) issue_loads(next, std::integral_constant<int, 0>{}, std::integral_constant<int, PF>{});
            }
            if constexpr (h == 0 && SYNC) {
                // the slot must be free before this plane's first store (end of this half); the next ticket goes to
                // the other waves through LDS at the same barrier
                if constexpr (!DET) {
                    if (t == 0) {
                        if (!chain_wait(cs.ctl + CTL_CNT + CTL_RING_MAX + slot, free_target, free_seen, cs.ctl, cs.timeout, cs.naps_rd)) lds_ctl[2] = 1;
                        lds_ctl[(iter + 1) & 1] = (int)next_ticket;
                    }
                }
            }
            __syncthreads();
            if constexpr (DET && h == 0) {
                // Deferred publication of the PREVIOUS plane: vmcnt retires in issue order, loads and stores alike, so a wave
                // that has the rows of THIS plane it asked for after its last stores (rows PF..15, waited for in step 0) has
                // those stores behind it; every wave is past step 0 here.  The producer never stalls on a store drain.
                if (t == 0 && unpublished >= 0) __hip_atomic_fetch_add(cs.frame_cnt + unpublished / cs.V, 1u, MMW_RLX_AGENT);
            }
            if constexpr (L1N && h == 0) {
                if (t == 0) {               // fixed summation order: the value does not depend on timing
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) a += lds_l1[i];
                    // DET: the consumer reads it from another XCD: write-through, drained with this wave's plane stores
                    if constexpr (DET) __hip_atomic_store(l1 + plane, a, MMW_RLX_AGENT);
                    else l1[plane] = a;
                }
            }
            if constexpr (SYNC) {
                if constexpr (h == 0) {
                    if constexpr (!DET) {
                        dead = __builtin_amdgcn_readfirstlane(lds_ctl[2]) != 0;
                        ring_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, dead ? 0 : (int)((unsigned)cs.ring * (unsigned)cs.V * (RD_S * RD_C * 8u)), 0x00020000);
                    }
                } else {
                    const int next = __builtin_amdgcn_readfirstlane(lds_ctl[(iter + 1) & 1]);
                    if (next < planes) issue_loads(next, std::integral_constant<int, 0>{}, std::integral_constant<int, PF>{});
                }
            }
            const int k1l = w & 7, c = 64 * (w >> 3) + l;
            cplx<float> b[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = lds[(k1l * 16 + j) * 128 + c];
            RegFFT<16, float>::run(b);
            __syncthreads();
            // ---- X2: row-major [rl = k1l + 8*k2][c], pitch 152
            static_for<16>([&](auto K) {
                constexpr int k2 = decltype(K)::value;
                lds[(k1l + 8 * k2) * RD_PITCH + c] = b[bitrev<16>(k2)];
            });
            __syncthreads();
            // ---- Doppler pass 1: thread (rl, j8), chirps 8*n1 + j8
            const int rl = t >> 3, j8 = t & 7;
            cplx<float> d[16];
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) d[n1] = lds[rl * RD_PITCH + 8 * n1 + j8];
            RegFFT<16, float>::run(d);
            // X3 in place: only this wave reads/writes rows 8w .. 8w+7 from here on
            static_for<16>([&](auto K) {
                constexpr int k1d = decltype(K)::value;
                lds[rl * RD_PITCH + k1d * 9 + j8] = cmul(d[bitrev<16>(k1d)], tw128_l[j8 * k1d]);
            });
            __syncthreads();
            // ---- Doppler pass 2 + store (two (row, k1d) units per lane)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = l + 64 * u;
                const int k1d = idx & 15, row = 8 * w + (idx >> 4);
                cplx<float> e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) e[q] = lds[row * RD_PITCH + k1d * 9 + q];
                RegFFT<8, float>::run(e);
                const int kr = 8 * h + (row & 7) + 16 * (row >> 3);
                static_for<8>([&](auto K) {
                    constexpr int k2d = decltype(K)::value;
                    const int kk = (k1d + 16 * k2d) ^ 64;       // fftshift over the 128 Doppler bins
                    if constexpr (SYNC) {
                        // sc1 (write-through, aux = 16) buffer store; 16 lanes write one whole 128-B line.  Plane base
                        // in soffset (scalar), lane part in voffset, the Doppler bin's constant in the immediate;
                        // after an abort the descriptor has zero records and the store is dropped.
                        constexpr int kc = ((16 * k2d) ^ 64) * 8;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, e[bitrev<8>(k2d)]), ring_rs,
                                                              (unsigned)((kr * RD_C + k1d) * 8 + kc), ring_soff, 16);
                    } else {
                        // non-temporal: the cube is written once and is larger than any cache (a plain store here cost the
                        // stand-alone launch 9 %: 1.50 -> 1.37 ms per 15000 planes, found through the sc1 stores of the SYNC form)
                        __builtin_nontemporal_store(e[bitrev<8>(k2d)], &dst[kr * RD_C + kk]);
                    }
                });
            }
            if constexpr (SYNC && !DET && h == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
            __syncthreads();
        });
        if constexpr (DET) {
            unpublished = plane;
            plane = __builtin_amdgcn_readfirstlane(lds_ctl[(iter + 1) & 1]);
        } else if constexpr (SYNC) {
            if (dead) return;
            if (t == 0) __hip_atomic_fetch_add(cs.ctl + CTL_CNT + slot, 1u, MMW_RLX_AGENT);   // plane published
            plane = __builtin_amdgcn_readfirstlane(lds_ctl[(iter + 1) & 1]);
        } else plane += gridDim.x;
    }
    if constexpr (DET) {
        if (unpublished >= 0) {                 // the last plane of this workgroup: drain, then publish
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t0 == 0) __hip_atomic_fetch_add(cs.frame_cnt + unpublished / cs.V, 1u, MMW_RLX_AGENT);
        }
    }
}

// ------------------------------------------------------------------ fused range-Doppler, planes that fit LDS
// k_rd_lds<S, C>: any power-of-two plane of at most 16384 cells (128 KiB) is transformed inside LDS in one pass over
// HBM: coalesced row loads + Hann x Hann -> LDS; range FFT down the columns as two register passes (in place, lanes
// walk the contiguous chirp index, so every LDS access is conflict free); Doppler FFT along the rows, the second
// pass storing straight to global memory with the fftshift folded into the index.  16 cells per thread.
constexpr int rdl_r1(int n) { return n >= 512 ? 32 : n >= 128 ? 16 : n >= 32 ? 8 : 4; }

template <int S, int C> struct RdLds {
    static constexpr int R1S = rdl_r1(S), R2S = S / R1S, R1C = rdl_r1(C), R2C = C / R1C;
    static constexpr int CELLS = S * C;
    static constexpr int NT = CELLS / 16 > 1024 ? 1024 : (CELLS / 16 < 64 ? 64 : CELLS / 16);
    static constexpr int VPT = CELLS / NT;                  // cells per thread (16, or more for 1024-thread planes)
    static constexpr int P = C + R2C;                       // row pitch: strided row reads of pass 1 stay conflict free
    static constexpr int LDS_BYTES = (S * P + S + C) * 8;
    static_assert(R1S >= R2S && R1C >= R2C && R2S >= 2 && R2C >= 2, "radix split");
    static_assert(CELLS <= 16384 && CELLS % NT == 0 && VPT % R2S == 0 && VPT % R2C == 0, "plane must fit the scheme");
};

template <int S, int C>
__global__ __launch_bounds__((RdLds<S, C>::NT)) void k_rd_lds(const f32x4 *__restrict__ in, cplx<float> *__restrict__ out,
                                                               const float *__restrict__ hann_s,
                                                               const float *__restrict__ hann_c,
                                                               const cplx<float> *__restrict__ tw_s_g,
                                                               const cplx<float> *__restrict__ tw_c_g, RawView rv,
                                                               int planes) {
    typedef RdLds<S, C> K;
    constexpr int NT = K::NT, P = K::P, R1S = K::R1S, R2S = K::R2S, R1C = K::R1C, R2C = K::R2C, VPT = K::VPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<float> *lds = reinterpret_cast<cplx<float> *>(smem);
    cplx<float> *tw_s = lds + S * P, *tw_c = tw_s + S;
    const int t = threadIdx.x;
    const bool raw = rv.ntx > 1;                 // uniform: raw cube, de-interleave folded into the loads
    long plane = blockIdx.x;
    if (raw) {
        plane = raw_block_plane(blockIdx.x, planes, rv);
        if (plane < 0 || skip_raw_plane(plane, rv)) return;
    } else plane = skip_block_plane(blockIdx.x, rv);
    const f32x4 *src = in + plane * (K::CELLS / 2);
    const cplx<float> *rsrc = raw ? raw_plane(reinterpret_cast<const cplx<float> *>(in), plane, S, C, rv) : nullptr;
    cplx<float> *dst = out + plane * K::CELLS;
    for (int i = t; i < S; i += NT) tw_s[i] = tw_s_g[i];
    for (int i = t; i < C; i += NT) tw_c[i] = tw_c_g[i];
    // ---- load + window
#pragma unroll
    for (int q = 0; q < VPT / 2; ++q) {
        const int idx = t + q * NT;                         // float4 index: two adjacent chirps of one sample row
        const int row = (2 * idx) / C, col = (2 * idx) % C;
        f32x4 v;
        if (raw) {
            const cplx<float> *e = rsrc + (long)(2 * idx) * rv.ntx;
            const cplx<float> a = e[0], b = e[rv.ntx];
            v = f32x4{a.x, a.y, b.x, b.y};
        } else v = __builtin_nontemporal_load(src + idx);
        const float w0 = hann_s[row] * hann_c[col], w1 = hann_s[row] * hann_c[col + 1];
        *reinterpret_cast<f32x4 *>(&lds[row * P + col]) = f32x4{v.x * w0, v.y * w0, v.z * w1, v.w * w1};
    }
    __syncthreads();
    // ---- range pass 1: items (column c, phase j), samples R2S*n1 + j, in place
    {
        constexpr int ITEMS = C * R2S, ROUNDS = (ITEMS + NT - 1) / NT;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int it = t + r * NT;
            if (ITEMS % NT == 0 || it < ITEMS) {
                const int c = it % C, j = it / C;
                cplx<float> a[R1S];
#pragma unroll
                for (int n1 = 0; n1 < R1S; ++n1) a[n1] = lds[(R2S * n1 + j) * P + c];
                RegFFT<R1S, float>::run(a);
                static_for<R1S>([&](auto K1) {
                    constexpr int k1 = decltype(K1)::value;
                    lds[(R2S * k1 + j) * P + c] = cmul(a[bitrev<R1S>(k1)], tw_s[j * k1]);
                });
            }
        }
    }
    __syncthreads();
    // ---- range pass 2: items (c, k1); results move to rows k1 + R1S*k2, so everything is read before it is written
    {
        constexpr int ITEMS = C * R1S, ROUNDS = ITEMS / NT;
        static_assert(ITEMS % NT == 0 && ROUNDS * R2S == VPT, "range pass 2 covers the plane exactly");
        cplx<float> v[ROUNDS][R2S];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int it = t + r * NT, c = it % C, k1 = it / C;
#pragma unroll
            for (int n2 = 0; n2 < R2S; ++n2) v[r][n2] = lds[(R2S * k1 + n2) * P + c];
            RegFFT<R2S, float>::run(v[r]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int it = t + r * NT, c = it % C, k1 = it / C;
            static_for<R2S>([&](auto K2) {
                constexpr int k2 = decltype(K2)::value;
                lds[(k1 + R1S * k2) * P + c] = v[r][bitrev<R2S>(k2)];
            });
        }
    }
    __syncthreads();
    // ---- Doppler pass 1: items (row s, phase j), chirps R2C*n1 + j, in place
    {
        constexpr int ITEMS = S * R2C, ROUNDS = (ITEMS + NT - 1) / NT;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int it = t + r * NT;
            if (ITEMS % NT == 0 || it < ITEMS) {
                const int j = it % R2C, row = it / R2C;
                cplx<float> a[R1C];
#pragma unroll
                for (int n1 = 0; n1 < R1C; ++n1) a[n1] = lds[row * P + R2C * n1 + j];
                RegFFT<R1C, float>::run(a);
                static_for<R1C>([&](auto K1) {
                    constexpr int k1 = decltype(K1)::value;
                    lds[row * P + R2C * k1 + j] = cmul(a[bitrev<R1C>(k1)], tw_c[j * k1]);
                });
            }
        }
    }
    __syncthreads();
    // ---- Doppler pass 2 + store (fftshift = XOR of the top Doppler bit)
    {
        constexpr int ITEMS = S * R1C, ROUNDS = ITEMS / NT;
        static_assert(ITEMS % NT == 0, "Doppler pass 2 covers the plane exactly");
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int it = t + r * NT, k1 = it % R1C, row = it / R1C;
            cplx<float> a[R2C];
#pragma unroll
            for (int n2 = 0; n2 < R2C; ++n2) a[n2] = lds[row * P + R2C * k1 + n2];
            RegFFT<R2C, float>::run(a);
            static_for<R2C>([&](auto K2) {
                constexpr int k2 = decltype(K2)::value;
                dst[row * C + ((k1 + R1C * k2) ^ (C / 2))] = a[bitrev<R2C>(k2)];
            });
        }
    }
}

template <int S, int C> int launch_rd_lds_sc(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, RawView rv) {
    const void *hs, *hc, *ts, *tc;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, S, &hs));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, C, &hc));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, S, &ts));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, C, &tc));
    typedef RdLds<S, C> K;
    auto go = [&](auto kern) -> int {
        if (K::LDS_BYTES > 64 * 1024)
            MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        K::LDS_BYTES));
        const unsigned grid = rv.ntx > 1 ? (unsigned)raw_grid(planes, rv) : (unsigned)skip_planes(planes, rv);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(K::NT), K::LDS_BYTES, ctx->stream, (const f32x4 *)d_in,
                           (cplx<float> *)d_out, (const float *)hs, (const float *)hc, (const cplx<float> *)ts,
                           (const cplx<float> *)tc, rv, planes);
        return check_launch("rd_lds");
    };
    return go(k_rd_lds<S, C>);
}

// planes handled by k_rd_lds (S x C, both powers of two, S*C <= 16384)
#define MMW_RD_LDS_SHAPES(X) \
    X(32, 32) X(64, 32) X(128, 32) X(256, 32) X(512, 32) X(32, 64) X(64, 64) X(128, 64) X(256, 64) \
    X(32, 128) X(64, 128) X(128, 128)

inline bool rd_lds_supported(int S, int C) {
#define X(s, c) if (S == s && C == c) return true;
    MMW_RD_LDS_SHAPES(X)
#undef X
    return false;
}

// launch_rd_lds / launch_rd_fused instantiate ~15 large kernels; they are compiled once, in mmw_tu_rd.hip
// (MMW_TU_RD), and only declared for the other translation units.
int launch_rd_lds(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, int S, int C, RawView rv = RawView{1, 0});
// d_l1 != nullptr: also l1[plane] (see k_plane_l1); *l1_done tells whether this launch produced it
int launch_rd_fused(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, int S, int C, RawView rv = RawView{1, 0},
                    float *d_l1 = nullptr, bool *l1_done = nullptr);

#ifdef MMW_TU_RD
int launch_rd_lds(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, int S, int C, RawView rv) {
#define X(s, c) if (S == s && C == c) return launch_rd_lds_sc<s, c>(ctx, d_in, d_out, planes, rv);
    MMW_RD_LDS_SHAPES(X)
#undef X
    return set_error(MMW_ERR_UNSUPPORTED, "no LDS-resident RD kernel for %dx%d", S, C);
}
#endif  // MMW_TU_RD

inline bool rd_fused_supported(int S, int C) { return S == RD_S && C == RD_C; }
// the chain's device-synchronised RD stage: `grid` persistent workgroups, n_items = frames * live planes
int launch_rd_fused_sync(mmw_ctx *ctx, const void *d_in, void *d_ring, int n_items, ChainSync cs, int grid);   // cs.ntx > 1: raw input
// the detection pipeline's producer: plain output cube + L1 norms, planes published per frame in cs.frame_cnt
int launch_rd_fused_det(mmw_ctx *ctx, const void *d_in, void *d_out, float *d_l1, int n_planes, ChainSync cs, int grid);
// Counters of a ticketed range-Doppler launch outside the chain (tickets, abort word, one counter per frame) + a float per plane
// for callers that do not want the L1 norms: an allocation of the context's own (mmw_detect_points runs such launches beside a
// pending tail that owns the scratch).  Grows by reallocation behind a synchronisation of the queues that may use it.
inline int ensure_help_sync(mmw_ctx *ctx, size_t words) {
    if (ctx->help_sync_words >= words) return MMW_OK;
    if (ctx->help_sync) {
        if (ctx->q_tail) MMW_HIP(hipStreamSynchronize(ctx->q_tail));
        MMW_HIP(hipStreamSynchronize(ctx->stream));
        MMW_HIP(hipFree(ctx->help_sync));
        ctx->help_sync = nullptr;
        ctx->help_sync_words = 0;
    }
    MMW_HIP(hipMalloc((void **)&ctx->help_sync, words * sizeof(unsigned)));
    ctx->help_sync_words = words;
    return MMW_OK;
}

#ifdef MMW_TU_RD
int launch_rd_fused_sync(mmw_ctx *ctx, const void *d_in, void *d_ring, int n_items, ChainSync cs, int grid) {
    const void *hs, *hc, *t256, *t128;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, RD_S, &hs));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, RD_C, &hc));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, 256, &t256));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, 128, &t128));
    auto go = [&](auto kern) -> int {
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS_BYTES));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), RD_LDS_BYTES, ctx->stream, (const f32x4 *)d_in, (cplx<float> *)d_ring,
                           n_items, (const float *)hs, (const float *)hc, (const cplx<float> *)t256, (const cplx<float> *)t128, cs,
                           (float *)nullptr);
        return MMW_OK;
    };
    if (cs.ntx > 1 && cs.i16) MMW_TRY(go(k_rd_fused_256x128_persist<false, 4, true, false, true, true>));
    else if (cs.ntx > 1) MMW_TRY(go(k_rd_fused_256x128_persist<false, 4, true, false, true>));
    else MMW_TRY(go(k_rd_fused_256x128_persist<true, 6, true>));
    return check_launch("rd_fused_sync");
}

int launch_rd_fused_det(mmw_ctx *ctx, const void *d_in, void *d_out, float *d_l1, int n_planes, ChainSync cs, int grid) {
    const void *hs, *hc, *t256, *t128;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, RD_S, &hs));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, RD_C, &hc));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, 256, &t256));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, 128, &t128));
    auto kern = k_rd_fused_256x128_persist<true, 4, true, true, false, false, true>;
    MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), RD_LDS_BYTES, ctx->stream, (const f32x4 *)d_in, (cplx<float> *)d_out, n_planes,
                       (const float *)hs, (const float *)hc, (const cplx<float> *)t256, (const cplx<float> *)t128, cs, d_l1);
    return check_launch("rd_fused_det");
}

int launch_rd_fused(mmw_ctx *ctx, const void *d_in, void *d_out, int planes, int S, int C, RawView rv, float *d_l1, bool *l1_done) {
    if (l1_done) *l1_done = false;
    if (!rd_fused_supported(S, C)) return set_error(MMW_ERR_UNSUPPORTED, "fused RD kernel is 256x128 only");
    const void *hs, *hc, *t256, *t128;
    MMW_TRY(get_table<float>(ctx, TAB_HANN, RD_S, &hs));
    MMW_TRY(get_table<float>(ctx, TAB_HANN, RD_C, &hc));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, 256, &t256));
    MMW_TRY(get_table<float>(ctx, TAB_TWIDDLE, 128, &t128));
    if (!ctx->rd_attr_set) {
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rd_fused_256x128<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS_BYTES));
        MMW_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rd_fused_256x128<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS_BYTES));
        ctx->rd_attr_set = true;
    }
    if (rv.ntx > 1) {            // raw cube: the de-interleave (and the int16 -> float conversion) is folded into the row loads
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS_BYTES);
            hipLaunchKernelGGL(kern, dim3((unsigned)raw_grid(planes, rv)), dim3(1024), RD_LDS_BYTES, ctx->stream,
                               (const f32x4 *)d_in, (cplx<float> *)d_out, planes, (const float *)hs, (const float *)hc,
                               (const cplx<float> *)t256, (const cplx<float> *)t128, rv, ctx->active_cus > 0 ? 0 : 1);
        };
        if (rv.i16) go(k_rd_fused_256x128<false, 0, true, true>);
        else go(k_rd_fused_256x128<false, 0, true>);
        return check_launch("rd_fused_raw");
    }
    const int blocks = (int)skip_planes(planes, rv);   // one plane per workgroup
    // Standalone launches use the persistent variant with 8 rows prefetched (+12 %); inside the overlapped
    // chain the one-plane-per-workgroup kernel is faster (measured), so the chain sets active_cus and gets it.
    const int pf = ctx->active_cus > 0 ? 0 : 8;
    if (pf == 8 && rv.vskip <= 2) {
        int grid = (ctx->active_cus > 0 ? ctx->active_cus : ctx->num_cu);
        if (ctx->rd_leave_cus > 0 && grid > 2 * ctx->rd_leave_cus) grid -= ctx->rd_leave_cus;      // (one workgroup fills a CU)
        if (grid > planes) grid = planes;
        auto launch = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS_BYTES);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), RD_LDS_BYTES, ctx->stream, (const f32x4 *)d_in,
                               (cplx<float> *)d_out, planes, (const float *)hs, (const float *)hc,
                               (const cplx<float> *)t256, (const cplx<float> *)t128, ChainSync{}, (float *)nullptr);
        };
        if (d_l1) {
            auto kern = k_rd_fused_256x128_persist<true, 4, false, true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RD_LDS_BYTES);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), RD_LDS_BYTES, ctx->stream, (const f32x4 *)d_in,
                               (cplx<float> *)d_out, planes, (const float *)hs, (const float *)hc,
                               (const cplx<float> *)t256, (const cplx<float> *)t128, ChainSync{}, d_l1);
            if (l1_done) *l1_done = true;
        } else launch(k_rd_fused_256x128_persist<true, 6>);     // 6 rows prefetched: 8 no longer fit the register budget (spills)
        return check_launch("rd_fused_persist");
    }
    hipLaunchKernelGGL(k_rd_fused_256x128<true>, dim3(blocks), dim3(1024), RD_LDS_BYTES, ctx->stream,
                       (const f32x4 *)d_in, (cplx<float> *)d_out, planes, (const float *)hs, (const float *)hc,
                       (const cplx<float> *)t256, (const cplx<float> *)t128, rv, ctx->active_cus > 0 ? 0 : 1);
    return check_launch("rd_fused");
}
#endif  // MMW_TU_RD

}  // namespace mmw
