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

template <int VIN, bool MAG>
__global__ __launch_bounds__(256) void k_angle64(const float4 *__restrict__ rd, void *__restrict__ out,
                                                  long pairs_per_frame, AngleWin win) {
    // grid.x covers pairs of adjacent bins of one frame; grid.y = frame
    const long pair = (long)blockIdx.x * 256 + threadIdx.x;
    if (pair >= pairs_per_frame) return;
    const long f = blockIdx.y;
    const float4 *src = rd + f * VIN * pairs_per_frame + pair;
    cplx<float> xa[VIN], xb[VIN];
#pragma unroll
    for (int v = 0; v < VIN; ++v) {
        const float4 t = src[(long)v * pairs_per_frame];
        const float h = win.h[v];
        xa[v] = {t.x * h, t.y * h};
        xb[v] = {t.z * h, t.w * h};
    }
    static_for<8>([&](auto K1) {
        constexpr int k1 = decltype(K1)::value;
        cplx<float> za[8], zb[8];
        static_for<8>([&](auto N2) {
            constexpr int n2 = decltype(N2)::value;
            cplx<float> ya = {0.f, 0.f}, yb = {0.f, 0.f};
            if constexpr (n2 < VIN) {
                ya = xa[n2];
                yb = xb[n2];
            }
            if constexpr (n2 + 8 < VIN) {
                ya = ya + mul_w<8, k1, float>(xa[n2 + 8]);
                yb = yb + mul_w<8, k1, float>(xb[n2 + 8]);
            }
            za[n2] = mul_w<64, n2 * k1, float>(ya);
            zb[n2] = mul_w<64, n2 * k1, float>(yb);
        });
        RegFFT<8, float>::run(za);
        RegFFT<8, float>::run(zb);
        static_for<8>([&](auto K2) {
            constexpr int k2 = decltype(K2)::value;
            constexpr int a = (k1 + 8 * k2 + 32) % 64;   // fftshift over the angle axis
            const cplx<float> va = za[bitrev<8>(k2)], vb = zb[bitrev<8>(k2)];
            const long o = (f * 64 + a) * pairs_per_frame + pair;
            if constexpr (MAG) {
                reinterpret_cast<float2 *>(out)[o] = make_float2(hypotf(va.x, va.y), hypotf(vb.x, vb.y));
            } else {
                reinterpret_cast<float4 *>(out)[o] = make_float4(va.x, va.y, vb.x, vb.y);
            }
        });
    });
}

template <int VIN>
int launch_angle64(mmw_ctx *ctx, const void *rd, void *out, int F, long bins, bool mag, const float *h) {
    AngleWin w;
    for (int i = 0; i < 16; ++i) w.h[i] = i < VIN ? h[i] : 0.f;
    const long pairs = bins / 2;
    dim3 grid((unsigned)((pairs + 255) / 256), (unsigned)F);
    if (mag)
        hipLaunchKernelGGL((k_angle64<VIN, true>), grid, dim3(256), 0, ctx->stream, (const float4 *)rd, out, pairs, w);
    else
        hipLaunchKernelGGL((k_angle64<VIN, false>), grid, dim3(256), 0, ctx->stream, (const float4 *)rd, out, pairs, w);
    return check_launch("angle64");
}

// fused range-Doppler kernel: filled in below once measured
inline int launch_rd_fused(mmw_ctx *, const void *, void *, int, int, int) {
    return set_error(MMW_ERR_UNSUPPORTED, "fused RD kernel not built");
}

}  // namespace mmw
