// In-register FFT building blocks (host + device compilable).
//
// RegFFT<N>::run(a) transforms a[0..N) in place with a decimation-in-frequency
// radix-2 network whose twiddles are compile-time constants; multiplications by
// 1, -j, -1, +j and (+-1 +- j)/sqrt(2) are specialised away.  Output X[k] lands in
// a[bitrev<N>(k)], so callers read/write with compile-time permuted indices and
// the array stays in VGPRs.  Forward transform, e^{-j 2 pi nk/N}, unnormalised --
// the convention of numpy.fft.fft that the reference calls
// (processors/range_doppler_resp.py:98-103).
#pragma once
#include <utility>
#include "mmw_twiddle_const.h"

#if defined(__HIPCC__)
#define MMW_HD __host__ __device__ __forceinline__
#else
#define MMW_HD inline
#endif

namespace mmw {

// Complex numbers are native 2-wide vectors (re, im): + - and scalar * are the built-in element-wise
// operators, which hipcc lowers to v_pk_add_f32 / v_pk_mul_f32 on the register pair without shuffles.
// The complex product is the named function cmul (the built-in vector * is element-wise!).
#if defined(__clang__)
template <typename T> using cplx = T __attribute__((ext_vector_type(2)));
#else
template <typename T> struct cplx_s {
    T x, y;
    cplx_s operator+(cplx_s b) const { return {x + b.x, y + b.y}; }
    cplx_s operator-(cplx_s b) const { return {x - b.x, y - b.y}; }
    cplx_s operator*(T s) const { return {x * s, y * s}; }
};
template <typename T> using cplx = cplx_s<T>;
#endif

template <typename T> MMW_HD cplx<T> cmul(cplx<T> a, cplx<T> b) {
    return cplx<T>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
constexpr bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
constexpr int bitrev_bits(int k, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((k >> i) & 1) << (bits - 1 - i);
    return r;
}
template <int N> constexpr int bitrev(int k) { return bitrev_bits(k, ilog2(N)); }

// a * W_N^K with W_N = exp(-j 2 pi / N), K compile-time, N | 64.  CT = complex type (any {x, y} type with + and -).
template <int N, int K, typename T, typename CT = cplx<T>> MMW_HD CT mul_w(CT a) {
    constexpr int k = ((K % N) + N) % N;
    static_assert(64 % N == 0 || N % 64 == 0, "compile-time twiddles cover N | 64");
    if constexpr (k == 0) {
        return a;
    } else if constexpr (4 * k == N) {
        return CT{a.y, -a.x};
    } else if constexpr (2 * k == N) {
        return CT{-a.x, -a.y};
    } else if constexpr (4 * k == 3 * N) {
        return CT{-a.y, a.x};
    } else if constexpr (8 * k == N) {
        constexpr T r = (T)0.70710678118654752440;
        return CT{(a.x + a.y) * r, (a.y - a.x) * r};
    } else if constexpr (8 * k == 3 * N) {
        constexpr T r = (T)0.70710678118654752440;
        return CT{(a.y - a.x) * r, -(a.x + a.y) * r};
    } else {
        constexpr T c = (T)twc::C64[k * (64 / N)];
        constexpr T s = (T)twc::S64[k * (64 / N)];
        return CT{a.x * c + a.y * s, a.y * c - a.x * s};
    }
}

template <int N, int OFF, int TOT, typename T, typename CT> struct DifStage {
    template <int... K>
    static MMW_HD void butterflies(CT (&a)[TOT], std::integer_sequence<int, K...>) {
        constexpr int H = N / 2;
        ((void)([&] {
            CT u = a[OFF + K], v = a[OFF + K + H];
            a[OFF + K] = u + v;
            a[OFF + K + H] = mul_w<N, K, T, CT>(u - v);
        }()), ...);
    }
    static MMW_HD void run(CT (&a)[TOT]) {
        if constexpr (N >= 2) {
            butterflies(a, std::make_integer_sequence<int, N / 2>{});
            DifStage<N / 2, OFF, TOT, T, CT>::run(a);
            DifStage<N / 2, OFF + N / 2, TOT, T, CT>::run(a);
        }
    }
};

// In-place N-point forward DFT of a[OFF..OFF+N) inside an array of TOT registers.
template <int N, typename T, int TOT = N, int OFF = 0, typename CT = cplx<T>> struct RegFFT {
    static_assert(is_pow2(N) && N <= 64, "register FFT sizes are powers of two up to 64");
    static MMW_HD void run(CT (&a)[TOT]) { DifStage<N, OFF, TOT, T, CT>::run(a); }
};

// compile-time loop helper: f(std::integral_constant<int, I>) for I in [0, N)
template <int N, typename F, int... I>
MMW_HD void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    ((void)f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> MMW_HD void static_for(F &&f) {
    static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

}  // namespace mmw
