// In-register DFTs of ANY small length R (host + device compilable), natural order in and out.
//
// The shipped TI configs give axis lengths like 63, 70, 90, 100, 126, 130, 200: their factors
// 2..16 (and up to 32) are the radices of the mixed-radix range-Doppler kernel (mmw_fft_mixed.h).  Round 1 evaluated
// every R-point level as a direct R x R matrix product (R complex MACs per point).  RegDFT<R> builds the transform at
// compile time instead:
//   R = 2^k            the radix-2 DIF network of mmw_fft.h (RegFFT), un-bit-reversed through constant indices;
//   R an odd prime     the real-symmetric form: with s_j = x_j + x_{R-j}, d_j = x_j - x_{R-j},
//                        X_k, X_{R-k} = (x_0 + sum_j cos(2 pi jk/R) s_j)  -/+  i (sum_j sin(2 pi jk/R) d_j)
//                      -- (R-1)^2 / 2 complex-by-REAL multiply-adds instead of R^2 complex ones;
//   R = a * b coprime  Good-Thomas prime-factor mapping (no twiddles): n = (b n1 + a n2) mod R, k = CRT(k1, k2);
//   R = p^m            Cooley-Tukey with compile-time twiddles.
// Every index is a constant expression, so the arrays stay in registers; every coefficient is a literal.
// Forward transform e^{-j 2 pi nk/R}, unnormalised (numpy.fft.fft, processors/range_doppler_resp.py:98-103).
#pragma once
#include "mmw_fft.h"

namespace mmw {
namespace dftc {

// ---- compile-time cos / sin of 2 pi k / n (long double Taylor series after octant reduction; exact at the octant points)
constexpr long double PI = 3.14159265358979323846264338327950288L;
constexpr long double taylor_sin(long double x) {       // |x| <= pi / 4
    long double term = x, sum = x;
    for (int i = 1; i < 14; ++i) {
        term *= -x * x / ((2 * i) * (2 * i + 1));
        sum += term;
    }
    return sum;
}
constexpr long double taylor_cos(long double x) {
    long double term = 1, sum = 1;
    for (int i = 1; i < 14; ++i) {
        term *= -x * x / ((2 * i - 1) * (2 * i));
        sum += term;
    }
    return sum;
}
constexpr long double sin2pi(long k, long n);
constexpr long double cos2pi(long k, long n) {
    k %= n;
    if (k < 0) k += n;
    if (2 * k > n) k = n - k;                               // cos(2 pi (1 - f)) = cos(2 pi f)
    if (4 * k > n) return -cos2pi(n - 2 * k, 2 * n);        // cos(pi - t) = -cos t, f in (1/4, 1/2]
    if (8 * k > n) return sin2pi(n - 4 * k, 4 * n);         // cos(pi/2 - t) = sin t
    if (k == 0) return 1.0L;
    return taylor_cos(2 * PI * (long double)k / (long double)n);
}
constexpr long double sin2pi(long k, long n) {
    k %= n;
    if (k < 0) k += n;
    if (2 * k > n) return -sin2pi(n - k, n);
    if (4 * k > n) return sin2pi(n - 2 * k, 2 * n);         // sin(pi - t) = sin t
    if (8 * k > n) return cos2pi(n - 4 * k, 4 * n);
    if (k == 0) return 0.0L;
    return taylor_sin(2 * PI * (long double)k / (long double)n);
}

constexpr int smallest_prime_factor(int n) {
    for (int d = 2; d * d <= n; ++d)
        if (n % d == 0) return d;
    return n;
}
constexpr bool prime(int n) { return n >= 2 && smallest_prime_factor(n) == n; }
// largest power of n's smallest prime factor dividing n
constexpr int prime_power_part(int n) {
    const int p = smallest_prime_factor(n);
    int q = 1;
    while (n % p == 0) {
        n /= p;
        q *= p;
    }
    return q;
}
constexpr int inv_mod(int a, int m) {       // a^-1 mod m (coprime, m small)
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 1;
}

}  // namespace dftc

// a * (c - j s) with literal c, s
template <typename T, typename CT> MMW_HD CT cmul_const(CT a, T c, T s) { return CT{a.x * c + a.y * s, a.y * c - a.x * s}; }

template <int R, typename T, typename CT = cplx<T>> struct RegDFT {
    static_assert(R >= 1 && R <= 64, "register DFT lengths are 1..64");

    static MMW_HD void run(CT (&x)[R]) {
        if constexpr (R == 1) {
        } else if constexpr (R == 2) {
            const CT a = x[0], b = x[1];
            x[0] = a + b;
            x[1] = a - b;
        } else if constexpr (is_pow2(R)) {
            RegFFT<R, T, R, 0, CT>::run(x);
            CT y[R];
            static_for<R>([&](auto K) { y[decltype(K)::value] = x[bitrev<R>(decltype(K)::value)]; });
            static_for<R>([&](auto K) { x[decltype(K)::value] = y[decltype(K)::value]; });
        } else if constexpr (dftc::prime(R)) {
            run_prime(x);
        } else {
            constexpr int A = dftc::prime_power_part(R), B = R / A;
            if constexpr (B == 1) run_cooley_tukey(x);      // R = p^m
            else run_good_thomas<A, B>(x);
        }
    }

    // odd prime: real-symmetric pairs
    static MMW_HD void run_prime(CT (&x)[R]) {
        constexpr int H = (R - 1) / 2;
        CT s[H], d[H];
        static_for<H>([&](auto J) {
            constexpr int j = decltype(J)::value + 1;
            s[j - 1] = x[j] + x[R - j];
            d[j - 1] = x[j] - x[R - j];
        });
        const CT x0 = x[0];
        CT sum = x0;
        static_for<H>([&](auto J) { sum = sum + s[decltype(J)::value]; });
        x[0] = sum;
        static_for<H>([&](auto K) {
            constexpr int k = decltype(K)::value + 1;
            CT t = x0, u = CT{(T)0, (T)0};
            static_for<H>([&](auto J) {
                constexpr int j = decltype(J)::value + 1;
                constexpr T c = (T)dftc::cos2pi((long)j * k, R), sn = (T)dftc::sin2pi((long)j * k, R);
                t = t + s[j - 1] * c;
                u = u + d[j - 1] * sn;
            });
            x[k] = CT{t.x + u.y, t.y - u.x};            // t - j u
            x[R - k] = CT{t.x - u.y, t.y + u.x};        // t + j u
        });
    }

    // R = A * B, gcd(A, B) = 1: n = (B n1 + A n2) mod R, k = (k1 B (B^-1 mod A) + k2 A (A^-1 mod B)) mod R
    template <int A, int B> static MMW_HD void run_good_thomas(CT (&x)[R]) {
        CT y[A][B];
        static_for<B>([&](auto N2) {
            constexpr int n2 = decltype(N2)::value;
            CT col[A];
            static_for<A>([&](auto N1) { col[decltype(N1)::value] = x[(B * decltype(N1)::value + A * n2) % R]; });
            RegDFT<A, T, CT>::run(col);
            static_for<A>([&](auto K1) { y[decltype(K1)::value][n2] = col[decltype(K1)::value]; });
        });
        constexpr int EA = B * dftc::inv_mod(B % A, A), EB = A * dftc::inv_mod(A % B, B);
        static_for<A>([&](auto K1) {
            constexpr int k1 = decltype(K1)::value;
            CT row[B];
            static_for<B>([&](auto N2) { row[decltype(N2)::value] = y[k1][decltype(N2)::value]; });
            RegDFT<B, T, CT>::run(row);
            static_for<B>([&](auto K2) { x[(k1 * EA + decltype(K2)::value * EB) % R] = row[decltype(K2)::value]; });
        });
    }

    // R = P * Q (P = smallest prime factor): n = Q n1 + n2, k = k1 + P k2, twiddle W_R^(n2 k1)
    static MMW_HD void run_cooley_tukey(CT (&x)[R]) {
        constexpr int P = dftc::smallest_prime_factor(R), Q = R / P;
        CT y[P][Q];
        static_for<Q>([&](auto N2) {
            constexpr int n2 = decltype(N2)::value;
            CT col[P];
            static_for<P>([&](auto N1) { col[decltype(N1)::value] = x[Q * decltype(N1)::value + n2]; });
            RegDFT<P, T, CT>::run(col);
            static_for<P>([&](auto K1) {
                constexpr int k1 = decltype(K1)::value;
                if constexpr (k1 == 0 || n2 == 0) y[k1][n2] = col[k1];
                else {
                    constexpr T c = (T)dftc::cos2pi((long)n2 * k1, R), sn = (T)dftc::sin2pi((long)n2 * k1, R);
                    y[k1][n2] = cmul_const<T, CT>(col[k1], c, sn);
                }
            });
        });
        static_for<P>([&](auto K1) {
            constexpr int k1 = decltype(K1)::value;
            CT row[Q];
            static_for<Q>([&](auto N2) { row[decltype(N2)::value] = y[k1][decltype(N2)::value]; });
            RegDFT<Q, T, CT>::run(row);
            static_for<Q>([&](auto K2) { x[k1 + P * decltype(K2)::value] = row[decltype(K2)::value]; });
        });
    }
};

}  // namespace mmw
