// Host-side check of the any-length in-register DFT (mmw_dft_small.h) against a direct O(N^2) DFT, R = 1..32 and a few more.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "mmw_dft_small.h"
using namespace mmw;

template <int N, typename T> double check() {
    cplx<T> a[N];
    double xr[N], xi[N];
    for (int i = 0; i < N; ++i) {
        a[i] = {(T)((double)rand() / RAND_MAX - 0.5), (T)((double)rand() / RAND_MAX - 0.5)};
        xr[i] = (double)a[i].x;
        xi[i] = (double)a[i].y;
    }
    RegDFT<N, T>::run(a);
    double err = 0, mx = 0;
    for (int k = 0; k < N; ++k) {
        double sr = 0, si = 0;
        for (int n = 0; n < N; ++n) {
            const double ang = -2.0 * M_PI * ((n * k) % N) / N;
            sr += xr[n] * cos(ang) - xi[n] * sin(ang);
            si += xr[n] * sin(ang) + xi[n] * cos(ang);
        }
        err = fmax(err, hypot((double)a[k].x - sr, (double)a[k].y - si));
        mx = fmax(mx, hypot(sr, si));
    }
    return err / mx;
}

template <int N> int one() {
    const double ef = check<N, float>(), ed = check<N, double>();
    printf("R=%d rel err f32 %.3g f64 %.3g\n", N, ef, ed);
    return (ef > 3e-6 || ed > 2e-14) ? 1 : 0;
}

template <int... N> int all(std::integer_sequence<int, N...>) { return (one<N + 1>() + ...); }

int main() {
    int bad = all(std::make_integer_sequence<int, 32>{});
    bad += one<35>() + one<45>() + one<49>() + one<63>();
    // the compile-time trigonometry against libm
    double worst = 0;
    for (int n = 1; n <= 64; ++n)
        for (int k = -3; k <= n + 3; ++k) {
            worst = fmax(worst, fabs((double)dftc::cos2pi(k, n) - cos(2.0 * M_PI * k / n)));
            worst = fmax(worst, fabs((double)dftc::sin2pi(k, n) - sin(2.0 * M_PI * k / n)));
        }
    printf("cos2pi/sin2pi worst abs err %.3g\n", worst);
    return (bad || worst > 4e-15) ? 1 : 0;
}
