// Host-side check of the in-register FFT network (mmw_fft.h) against a direct O(N^2) DFT.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "mmw_fft.h"
using namespace mmw;

template <int N, typename T> double check() {
    cplx<T> a[N];
    double xr[N], xi[N];
    for (int i = 0; i < N; ++i) {
        xr[i] = (double)rand() / RAND_MAX - 0.5;
        xi[i] = (double)rand() / RAND_MAX - 0.5;
        a[i] = {(T)xr[i], (T)xi[i]};
        xr[i] = (double)a[i].x; xi[i] = (double)a[i].y;
    }
    RegFFT<N, T>::run(a);
    double err = 0, mx = 0;
    for (int k = 0; k < N; ++k) {
        double sr = 0, si = 0;
        for (int n = 0; n < N; ++n) {
            double ang = -2.0 * M_PI * ((n * k) % N) / N;
            sr += xr[n] * cos(ang) - xi[n] * sin(ang);
            si += xr[n] * sin(ang) + xi[n] * cos(ang);
        }
        cplx<T> g = a[bitrev<N>(k)];
        err = fmax(err, hypot((double)g.x - sr, (double)g.y - si));
        mx = fmax(mx, hypot(sr, si));
    }
    return err / mx;
}

int main() {
    int bad = 0;
#define CHK(N) { double ef = check<N, float>(), ed = check<N, double>(); \
    printf("N=%d rel err f32 %.3g f64 %.3g\n", N, ef, ed); if (ef > 2e-6 || ed > 1e-14) bad = 1; }
    CHK(2) CHK(4) CHK(8) CHK(16) CHK(32) CHK(64)
    return bad;
}
