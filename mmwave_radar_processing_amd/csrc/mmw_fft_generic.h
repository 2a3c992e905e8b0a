// Generic batched 1-D FFT along one axis of a strided 3-D view [outer][axis][inner].
//
// Three kernels behind one dispatcher (launch_fft_axis):
//   k_fft_strided : power-of-two N, inner index contiguous in memory (range FFT over samples,
//                   angle FFT over antennas).  Lanes walk the inner index (coalesced), each thread
//                   runs an R1-point register FFT, one LDS exchange, then R2-point register FFTs.
//   k_fft_contig  : power-of-two N along the contiguous axis (Doppler FFT over chirps).
//   k_dft_direct  : any N (the shipped TI cfgs have 63/70/90/100/127/... samples and loops,
//                   SURVEY.md F5): O(N^2) table-driven DFT out of LDS.
// Window multiply (np.hanning tables), zero padding (n_in < N), fftshift of the output index and
// |.| are folded into the load / store so no stage makes an extra pass over HBM.
// The fused fast paths for the headline shape live in mmw_fft_fused.h; these kernels are the
// any-shape path and the float64 CFAR-plane path.
#pragma once
#include "mmw_ctx.h"

namespace mmw {

struct FftArgs {
    const void *in;
    void *out;
    long in_outer_stride, in_axis_stride, in_inner_stride;     // complex elements
    long out_outer_stride, out_axis_stride, out_inner_stride;  // output elements
    int outer;        // batch count over the outer index
    int inner;        // count of inner indices
    int n_in;         // valid input length along the axis (<= N); the rest is zero padding
    const void *win_axis;   // T[n_in] or nullptr
    const void *win_inner;  // T[inner] or nullptr
    const void *tw;         // cplx<T>[N]: W_N^m
    double scale;           // extra scalar on the input
    int shift;              // 1: output index k -> (k + N/2) mod N   (np.fft.fftshift, even N)
    int magnitude;          // 1: write |X| as T instead of cplx<T>
};

template <typename T> __device__ __forceinline__ T mag(cplx<T> v);
template <> __device__ __forceinline__ float mag<float>(cplx<float> v) { return hypotf(v.x, v.y); }
template <> __device__ __forceinline__ double mag<double>(cplx<double> v) { return hypot(v.x, v.y); }

template <typename T> __device__ __forceinline__ void store_out(const FftArgs &p, long off, cplx<T> v) {
    if (p.magnitude)
        ((T *)p.out)[off] = mag<T>(v);
    else
        ((cplx<T> *)p.out)[off] = v;
}

constexpr int strided_tile(int N, int elem_bytes) {
    int b = 65536 / (N * elem_bytes);
    return b > 32 ? 32 : (b < 1 ? 1 : b);
}

constexpr int contig_tile(int R1, int R2, int elem_bytes) {
    int b = 256 / R2;
    const int cap = 49152 / (R1 * (R2 + 1) * elem_bytes);
    if (b > cap) b = cap;
    return b < 1 ? 1 : b;
}

// ------------------------------------------------------------------ inner-contiguous axis FFT
template <typename T, typename TIN, int N, int R1, int R2, int B>
__global__ __launch_bounds__(B *R2) void k_fft_strided(FftArgs p) {
    static_assert(R1 * R2 == N && R1 >= R2 && R1 % R2 == 0, "N = R1*R2, R1 >= R2");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T> *lds = reinterpret_cast<cplx<T> *>(smem);  // [N][B]
    const int tid = threadIdx.x;
    const int b = tid % B, j = tid / B;
    const int tiles = (p.inner + B - 1) / B;
    const int outer = blockIdx.x / tiles;
    const long i = (long)(blockIdx.x % tiles) * B + b;
    const bool active = i < p.inner;
    const cplx<TIN> *in = reinterpret_cast<const cplx<TIN> *>(p.in);
    const cplx<T> *tw = reinterpret_cast<const cplx<T> *>(p.tw);
    const T *wa = reinterpret_cast<const T *>(p.win_axis);
    const T *wi = reinterpret_cast<const T *>(p.win_inner);

    cplx<T> a[R1];
    T sc = (T)p.scale;
    if (active && wi) sc *= wi[i];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
        const int n = R2 * n1 + j;
        cplx<T> v = cplx<T>{(T)0, (T)0};
        if (active && n < p.n_in) {
            const cplx<TIN> x = in[(long)outer * p.in_outer_stride + (long)n * p.in_axis_stride + i * p.in_inner_stride];
            const T w = wa ? wa[n] * sc : sc;
            v = cplx<T>{(T)x.x, (T)x.y} * w;
        }
        a[n1] = v;
    }
    RegFFT<R1, T>::run(a);
    static_for<R1>([&](auto K1) {
        constexpr int k1 = decltype(K1)::value;
        lds[(k1 * R2 + j) * B + b] = cmul(a[bitrev<R1>(k1)], tw[j * k1]);
    });
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R1 / R2; ++q) {
        const int k1 = j + R2 * q;
        cplx<T> c[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = lds[(k1 * R2 + n2) * B + b];
        RegFFT<R2, T>::run(c);
        if (active) {
            static_for<R2>([&](auto K2) {
                constexpr int k2 = decltype(K2)::value;
                int k = k1 + R1 * k2;
                if (p.shift) k = (k + N / 2) % N;
                store_out<T>(p, (long)outer * p.out_outer_stride + (long)k * p.out_axis_stride + i * p.out_inner_stride,
                             c[bitrev<R2>(k2)]);
            });
        }
    }
}

// ------------------------------------------------------------------ contiguous-axis FFT (rows)
template <typename T, typename TIN, int N, int R1, int R2, int B>
__global__ __launch_bounds__(B *R2) void k_fft_contig(FftArgs p) {
    static_assert(R1 * R2 == N && R1 >= R2 && R1 % R2 == 0, "N = R1*R2, R1 >= R2");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T> *lds = reinterpret_cast<cplx<T> *>(smem);  // [B][R1][R2+1]
    constexpr int P = R2 + 1;
    const int tid = threadIdx.x;
    const int j = tid % R2, b = tid / R2;
    const long row = (long)blockIdx.x * B + b;
    const bool active = row < p.outer;
    const cplx<TIN> *in = reinterpret_cast<const cplx<TIN> *>(p.in);
    const cplx<T> *tw = reinterpret_cast<const cplx<T> *>(p.tw);
    const T *wa = reinterpret_cast<const T *>(p.win_axis);

    cplx<T> a[R1];
    const T sc = (T)p.scale;
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
        const int n = R2 * n1 + j;
        cplx<T> v = cplx<T>{(T)0, (T)0};
        if (active && n < p.n_in) {
            const cplx<TIN> x = in[row * p.in_outer_stride + (long)n * p.in_axis_stride];
            const T w = wa ? wa[n] * sc : sc;
            v = cplx<T>{(T)x.x, (T)x.y} * w;
        }
        a[n1] = v;
    }
    RegFFT<R1, T>::run(a);
    static_for<R1>([&](auto K1) {
        constexpr int k1 = decltype(K1)::value;
        lds[(b * R1 + k1) * P + j] = cmul(a[bitrev<R1>(k1)], tw[j * k1]);
    });
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R1 / R2; ++q) {
        const int item = tid + q * (B * R2);
        const int k1 = item % R1, bb = item / R1;
        const long orow = (long)blockIdx.x * B + bb;
        cplx<T> c[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) c[n2] = lds[(bb * R1 + k1) * P + n2];
        RegFFT<R2, T>::run(c);
        if (orow < p.outer) {
            static_for<R2>([&](auto K2) {
                constexpr int k2 = decltype(K2)::value;
                int k = k1 + R1 * k2;
                if (p.shift) k = (k + N / 2) % N;
                store_out<T>(p, orow * p.out_outer_stride + (long)k * p.out_axis_stride, c[bitrev<R2>(k2)]);
            });
        }
    }
}

// ------------------------------------------------------------------ any-N direct DFT
// One workgroup: B inner columns of one outer index; x[n_in][B] staged in LDS (windowed), thread
// (b, kq) accumulates outputs k = kq, kq + KT, ... with the twiddle index (n*k) mod N kept exact.
template <typename T, typename TIN>
__global__ __launch_bounds__(256) void k_dft_direct(FftArgs p, int N, int B) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx<T> *x = reinterpret_cast<cplx<T> *>(smem);  // [n_in][B]
    const int tid = threadIdx.x;
    const int b = tid % B, kq = tid / B, KT = 256 / B;
    const int tiles = (p.inner + B - 1) / B;
    const int outer = blockIdx.x / tiles;
    const long i = (long)(blockIdx.x % tiles) * B + b;
    const bool active = i < p.inner;
    const cplx<TIN> *in = reinterpret_cast<const cplx<TIN> *>(p.in);
    const cplx<T> *tw = reinterpret_cast<const cplx<T> *>(p.tw);
    const T *wa = reinterpret_cast<const T *>(p.win_axis);
    const T *wi = reinterpret_cast<const T *>(p.win_inner);
    T sc = (T)p.scale;
    if (active && wi) sc *= wi[i];
    for (int n = kq; n < p.n_in; n += KT) {
        cplx<T> v = cplx<T>{(T)0, (T)0};
        if (active) {
            const cplx<TIN> xi = in[(long)outer * p.in_outer_stride + (long)n * p.in_axis_stride + i * p.in_inner_stride];
            const T w = wa ? wa[n] * sc : sc;
            v = cplx<T>{(T)xi.x, (T)xi.y} * w;
        }
        x[n * B + b] = v;
    }
    __syncthreads();
    if (!active) return;
    for (int k = kq; k < N; k += KT) {
        cplx<T> acc = cplx<T>{(T)0, (T)0};
        int idx = 0;
        for (int n = 0; n < p.n_in; ++n) {
            acc = acc + cmul(x[n * B + b], tw[idx]);
            idx += k;
            if (idx >= N) idx -= N;
        }
        int kk = k;
        if (p.shift) kk = (k + N / 2) % N;   // np.fft.fftshift: out[(k + N//2) % N] = X[k], odd N too
        store_out<T>(p, (long)outer * p.out_outer_stride + (long)kk * p.out_axis_stride + i * p.out_inner_stride, acc);
    }
}

// ------------------------------------------------------------------ dispatch
template <typename T, typename TIN, int N, int R1, int R2, int B>
int launch_contig_b(mmw_ctx *ctx, const FftArgs &p) {
    const size_t lds = (size_t)B * R1 * (R2 + 1) * sizeof(cplx<T>);
    const long blocks = ((long)p.outer + B - 1) / B;
    hipLaunchKernelGGL((k_fft_contig<T, TIN, N, R1, R2, B>), dim3((unsigned)blocks), dim3(B * R2), lds, ctx->stream, p);
    return check_launch("fft_contig");
}

template <typename T, typename TIN, int N, int R1, int R2, int B>
int launch_strided_b(mmw_ctx *ctx, const FftArgs &p) {
    const size_t lds = (size_t)N * B * sizeof(cplx<T>);
    const long tiles = (p.inner + B - 1) / B;
    hipLaunchKernelGGL((k_fft_strided<T, TIN, N, R1, R2, B>), dim3((unsigned)(tiles * p.outer)), dim3(B * R2), lds,
                       ctx->stream, p);
    return check_launch("fft_strided");
}

template <typename T, typename TIN, int N, int R1, int R2>
int launch_pow2(mmw_ctx *ctx, const FftArgs &p, bool contiguous) {
    if (contiguous) {
        // (tile sizes 8/16/32 rows or columns measured within 3 % of each other on 256x128 planes)
        constexpr int B = contig_tile(R1, R2, (int)sizeof(cplx<T>));
        return launch_contig_b<T, TIN, N, R1, R2, B>(ctx, p);
    } else {
        constexpr int B = strided_tile(N, (int)sizeof(cplx<T>));
        // few columns in all (the Hann + FFT along range behind the Bartlett contraction: 64 columns x a handful of frames):
        // a quarter of the tile width puts four times as many workgroups on the chip (12 -> 6 us at 16 x 256 x 64)
        if constexpr (B >= 16 && sizeof(T) == 4 && sizeof(TIN) == 4) {
            const long tiles = (p.inner + B - 1) / B;
            if (tiles * p.outer * 2 <= ctx->num_cu && p.inner > B / 4) return launch_strided_b<T, TIN, N, R1, R2, B / 4>(ctx, p);
        }
        return launch_strided_b<T, TIN, N, R1, R2, B>(ctx, p);
    }
}

// contiguous == true: the axis is the fastest-varying index (in_axis_stride == 1, inner == 1,
// p.outer rows with in_outer_stride / out_outer_stride row pitches).
template <typename T, typename TIN>
int launch_fft_axis(mmw_ctx *ctx, FftArgs p, int N, bool contiguous) {
    MMW_REQUIRE(N >= 1 && p.n_in >= 0 && p.n_in <= N, "fft: bad N=%d n_in=%d", N, p.n_in);
    if (p.outer <= 0 || p.inner <= 0) return MMW_OK;
    MMW_TRY(get_table<T>(ctx, TAB_TWIDDLE, N, &p.tw));
    switch (N) {
        case 8: return launch_pow2<T, TIN, 8, 4, 2>(ctx, p, contiguous);
        case 16: return launch_pow2<T, TIN, 16, 4, 4>(ctx, p, contiguous);
        case 32: return launch_pow2<T, TIN, 32, 8, 4>(ctx, p, contiguous);
        case 64: return launch_pow2<T, TIN, 64, 8, 8>(ctx, p, contiguous);
        case 128: return launch_pow2<T, TIN, 128, 16, 8>(ctx, p, contiguous);
        case 256: return launch_pow2<T, TIN, 256, 16, 16>(ctx, p, contiguous);
        case 512: return launch_pow2<T, TIN, 512, 32, 16>(ctx, p, contiguous);
        case 1024: return launch_pow2<T, TIN, 1024, 32, 32>(ctx, p, contiguous);
        default: break;
    }
    MMW_REQUIRE(N <= 4096, "fft: axis length %d not supported (max 4096)", N);
    // direct DFT: view the contiguous case as inner = rows with a row-pitch stride
    if (contiguous) {
        p.inner = p.outer;
        p.in_inner_stride = p.in_outer_stride;
        p.out_inner_stride = p.out_outer_stride;
        p.outer = 1;
        p.in_outer_stride = p.out_outer_stride = 0;
        p.win_inner = nullptr;
    }
    int B = 16;
    while (B > 1 && (size_t)p.n_in * B * sizeof(cplx<T>) > 48 * 1024) B >>= 1;
    const size_t lds = (size_t)(p.n_in > 0 ? p.n_in : 1) * B * sizeof(cplx<T>);
    MMW_REQUIRE(lds <= 160 * 1024, "fft: axis length %d too large for the direct DFT tile", N);
    const long tiles = (p.inner + B - 1) / B;
    hipLaunchKernelGGL((k_dft_direct<T, TIN>), dim3((unsigned)(tiles * p.outer)), dim3(256), lds, ctx->stream, p, N, B);
    return check_launch("dft_direct");
}

}  // namespace mmw
