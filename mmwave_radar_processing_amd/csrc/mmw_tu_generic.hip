// Translation unit: the any-shape FFT kernels (k_fft_strided / k_fft_contig / k_dft_direct) in the three precisions used.
#define MMW_TU_GENERIC
#include "mmw_launch.h"
namespace mmw {
#define X(T, TIN) template int launch_fft_axis<T, TIN>(mmw_ctx *, FftArgs, int, bool);
MMW_FFT_AXIS_INSTANCES(X)
#undef X
}  // namespace mmw
