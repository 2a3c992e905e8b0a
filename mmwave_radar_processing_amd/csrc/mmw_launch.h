// Explicit-instantiation declarations: the heavy launchers (and the kernels behind them) are compiled once each in
// their own translation unit (mmw_tu_*.hip) so the library builds in parallel; every other unit only links to them.
#pragma once
#include "mmw_fft_generic.h"
#include "mmw_fft_fused.h"
#include "mmw_fft_mixed.h"

namespace mmw {

#define MMW_FFT_AXIS_INSTANCES(X) X(float, float) X(double, float) X(double, double)
#define MMW_ANGLE_V_INSTANCES(X) X(4) X(8) X(12) X(16)

#ifndef MMW_TU_GENERIC
#define X(T, TIN) extern template int launch_fft_axis<T, TIN>(mmw_ctx *, FftArgs, int, bool);
MMW_FFT_AXIS_INSTANCES(X)
#undef X
#endif

#ifndef MMW_TU_ANGLE
#define X(V)                                                                                                        \
    extern template int launch_angle64<V>(mmw_ctx *, const void *, void *, int, long, bool, const float *, bool);  \
    extern template int launch_angle64_sync<V>(mmw_ctx *, const void *, void *, long, bool, const float *, bool, ChainSync, int); \
    extern template int launch_angle64_rmean<V>(mmw_ctx *, const void *, float *, size_t, float *, int, int, int, int, int, \
                                                const float *, bool);
MMW_ANGLE_V_INSTANCES(X)
#undef X
#endif

#ifndef MMW_TU_MIXED_F32
extern template int launch_rd_mixed<float, false>(mmw_ctx *, const void *, long, void *, int, int, int, RawView);
#endif
#ifndef MMW_TU_MIXED_F64
extern template int launch_rd_mixed<double, true>(mmw_ctx *, const void *, long, void *, int, int, int, RawView);
#endif

}  // namespace mmw
