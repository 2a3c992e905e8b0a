// Steering-matrix beamformers (Bartlett / Capon) -- MFMA kernels.
#pragma once
#include "mmw_ctx.h"
#include "mmw_fft_generic.h"

namespace mmw {

inline int bartlett(mmw_ctx *, const void *, const double *, const double *, void *, int, int, int, double) {
    return set_error(MMW_ERR_UNSUPPORTED, "bartlett: not built yet");
}
inline int capon(mmw_ctx *, const void *, const double *, float *, int, int, int, int, double) {
    return set_error(MMW_ERR_UNSUPPORTED, "capon: not built yet");
}

}  // namespace mmw
