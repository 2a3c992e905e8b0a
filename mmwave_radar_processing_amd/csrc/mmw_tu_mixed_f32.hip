// Translation unit: k_rd_mixed, complex64 spectrum variants.
#define MMW_TU_MIXED_F32
#include "mmw_launch.h"
namespace mmw {
template int launch_rd_mixed<float, false>(mmw_ctx *, const void *, long, void *, int, int, int, RawView);
}  // namespace mmw
