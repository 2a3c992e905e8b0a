// Translation unit: k_rd_mixed, float64 magnitude (CFAR plane) variants.
#define MMW_TU_MIXED_F64
#include "mmw_launch.h"
namespace mmw {
template int launch_rd_mixed<double, true>(mmw_ctx *, const void *, long, void *, int, int, int, RawView);
}  // namespace mmw
