// Translation unit: k_angle64 and k_angle64_rmean for V = 4, 8, 12, 16 antennas.
#define MMW_TU_ANGLE
#include "mmw_launch.h"
namespace mmw {
#define X(V)                                                                                                \
    template int launch_angle64<V>(mmw_ctx *, const void *, void *, int, long, bool, const float *, bool); \
    template int launch_angle64_sync<V>(mmw_ctx *, const void *, void *, long, bool, const float *, bool, ChainSync, int); \
    template int launch_angle64_rmean<V>(mmw_ctx *, const void *, float *, size_t, float *, int, int, int, int, int, \
                                         const float *, bool);
MMW_ANGLE_V_INSTANCES(X)
#undef X
}  // namespace mmw
