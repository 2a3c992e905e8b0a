// Translation unit: the single-pass power-of-two range-Doppler kernels (k_rd_fused_256x128*, k_rd_lds<S, C>).
#define MMW_TU_RD
#include "mmw_launch.h"
