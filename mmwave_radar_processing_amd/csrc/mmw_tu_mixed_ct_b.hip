// Translation unit: compile-time specialised mixed-radix RD kernels, second half of the shipped cfg shapes.
#define MMW_TU_MIXED_CT_B
#include "mmw_fft_mixed_ct.h"
