// Translation unit: compile-time specialised mixed-radix RD kernels, first half of the shipped cfg shapes (+ dispatch).
#define MMW_TU_MIXED_CT_A
#include "mmw_fft_mixed_ct.h"
