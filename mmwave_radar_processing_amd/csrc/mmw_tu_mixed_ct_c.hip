// Translation unit: split (register-carried radix-2) range-Doppler kernels for planes of 2 x 16384 cells.
#define MMW_TU_MIXED_CT_C
#include "mmw_fft_split_ct.h"
