// float32 matrix products on the bfloat16 matrix cores: exact three-way operand splits (used by the Bartlett contraction,
// mmw_beamform.h, and by the 127-point DFT level of the mixed-radix range-Doppler kernels, mmw_fft_mixed_ct.h).
// float32 MFMAs never co-execute with vector instructions on this chip, bfloat16 MFMAs do (profiles/r04_coexec.json).
#pragma once
#include <hip/hip_runtime.h>

namespace mmw {

typedef float v16f __attribute__((ext_vector_type(16)));

// Exact three-way split of eight float32 values into bfloat16 fragments (element j of each fragment = piece of v[j]):
// v = p1 + p2 + p3 with p1 = v truncated to 8 significant bits, p2 = (v - p1) truncated, p3 = the rest (<= 8 bits: exact).
// Truncation (a mask) instead of rounding keeps every remainder exactly representable in float32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16x3(const float (&v)[8], bf16x8 &f1, bf16x8 &f2, bf16x8 &f3) {
    u32x4 p1, p2, p3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = v[2 * q], b = v[2 * q + 1];
        const unsigned a1 = __builtin_bit_cast(unsigned, a) & 0xffff0000u, b1 = __builtin_bit_cast(unsigned, b) & 0xffff0000u;
        const float ra = a - __builtin_bit_cast(float, a1), rb = b - __builtin_bit_cast(float, b1);
        const unsigned a2 = __builtin_bit_cast(unsigned, ra) & 0xffff0000u, b2 = __builtin_bit_cast(unsigned, rb) & 0xffff0000u;
        const float sa = ra - __builtin_bit_cast(float, a2), sb = rb - __builtin_bit_cast(float, b2);
        const unsigned a3 = __builtin_bit_cast(unsigned, sa), b3 = __builtin_bit_cast(unsigned, sb);
        p1[q] = (a1 >> 16) | b1;            // element 2 q in the low half, 2 q + 1 in the high half
        p2[q] = (a2 >> 16) | b2;
        p3[q] = (a3 >> 16) | (b3 & 0xffff0000u);
    }
    f1 = __builtin_bit_cast(bf16x8, p1);
    f2 = __builtin_bit_cast(bf16x8, p2);
    f3 = __builtin_bit_cast(bf16x8, p3);
}
// acc += A B over 16 values of k, A and B given as three-way splits: the six products down to 2^-24 of |a||b|
// (a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1; what is dropped -- a2 b3, a3 b2, a3 b3 -- is <= 3 * 2^-24 |a||b|)
__device__ __forceinline__ v16f mfma_bf16x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], v16f acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// the same on 16 x 16 tiles over 32 values of k (v_mfma_f32_16x16x32_bf16: lane l holds A[row l & 15][k = 8 (l >> 4) + j],
// B[k = 8 (l >> 4) + j][col l & 15]; C/D: col = l & 15, row = 4 (l >> 4) + reg)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mfma16_bf16x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], v4f acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// the leading piece alone (element j = v[j] truncated to 8 significant bits)
__device__ __forceinline__ bf16x8 top_bf16(const float (&v)[8]) {
    u32x4 p1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        p1[q] = (__builtin_bit_cast(unsigned, v[2 * q]) >> 16) | (__builtin_bit_cast(unsigned, v[2 * q + 1]) & 0xffff0000u);
    return __builtin_bit_cast(bf16x8, p1);
}

}  // namespace mmw
