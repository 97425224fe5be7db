// Device helpers shared by the bf16 kernels (gfx950).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vtts_internal.h"

namespace vtts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);  // v_cvt_pk_bf16_f32 (round-to-nearest-even)
    return __builtin_bit_cast(unsigned, b);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ float lrelu_f(float v, float s) { return v >= 0.0f ? v : v * s; }
__device__ __forceinline__ unsigned lrelu_bf16x2(unsigned u, float s) {
    return pack_bf16x2(lrelu_f(bf16_lo(u), s), lrelu_f(bf16_hi(u), s));
}


// 16-byte-slot XOR swizzle of a channels-last LDS tile with SPR slots per row: consecutive time rows
// (the 32 lanes of an MFMA B fragment read) land on distinct slots of the 256-byte bank row.
template <int SPR>
__device__ __forceinline__ int swz_of(int row) {
    constexpr int RPB = SPR >= 16 ? 1 : 16 / SPR;
    constexpr int MASK = (SPR >= 16 ? 16 : SPR) - 1;
    return (row / RPB) & MASK;
}

}  // namespace vtts
