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
// slope 0.1 < 1: leaky_relu(v) = max(v, 0.1 v) — identical values (v >= 0: v >= 0.1 v; v < 0: 0.1 v > v), one op fewer
// max(a, b) as ONE v_max_f32: fmaxf() makes hipcc canonicalise (v_max_f32 x, x, x) every operand it cannot prove is not a
// signalling NaN — one extra VALU instruction per element of the staging and epilogue phases, and those share the SIMD's
// issue port with the co-resident workgroup's MFMAs (profiles/r01_j_kbench_findings.md).  Same value for non-NaN inputs.
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lrelu01(float v) { return vmax_raw(v, v * 0.1f); }
// LeakyReLU(0.1) of two values rounded to a bf16 pair: one packed multiply, two maxes, one convert
__device__ __forceinline__ unsigned lrelu01_pack(float lo, float hi) {
    const f32x2 v = {lo, hi};
    const f32x2 m = v * 0.1f;  // v_pk_mul_f32
    return pack_bf16x2(vmax_raw(v.x, m.x), vmax_raw(v.y, m.y));
}
__device__ __forceinline__ unsigned lrelu_bf16x2(unsigned u, float s) {
    return pack_bf16x2(lrelu_f(bf16_lo(u), s), lrelu_f(bf16_hi(u), s));
}


// 16-byte-slot XOR swizzle of a channels-last LDS tile with SPR slots per row: consecutive time rows
// (the 32 lanes of an MFMA B fragment read) land on distinct slots of the 256-byte bank row.
// kernel-development builds (-DVTTS_TIMELINE=1, tools/kbench): thread 0 of every workgroup stamps the
// shader clock at phase boundaries into BConvArgs::dbg[wg * 16 + i]; slot 15 = (XCC_ID << 32) | HW_ID
#ifndef VTTS_TIMELINE
#define VTTS_TIMELINE 0
#endif
#if VTTS_TIMELINE
#define VTTS_TL(a, wg, i)                                                                  \
    do {                                                                                   \
        if ((a).dbg && threadIdx.x == 0) (a).dbg[(size_t)(wg) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define VTTS_TL_ID(a, wg)                                                                  \
    do {                                                                                   \
        if ((a).dbg && threadIdx.x == 0)                                                   \
            (a).dbg[(size_t)(wg) * 16 + 15] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | \
                                              (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); \
    } while (0)
#else
#define VTTS_TL(a, wg, i) do { } while (0)
#define VTTS_TL_ID(a, wg) do { } while (0)
#endif

// 16-byte-per-lane LDS-DMA that hipcc does not see (MI355X guide §5.7): a wave that issues
// __builtin_amdgcn_global_load_lds gets every later ds_read wait as s_waitcnt lgkmcnt(0) instead of a counted one
// (measured on the fragment pipeline of kernels_bf16_rb.hip).  lds_dst = wave-uniform LDS byte address; lane i's 16
// bytes land at lds_dst + 16*i.  Completion is the caller's business: a counted s_waitcnt vmcnt, then a barrier.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

template <int SPR>
__device__ __forceinline__ int swz_of(int row) {
    constexpr int RPB = SPR >= 16 ? 1 : 16 / SPR;
    constexpr int MASK = (SPR >= 16 ? 16 : SPR) - 1;
    return (row / RPB) & MASK;
}

}  // namespace vtts
