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
// Plain multiply / add.  No v_pk_*_f32 may appear in these kernels (see lrelu01_pack below): the bf16 kernel files are compiled with
// -fno-slp-vectorize (csrc/build.py) so that hipcc does not pair scalar f32 operations into packed ones.  NOT inline asm: a VALU
// instruction hidden in an asm statement that reads an MFMA result gets none of the wait states the hazard recogniser inserts
// between an MFMA and a VALU read of its destination (gfx950 does not interlock that) — the first cut of this change read stale
// accumulators in epilogue 1 (tools/kbench check: err/tol 1.27 .. 26).
__device__ __forceinline__ float vmul_raw(float a, float b) { return a * b; }
__device__ __forceinline__ float vadd_raw(float a, float b) { return a + b; }
// MRF mean (model.py:121 `x = xs / num_kernels`) in the bf16 kernels: a multiplication by the reciprocal, formed once per thread.  The fp32
// engine divides (it answers to the 1e-4 bar); here the value is rounded to bf16 right afterwards, a product and a quotient differ by at
// most one fp32 ulp (2^-16 of a bf16 ulp), and an IEEE division is 8 VALU instructions per element (v_div_scale x 2, v_rcp, 3 fma,
// v_div_fmas, v_div_fixup: ~1000 per tile and wave on the four launches per pass that end a stage — they ran 10-13 % behind their siblings).
#ifndef VTTS_MRF_DIV  // A/B switch: 1 = round 2's IEEE division per element
#define VTTS_MRF_DIV 0
#endif
__device__ __forceinline__ float mrf_recip(float div) { return 1.0f / div; }
// LeakyReLU with a slope in (0, 1] (the model's: 0.1, 0.01, model.py:5,122; the engine rejects others): max(v, s v) — the same value as the
// compare-and-select form for every finite v (v >= 0: s v <= v; v < 0: s v > v), two VALU instructions instead of three
__device__ __forceinline__ float lrelu_f(float v, float s) {
    float r;
    const float m = vmul_raw(v, s);
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
    return r;
}
// slope 0.1 < 1: leaky_relu(v) = max(v, 0.1 v) — identical values (v >= 0: v >= 0.1 v; v < 0: 0.1 v > v), one op fewer
// max(a, b) as ONE v_max_f32: fmaxf() makes hipcc canonicalise (v_max_f32 x, x, x) every operand it cannot prove is not a
// signalling NaN — one extra VALU instruction per element of the staging and epilogue phases, and those share the SIMD's
// issue port with the co-resident workgroup's MFMAs (profiles/r01_j_kbench_findings.md).  Same value for non-NaN inputs.
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lrelu01(float v) { return vmax_raw(v, vmul_raw(v, 0.1f)); }
// LeakyReLU(0.1) of two values rounded to a bf16 pair: two multiplies, two maxes, one convert.
// NOT v_pk_mul_f32 (round 2: "one op fewer"): the packed-f32 VALU instructions (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32) do not
// issue at all while the OTHER wave of the SIMD streams MFMAs — overlap 0.01 against 0.8 for every plain VALU instruction
// (tools/kbench/coissue.hip, profiles/r03_a_coissue_findings.md) — and the staging / epilogue phases that use this run beside the
// co-resident workgroup's MFMA loops by design.  The file is built with -fno-slp-vectorize so that hipcc cannot re-pack them.
#ifndef VTTS_PK_F32  // A/B switch: 1 = round 2's packed multiply
#define VTTS_PK_F32 0
#endif
__device__ __forceinline__ unsigned lrelu01_pack(float lo, float hi) {
#if VTTS_PK_F32
    const f32x2 v = {lo, hi};
    const f32x2 m = v * 0.1f;  // v_pk_mul_f32
    return pack_bf16x2(vmax_raw(v.x, m.x), vmax_raw(v.y, m.y));
#else
    const float k = 0.1f;
    return pack_bf16x2(vmax_raw(lo, vmul_raw(lo, k)), vmax_raw(hi, vmul_raw(hi, k)));
#endif
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

// Byte offset of 16-byte slot `slot` of row `row` in a channels-last LDS tile of SPR slots (8 bf16 channels each) per row.
//   SPR >= 16 (C >= 128): row-major, the slot XOR-swizzled by the row (above): B-fragment reads (ds_read_b128: 16-lane groups
//     {0-3,12-15,20-27}, {4-11,16-19,28-31}, banks (a/4) mod 64), epilogue writes (ds_write_b128: 8 consecutive lanes = 8
//     consecutive rows, banks (a/4) mod 32) and staging writes are all conflict-free (tools/kbench/lds_conflicts.py).
//   SPR = 4, 8 (C = 32, 64; 64- / 128-byte rows): NO function f makes  row * pitch + ((slot ^ f(row)) << 4)  conflict-free for the
//     reads and the epilogue writes at once (the writes need f(r), f(r+2), f(r+4), f(r+6) distinct, hence f = pi((r >> 1) & 3),
//     which collides on a read group's rows r, r+12, r+20, r+24); round 2's choice served the reads and left every epilogue write
//     2-way conflicted (8.5 M conflict cycles per C = 32 / 64 pair launch, 44-46 M per whole-ResBlock launch:
//     profiles/r02_b_pmc_bf16.md).  These tiles are stored in blocks of 16 rows, slot-major inside a block: a block's 16 rows
//     of one slot are one 256-byte LDS line, so any 16 rows that are distinct mod 16 (a read group) and any 8 consecutive rows
//     (a write group) hit distinct banks.  Tile rows are allocated in multiples of 16.
#ifndef VTTS_TILE_BLOCKED  // A/B switch (tools/ab_bench.sh): 0 = round 2's XOR swizzle at every pitch
#define VTTS_TILE_BLOCKED 1
#endif
template <int SPR>
__device__ __forceinline__ int tile_off(int row, int slot) {
    if constexpr (SPR >= 16 || !VTTS_TILE_BLOCKED) return row * (SPR * 16) + ((slot ^ swz_of<SPR>(row)) << 4);
    else return (row >> 4) * (SPR * 256) + (slot << 8) + ((row & 15) << 4);
}
constexpr int tile_rows16(int rows) { return (rows + 15) / 16 * 16; }

}  // namespace vtts
