// Kernel-development reproducer (NOT part of the product), round 5: single-instruction-type AGGRESSORS for the packed-f32 miscompute.  The victim is
// the failing artefact itself (the acoustic model of a library variant with v_pk_fma_f32 written out in nat_dec_proj_prenet_k:
// tools/experiments/r05/pkfma_bisect.py); this library supplies what runs BESIDE it on the other stream, one instruction class at a time, on every CU
// (two 4-wave workgroups per CU, ~`iters` rounds per launch):
//   1  v_mfma_f32_32x32x16_bf16 on register operands (what tools/kbench/pkfma_hazard.hip already showed harmless)
//   2  v_permlane32_swap
//   3  v_cvt_pk_bf16_f32
//   4  ds_read_b128 / ds_write_b128
//   5  v_mfma_f32_32x32x16_bf16 fed by ds_read_b128 every step (operands from LDS, as the generators' loops)
//   6  everything above interleaved
//   7  v_mfma_f32_32x32x2_f32 (the fp32 generator's instruction)
//   8  plain VALU (v_fma_f32 / v_max_f32 / v_mul_f32)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -c pk_aggressor.hip -o x.o && clang++ -shared -fPIC x.o -o libpk_aggressor.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void agg_k(int kind, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256 * 4 * 4];
    unsigned s = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    auto rnd = [&]() {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
    };
    for (int i = threadIdx.x; i < 256 * 16; i += 256) lds[i] = __builtin_bit_cast(unsigned, rnd()) & 0x3f803f80u;
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)rnd();
        b[e] = (__bf16)rnd();
    }
    float f0 = rnd(), f1 = rnd(), f2 = rnd(), f3 = rnd();
    unsigned u0 = s, u1 = s * 3u, p0 = 0, p1 = 0;
    volatile u32x4* my = reinterpret_cast<volatile u32x4*>(lds) + threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
        if (kind == 1 || kind == 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
        if (kind == 2 || kind == 6) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u0), "+v"(u1));
        }
        if (kind == 3 || kind == 6) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p0) : "v"(f0), "v"(f1));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(f2), "v"(f3));
                f0 = __builtin_bit_cast(float, (p0 << 16) | 0x3f000000u);
                f2 = __builtin_bit_cast(float, (p1 & 0xffff0000u) | 0x3f000000u);
            }
        }
        if (kind == 4 || kind == 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 v = my[j];
                v.x ^= 1u;
                my[(j + 1) & 3] = v;
            }
        }
        if (kind == 5) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 v = my[j];
                const bf16x8 bb = __builtin_bit_cast(bf16x8, v);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[j], 0, 0, 0);
            }
        }
        if (kind == 7) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(f0), "v"(f1));
        }
        if (kind == 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f0) : "v"(f1), "v"(f2));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(f3) : "v"(f0));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
            }
        }
        if ((it & 255) == 255) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] *= 1e-6f;
        }
    }
    float t = f0 + f2 + f3 + (float)(u0 ^ u1) + (float)(p0 + p1);
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[j][0] + acc[j][15];
    if (t == 123.456f) *sink = t + (float)lds[threadIdx.x];
}

extern "C" __attribute__((visibility("default"))) int pk_aggressor_launch(void* stream, int kind, int wgs, int iters, float* sink) {
    hipLaunchKernelGGL(agg_k, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), kind, iters, sink);
    return (int)hipGetLastError();
}
