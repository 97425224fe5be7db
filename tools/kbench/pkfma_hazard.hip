// Kernel-development reproducer (NOT part of the product): does a wave's packed-f32 VALU arithmetic (v_pk_fma_f32 with op_sel broadcasts — what
// hipcc's SLP vectoriser makes of `a.x = fmaf(x.x, w, a.x); a.y = fmaf(x.y, w, a.y)`) stay correct while ANOTHER wave of the same SIMD streams
// matrix instructions?  Found in round 4: the NAT decoder's projection / prenet kernel (174 v_pk_fma_f32) produced wrong values for its EVEN
// sentences — the LOW halves of its packed register pairs — whenever the bf16 HiFi-GAN generator ran beside it on the chip, never beside the
// fp32 generator, a torch matmul or nothing, and never once nat.hip was built with -fno-slp-vectorize (tools/experiments/r04/diag_pipe3.py).
//
// One 512-thread workgroup per CU: waves 0-3 take role A (mode 0: idle, 1: v_mfma_f32_32x32x16_bf16 stream, 2: v_mfma_f32_32x32x2_f32 stream),
// waves 4-7 role B: a chain of v_pk_fma_f32 (both broadcast forms the compiler emits) next to the same chain in scalar v_fma_f32, compared bit
// for bit at the end.  Wave i and wave i + 4 share a SIMD.  Prints the number of lanes whose LOW / HIGH half differs from the scalar chain.
//   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/pkfma_hazard.hip -o tools/kbench/bin/pkfma_hazard
//   run:   tools/kbench/bin/pkfma_hazard [iters=20000]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                                   \
        }                                                                                              \
    } while (0)

__device__ __forceinline__ float rnd(unsigned& s) {  // xorshift -> [-1, 1)
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

__global__ __launch_bounds__(512) void hazard_k(int mode, int iters, unsigned* mism, float* sink) {
    extern __shared__ float pad[];  // 100 KB: one workgroup per CU
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (mode == 0) return;
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        unsigned s = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
        if (mode == 1) {
            bf16x8 a, b;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[e] = (__bf16)rnd(s);
                b[e] = (__bf16)rnd(s);
            }
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        } else {
            const float a = rnd(s), b = rnd(s);
            for (int it = 0; it < iters / 2; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        }
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[j][0] + acc[j][15];
        if (t == 123.456f) *sink = t;
        return;
    }
    // role B
    unsigned s = 987654321u + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    f32x2 pk = {rnd(s), rnd(s)};
    float s0 = pk.x, s1 = pk.y;
    for (int it = 0; it < iters; ++it) {
        const f32x2 x = {rnd(s), rnd(s)};
        const f32x2 w = {rnd(s) * 0.5f, rnd(s) * 0.5f};
        // pk.xy = x.xy * w.xx + pk.xy      (low half of src1 for both lanes)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pk) : "v"(x), "v"(w));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x.x), "v"(w.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x.y), "v"(w.x));
        // pk.xy = x.yx?  no: x.xy * w.yy + pk.xy   (high half of src1 for both lanes)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(pk) : "v"(x), "v"(w));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x.x), "v"(w.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x.y), "v"(w.y));
        // keep the chain bounded
        pk.x *= 0.75f;
        pk.y *= 0.75f;
        s0 *= 0.75f;
        s1 *= 0.75f;
    }
    if (__builtin_bit_cast(unsigned, pk.x) != __builtin_bit_cast(unsigned, s0)) atomicAdd(&mism[0], 1u);
    if (__builtin_bit_cast(unsigned, pk.y) != __builtin_bit_cast(unsigned, s1)) atomicAdd(&mism[1], 1u);
    if (pk.x + pk.y + s0 + s1 == 123.456f) *sink = pk.x;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned* mism;
    float* sink;
    CK(hipMalloc(&mism, 8));
    CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hazard_k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    const char* names[3] = {"role A idle", "role A streams v_mfma_f32_32x32x16_bf16", "role A streams v_mfma_f32_32x32x2_f32"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(mism, 0, 8));
            hipLaunchKernelGGL(hazard_k, dim3(512), dim3(512), 100 * 1024, 0, mode, iters, mism, sink);
            CK(hipDeviceSynchronize());
            unsigned h[2];
            CK(hipMemcpy(h, mism, 8, hipMemcpyDeviceToHost));
            printf("%-44s: %d x 2 v_pk_fma_f32 per lane, %d role-B lanes: LOW halves wrong in %u lanes, HIGH halves wrong in %u lanes\n", names[mode], iters,
                   512 * 256, h[0], h[1]);
        }
    return 0;
}
