// Kernel-development reproducer (NOT part of the product): does a wave's packed-f32 VALU arithmetic (v_pk_fma_f32 with op_sel broadcasts — what
// hipcc's SLP vectoriser makes of `a.x = fmaf(x.x, w, a.x); a.y = fmaf(x.y, w, a.y)`) stay correct while ANOTHER wave of the same SIMD streams
// matrix instructions?  Found in round 4: the NAT decoder's projection / prenet kernel (174 v_pk_fma_f32) produced wrong values for its EVEN
// sentences — the LOW halves of its packed register pairs — whenever the bf16 HiFi-GAN generator ran beside it on the chip, never beside the
// fp32 generator, a torch matmul or nothing, and never once nat.hip was built with -fno-slp-vectorize (tools/experiments/r04/diag_pipe3.py).
//
// One 512-thread workgroup per CU: waves 0-3 take role A (mode 0: idle, 1: v_mfma_f32_32x32x16_bf16 stream, 2: v_mfma_f32_32x32x2_f32 stream),
// waves 4-7 role B: a chain of v_pk_fma_f32 (both broadcast forms the compiler emits), v_pk_add_f32 and v_pk_mul_f32 on operands read from LDS;
// a lane's final pair is compared, bit for bit, with what the same chain gave beside idle role-A waves.  Wave i and wave i + 4 share a SIMD.
// RESULT (gpurun_out/r04_run7): see profiles/r04_a_pkfma_findings.md.
//   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/pkfma_hazard.hip -o tools/kbench/bin/pkfma_hazard
//   run:   tools/kbench/bin/pkfma_hazard [iters=20000]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

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

__global__ __launch_bounds__(512) void hazard_k(int mode, int iters, f32x2* out, float* sink) {
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
    // role B: a chain of packed-f32 arithmetic on operands that come from LDS (as in the projection kernel); the lane's final pair goes to
    // `out`, and the host compares the runs beside an MFMA stream with the run beside idle waves
    unsigned s = 987654321u + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float* my = pad + (threadIdx.x - 256) * 8;
    f32x2 pk = {rnd(s), rnd(s)};
    for (int it = 0; it < iters; ++it) {
        my[0] = rnd(s); my[1] = rnd(s); my[2] = rnd(s) * 0.5f; my[3] = rnd(s) * 0.5f;
        const f32x2 x = *reinterpret_cast<volatile f32x2*>(my);
        const f32x2 w = *reinterpret_cast<volatile f32x2*>(my + 2);
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(pk) : "v"(x), "v"(w));  // both lanes take src1's first register
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(pk) : "v"(x), "v"(w));     // both lanes take src1's second register
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk) : "v"(w));
        const f32x2 k = {0.5f, 0.5f};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk) : "v"(k));
    }
    out[(size_t)blockIdx.x * 256 + (threadIdx.x - 256)] = pk;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int WGS = 512, N = WGS * 256;
    f32x2* out;
    float* sink;
    CK(hipMalloc(&out, (size_t)N * sizeof(f32x2)));
    CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hazard_k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    const char* names[3] = {"role A idle", "role A streams v_mfma_f32_32x32x16_bf16", "role A streams v_mfma_f32_32x32x2_f32"};
    std::vector<f32x2> ref(N), got(N);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(out, 0, (size_t)N * sizeof(f32x2)));
            hipLaunchKernelGGL(hazard_k, dim3(WGS), dim3(512), 100 * 1024, 0, mode, iters, out, sink);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), out, (size_t)N * sizeof(f32x2), hipMemcpyDeviceToHost));
            if (rep == 0 && mode == 0) ref = got;
            unsigned lo = 0, hi = 0;
            for (int i = 0; i < N; ++i) {
                const float g0 = got[i].x, g1 = got[i].y, r0 = ref[i].x, r1 = ref[i].y;
                lo += memcmp(&g0, &r0, 4) != 0;
                hi += memcmp(&g1, &r1, 4) != 0;
            }
            printf("%-44s: %d x (2 v_pk_fma_f32 + v_pk_add_f32 + v_pk_mul_f32) per lane, %d role-B lanes: vs the first idle run LOW halves differ in %u lanes, HIGH halves in %u\n",
                   names[mode], iters, N, lo, hi);
        }
    return 0;
}
