// Kernel-development reproducer (NOT part of the product), round 5.  The round-4 miscompute bisected on the failing artefact
// (tools/experiments/r05/pkfma_bisect.sh: product build 0 of 4 trials wrong; v_pk_fma_f32 written out in nat_dec_proj_prenet_k's `partial` loop
// ALONE — any one of its three phases, no SLP anywhere — 4 of 4 wrong, even rows = LOW halves only) is carried into a self-checking victim:
// the `partial` loop of viettts_amd/csrc/nat.hip computed TWICE by every thread from the same operands — once as v_pk_fma_f32 with the src1
// broadcasts, once as four scalar v_fma_f32 chains (bit-identical arithmetic) — and compared bit for bit inside the kernel.  The aggressor is
// whatever the host runs beside it (tools/experiments/r05/pkfma_victim.py: the real bf16 / fp32 / bf16x3 generators, or nothing).
//   variant bit 0: weights by global_load_dwordx4 inside the loop (as nat.hip) instead of from LDS
//   variant bit 1: 1024-thread workgroups (as nat.hip) instead of 256
//   variant bit 2: no LDS operand reads at all in the loop (x from registers)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC tools/kbench/pkfma_victim.hip -o tools/kbench/bin/libpkfma_victim.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void victim_k(int variant, int iters, int rows, const float4* __restrict__ wg, unsigned* counts) {
    extern __shared__ float4 lds4[];  // [rows] x values (4 "sentences" per row) + [rows / 4 * THREADS?]: weights when not from global
    float4* xs = lds4;
    const int g = threadIdx.x;
    unsigned s = 2463534242u + (blockIdx.x * THREADS + g) * 2654435761u;
    auto rnd = [&]() {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
    };
    for (int k = g; k < rows; k += THREADS) xs[k] = make_float4(rnd(), rnd(), rnd(), rnd());
    __syncthreads();
    const bool w_global = variant & 1, x_regs = variant & 4;
    unsigned lo = 0, hi = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        float4 xr = make_float4(rnd(), rnd(), rnd(), rnd());
#pragma unroll 4
        for (int k = 0; k < rows; k += 4) {
            float4 wv;
            if (w_global) wv = wg[(size_t)(k >> 2) * THREADS + g];                         // one 16-byte load per lane = 4 rows of its column
            else wv = make_float4(xs[k].y, xs[(k + 1) % rows].z, xs[(k + 2) % rows].w, xs[(k + 3) % rows].x);
            float4 x0, x1, x2, x3;
            if (x_regs) {
                x0 = xr; x1 = make_float4(xr.y, xr.z, xr.w, xr.x); x2 = make_float4(xr.z, xr.w, xr.x, xr.y); x3 = make_float4(xr.w, xr.x, xr.y, xr.z);
                xr.x += 0.125f;
            } else {
                x0 = xs[k]; x1 = xs[k + 1]; x2 = xs[k + 2]; x3 = xs[k + 3];                // ds_read_b128 x 4
            }
            a01 = __builtin_elementwise_fma(f32x2_t{x0.x, x0.y}, f32x2_t{wv.x, wv.x}, a01); a23 = __builtin_elementwise_fma(f32x2_t{x0.z, x0.w}, f32x2_t{wv.x, wv.x}, a23);
            a01 = __builtin_elementwise_fma(f32x2_t{x1.x, x1.y}, f32x2_t{wv.y, wv.y}, a01); a23 = __builtin_elementwise_fma(f32x2_t{x1.z, x1.w}, f32x2_t{wv.y, wv.y}, a23);
            a01 = __builtin_elementwise_fma(f32x2_t{x2.x, x2.y}, f32x2_t{wv.z, wv.z}, a01); a23 = __builtin_elementwise_fma(f32x2_t{x2.z, x2.w}, f32x2_t{wv.z, wv.z}, a23);
            a01 = __builtin_elementwise_fma(f32x2_t{x3.x, x3.y}, f32x2_t{wv.w, wv.w}, a01); a23 = __builtin_elementwise_fma(f32x2_t{x3.z, x3.w}, f32x2_t{wv.w, wv.w}, a23);
            s0 = __builtin_fmaf(x0.x, wv.x, s0); s1 = __builtin_fmaf(x0.y, wv.x, s1); s2 = __builtin_fmaf(x0.z, wv.x, s2); s3 = __builtin_fmaf(x0.w, wv.x, s3);
            s0 = __builtin_fmaf(x1.x, wv.y, s0); s1 = __builtin_fmaf(x1.y, wv.y, s1); s2 = __builtin_fmaf(x1.z, wv.y, s2); s3 = __builtin_fmaf(x1.w, wv.y, s3);
            s0 = __builtin_fmaf(x2.x, wv.z, s0); s1 = __builtin_fmaf(x2.y, wv.z, s1); s2 = __builtin_fmaf(x2.z, wv.z, s2); s3 = __builtin_fmaf(x2.w, wv.z, s3);
            s0 = __builtin_fmaf(x3.x, wv.w, s0); s1 = __builtin_fmaf(x3.y, wv.w, s1); s2 = __builtin_fmaf(x3.z, wv.w, s2); s3 = __builtin_fmaf(x3.w, wv.w, s3);
        }
        lo += (__builtin_bit_cast(unsigned, a01.x) != __builtin_bit_cast(unsigned, s0)) + (__builtin_bit_cast(unsigned, a23.x) != __builtin_bit_cast(unsigned, s2));
        hi += (__builtin_bit_cast(unsigned, a01.y) != __builtin_bit_cast(unsigned, s1)) + (__builtin_bit_cast(unsigned, a23.y) != __builtin_bit_cast(unsigned, s3));
    }
    if (lo) atomicAdd(&counts[0], lo);
    if (hi) atomicAdd(&counts[1], hi);
    if (g == 0) atomicAdd(&counts[2], 1u);  // workgroups that ran
}

extern "C" __attribute__((visibility("default"))) int pkfma_victim_launch(void* stream, int variant, int wgs, int iters, int rows, const void* weights, unsigned* counts) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)rows * 16;
    if (variant & 2) hipLaunchKernelGGL(victim_k<1024>, dim3(wgs), dim3(1024), lds, s, variant, iters, rows, static_cast<const float4*>(weights), counts);
    else hipLaunchKernelGGL(victim_k<256>, dim3(wgs), dim3(256), lds, s, variant, iters, rows, static_cast<const float4*>(weights), counts);
    return (int)hipGetLastError();
}
