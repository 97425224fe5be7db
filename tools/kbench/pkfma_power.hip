// Kernel-development reproducer (NOT part of the product), round 5: the second attempt at the packed-f32 miscompute of round 4
// (profiles/r04_a_pkfma_findings.md).  What was known: the SLP-vectorised NAT decoder kernels (174 v_pk_fma_f32 ... op_sel_hi:[1,0,1] in
// nat_dec_proj_prenet_k) computed WRONG low halves whenever the **bf16** generator ran beside them — never beside the fp32 generator, a torch
// matmul or nothing — and tools/kbench/pkfma_hazard.hip (a v_pk chain beside an MFMA stream ON THE SAME SIMD, operands constant over the
// iterations) did not reproduce it.
//
// Hypothesis tested here (H3): the trigger is not instruction adjacency but the ELECTRICAL state of the chip.  The bf16 generator is the only
// neighbour that pins the package at its 1400 W cap (profiles/r03_f_power_clock.md: 1393-1395 W, shader clock pulled to 1.6-1.8 GHz; the fp32
// generator runs at 1.15 kW and the full 2.38 GHz), and the earlier microbenchmark's MFMA stream multiplied the SAME two registers for ever — low
// toggle rate, ~1 kW (profiles/r03_e_mfma_peak.txt: constant operands 1040 W, pseudo-random 1300 W+).  So:
//   * AGGRESSOR (stream A): a persistent kernel, two 4-wave workgroups per CU on every CU, streaming v_mfma_f32_32x32x16_bf16 on operands that CHANGE
//     every instruction (a rotating set of pseudo-random registers) — or constant ones, or fp32 MFMAs, or nothing;
//   * VICTIM (stream B): thousands of short launches shaped like the decoder's projection kernel (64 workgroups x 256 threads, operands from LDS,
//     four independent accumulator pairs) running either a v_pk_fma_f32 chain (both broadcast forms hipcc's SLP vectoriser emits) or the SAME
//     arithmetic as scalar v_fma_f32 — every launch has identical inputs, so every launch must reproduce launch 0 (taken on an idle chip) bit for bit;
//     the kernel itself compares and counts mismatching LOW / HIGH halves (and keeps the first few records).
// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/kbench/pkfma_power.hip -o tools/kbench/bin/pkfma_power
// run:   tools/kbench/bin/pkfma_power [aggressor_ms=1500] [victim_iters=400]
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

// mode 1: bf16 MFMAs, operands rotate through 4 pseudo-random register sets (high toggle rate); 2: bf16 MFMAs on constant operands;
// 3: fp32 MFMAs (v_mfma_f32_32x32x2_f32) on rotating operands.  Runs until *stop != 0 (the host sets it) or `max_iters`.
__global__ __launch_bounds__(256, 2) void aggressor_k(int mode, long max_iters, volatile int* stop, float* sink) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    unsigned s = 1234567u + threadIdx.x * 7919u + blockIdx.x * 104729u;
    bf16x8 a[4], b[4];
    float fa[4], fb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[q][e] = (__bf16)rnd(s);
            b[q][e] = (__bf16)rnd(s);
        }
        fa[q] = rnd(s);
        fb[q] = rnd(s);
    }
    for (long it = 0; it < max_iters; ++it) {
        if ((it & 255) == 0 && *stop) break;
        if (mode == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[(q + j) & 3]), "v"(b[q]));
        } else if (mode == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[0]), "v"(b[0]));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(fa[(q + j) & 3]), "v"(fb[q]));
        }
        // keep the accumulators finite: fold them back now and then (a handful of VALU instructions per 4096 MFMAs)
        if ((it & 255) == 255) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = acc[j][e] * 1e-6f;
        }
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[j][0] + acc[j][15];
    if (t == 123.456f) *sink = t;
}

// One victim launch: 64 workgroups x 256 threads.  LDS holds 4 KB of pseudo-random operands (the same in every launch); every lane runs `iters`
// rounds of 8 fused multiply-adds on 4 accumulator pairs — packed (v_pk_fma_f32 with the two broadcast forms) or scalar (v_fma_f32 x 2, the same
// arithmetic) — and compares its 8 results with `ref` (launch 0 writes `ref` instead).
struct Rec {
    unsigned launch, wg, lane, idx;
    float got, want;
};
__global__ __launch_bounds__(256) void victim_k(int packed, int iters, int launch, f32x2* ref, unsigned* counts, Rec* recs) {
    __shared__ float ops[1024];
    unsigned s = 987654321u + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        unsigned q = 77u + i * 2246822519u + blockIdx.x * 3266489917u;
        ops[i] = rnd(q) * 0.75f;
    }
    __syncthreads();
    f32x2 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x2{rnd(s), rnd(s)};
    const int base = (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
        const int o = (base + it * 8) & 1020;
        const f32x2 x = *reinterpret_cast<const f32x2*>(&ops[o]);                 // ds_read_b64
        const f32x2 w = *reinterpret_cast<const f32x2*>(&ops[(o + 514) & 1022]);
        if (packed) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "v"(x), "v"(w));  // both halves x w.x
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[j]) : "v"(x), "v"(w));     // both halves x w.y
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j].x = __builtin_fmaf(x.x, w.x, acc[j].x);
                acc[j].y = __builtin_fmaf(x.y, w.x, acc[j].y);
                acc[j].x = __builtin_fmaf(x.x, w.y, acc[j].x);
                acc[j].y = __builtin_fmaf(x.y, w.y, acc[j].y);
            }
        }
        // keep the values bounded: |x|, |w| < 0.75 => the sums drift; pull them back (scalar multiplies in both variants)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j].x *= 0.5f;
            acc[j].y *= 0.5f;
        }
    }
    const size_t slot = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (launch == 0) {
            ref[slot + j] = acc[j];
            continue;
        }
        const f32x2 r = ref[slot + j];
        const bool lo = __builtin_bit_cast(unsigned, acc[j].x) != __builtin_bit_cast(unsigned, r.x);
        const bool hi = __builtin_bit_cast(unsigned, acc[j].y) != __builtin_bit_cast(unsigned, r.y);
        if (lo) atomicAdd(&counts[0], 1u);
        if (hi) atomicAdd(&counts[1], 1u);
        if (lo || hi) {
            const unsigned k = atomicAdd(&counts[2], 1u);
            if (k < 16) recs[k] = Rec{(unsigned)launch, blockIdx.x, threadIdx.x, (unsigned)(2 * j + (lo ? 0 : 1)), lo ? acc[j].x : acc[j].y, lo ? r.x : r.y};
        }
    }
}

int main(int argc, char** argv) {
    const int agg_ms = argc > 1 ? atoi(argv[1]) : 1500;
    const int iters = argc > 2 ? atoi(argv[2]) : 400;
    int dev_cus = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    dev_cus = prop.multiProcessorCount;
    const int VWG = 64, N = VWG * 256 * 4;
    f32x2* ref[2];
    unsigned* counts;
    Rec* recs;
    float* sink;
    int* stop;
    CK(hipMalloc(&ref[0], (size_t)N * sizeof(f32x2)));
    CK(hipMalloc(&ref[1], (size_t)N * sizeof(f32x2)));
    CK(hipMalloc(&counts, 16));
    CK(hipMalloc(&recs, 16 * sizeof(Rec)));
    CK(hipMalloc(&sink, 4));
    CK(hipHostMalloc(&stop, 4, hipHostMallocMapped));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    // references on an idle chip
    for (int pk = 0; pk < 2; ++pk) {
        hipLaunchKernelGGL(victim_k, dim3(VWG), dim3(256), 0, sb, pk, iters, 0, ref[pk], counts, recs);
    }
    CK(hipStreamSynchronize(sb));
    {
        std::vector<f32x2> r0(N), r1(N);
        CK(hipMemcpy(r0.data(), ref[0], (size_t)N * sizeof(f32x2), hipMemcpyDeviceToHost));
        CK(hipMemcpy(r1.data(), ref[1], (size_t)N * sizeof(f32x2), hipMemcpyDeviceToHost));
        size_t d = 0;
        for (int i = 0; i < N; ++i) d += memcmp(&r0[i], &r1[i], sizeof(f32x2)) != 0;
        printf("idle chip: packed and scalar variants agree in %d of %d accumulator pairs (same arithmetic: v_pk_fma_f32 == 2 x v_fma_f32)\n", (int)(N - d), N);
    }
    const char* names[4] = {"nothing", "bf16 MFMAs, operands change every instruction (2 workgroups x 4 waves on every CU)", "bf16 MFMAs, constant operands",
                            "fp32 MFMAs, operands change every instruction"};
    printf("%d CUs; victim: 64 workgroups x 256 threads, %d rounds of 8 FMAs on 4 accumulator pairs from LDS operands; aggressor runs ~%d ms per case\n", dev_cus, iters, agg_ms);
    for (int mode = 0; mode < 4; ++mode)
        for (int pk = 1; pk >= 0; --pk) {
            CK(hipMemset(counts, 0, 16));
            *stop = 0;
            if (mode) hipLaunchKernelGGL(aggressor_k, dim3(dev_cus * 2), dim3(256), 0, sa, mode, 1l << 40, stop, sink);
            // give the aggressor ~100 ms to bring the package to its power state, then launch victims for agg_ms
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            struct timespec ts = {0, 100 * 1000 * 1000};
            nanosleep(&ts, nullptr);
            int launches = 0;
            CK(hipEventRecord(e0, sb));
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (;;) {
                for (int q = 0; q < 64; ++q) hipLaunchKernelGGL(victim_k, dim3(VWG), dim3(256), 0, sb, pk, iters, ++launches, ref[pk], counts, recs);
                CK(hipStreamSynchronize(sb));
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > agg_ms) break;
            }
            CK(hipEventRecord(e1, sb));
            CK(hipStreamSynchronize(sb));
            *stop = 1;
            if (mode) CK(hipStreamSynchronize(sa));
            unsigned c[4];
            CK(hipMemcpy(c, counts, 16, hipMemcpyDeviceToHost));
            printf("aggressor: %-86s | victim %-22s: %6d launches, %u LOW-half and %u HIGH-half values differ from the idle-chip launch\n", names[mode],
                   pk ? "v_pk_fma_f32 chain" : "scalar v_fma_f32 chain", launches, c[0], c[1]);
            if (c[2]) {
                Rec r[16];
                CK(hipMemcpy(r, recs, sizeof(r), hipMemcpyDeviceToHost));
                for (unsigned k = 0; k < (c[2] < 4 ? c[2] : 4); ++k)
                    printf("    launch %u wg %u lane %u value %u: got %.9g want %.9g\n", r[k].launch, r[k].wg, r[k].lane, r[k].idx, r[k].got, r[k].want);
            }
            fflush(stdout);
        }
    return 0;
}
