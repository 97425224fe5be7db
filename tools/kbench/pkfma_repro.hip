// Kernel-development reproducer (NOT part of the product), round 5: the round-4 "packed-f32 miscompute" reduced to ONE instruction pattern.
//
// MECHANISM (profiles/r05_b_pkfma_findings.md).  On gfx950 a VOP3P packed-f32 instruction (v_pk_fma_f32 ...) that is the first reader of a VGPR
// written by a VECTOR-MEMORY LOAD, issued right behind the `s_waitcnt vmcnt(..)` that waits for that load, can read the register's OLD contents in
// its LOW half while another wave of the same SIMD streams v_mfma_f32_32x32x16_bf16: the counter has dropped but the returning data has not
// reached the register file yet for the packed instruction's (earlier) operand fetch; the HIGH half — fetched a cycle later, even when op_sel
// points it at the SAME register — and every scalar VALU instruction see the new value.  Not beside v_mfma_f32_32x32x2_f32, plain VALU, LDS traffic,
// v_permlane32_swap or v_cvt_pk_bf16_f32 streams; not when a few wait states (or any other instruction) separate the s_waitcnt from the packed read.
// hipcc's hazard recogniser knows no such rule, and its SLP vectoriser forms exactly this sequence from `a.x = fmaf(x.x, w, a.x); a.y = fmaf(x.y, w, a.y)`
// with w fresh from a global_load_dwordx4 (viettts_amd/csrc/nat.hip: nat_dec_proj_prenet_k).
//
// This binary: VICTIM waves run, per iteration,   global_load_dwordx2 W, [ptr] ; s_waitcnt vmcnt(0) ; <gap> ; v_pk_fma_f32 acc, x, W, acc op_sel_hi:[1,0,1]
// and compare both halves with v_fma_f32 of the same operands (W re-read after the fact); AGGRESSOR waves on the other stream run a bare MFMA stream.
// Rows of the output: aggressor in {none, bf16 MFMA, fp32 MFMA} x gap in {0, s_nop 0, s_nop 1, s_nop 3} -> mismatching LOW / HIGH halves.
// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/kbench/pkfma_repro.hip -o tools/kbench/bin/pkfma_repro
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

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

__global__ __launch_bounds__(256, 2) void aggressor_k(int kind, volatile int* stop, float* sink) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.25f + 0.001f * (threadIdx.x + e));
        b[e] = (__bf16)(0.5f - 0.002f * e);
    }
    const float fa = 0.3f, fb = 0.7f;
    for (long it = 0; it < (1l << 40); ++it) {
        if ((it & 255) == 0 && *stop) break;
        if (kind == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(fa), "v"(fb));
        }
        if ((it & 255) == 255) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] *= 1e-6f;
        }
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[j][0] + acc[j][15];
    if (t == 123.456f) *sink = t;
}

// GAP: 0 = nothing between the s_waitcnt and the packed read, 1 = s_nop 0, 2 = s_nop 1, 3 = s_nop 3
// OVL: 1 = the load's destination registers ARE its address registers (what hipcc emitted in nat.hip: global_load_dwordx4 v[14:17], v[14:15], off)
#define VICTIM_ASM(GAPSTR)                                                                                                                              \
    do {                                                                                                                                                \
        if (OVL == 2) {                                                                                                                                 \
            /* the loop as hipcc emitted it: a 16-byte load whose destination v[40:43] starts at its own address registers, four LDS reads in flight, */ \
            /* ONE wait for both, the packed FMA right behind it reading the load's first register with the src1 broadcast */                         \
            unsigned wlo;                                                                                                                               \
            asm volatile("v_mov_b32 v40, %3\n\tv_mov_b32 v41, %4\n\t"                                                                                  \
                         "global_load_dwordx4 v[40:43], v[40:41], off\n\t"                                                                              \
                         "ds_read_b128 v[44:47], %5\n\tds_read_b128 v[48:51], %5 offset:16\n\tds_read_b128 v[52:55], %5 offset:32\n\tds_read_b128 v[56:59], %5 offset:48\n\t" \
                         "s_waitcnt vmcnt(0) lgkmcnt(3)\n\t" GAPSTR "v_pk_fma_f32 %0, %2, v[40:41], %0 op_sel_hi:[1,0,1]\n\t"                           \
                         "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\tv_mov_b32 %1, v40"                                                                         \
                         : "+v"(acc), "=&v"(wlo)                                                                                                        \
                         : "v"(x), "v"((unsigned)(unsigned long long)p), "v"((unsigned)((unsigned long long)p >> 32)), "v"(ldsaddr)                     \
                         : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",    \
                           "v56", "v57", "v58", "v59");                                                                                                 \
            wv = f32x2{__builtin_bit_cast(float, wlo), 0.f};                                                                                            \
        } else if (OVL) {                                                                                                                                      \
            unsigned long long aw = (unsigned long long)p;                                                                                              \
            asm volatile("global_load_dwordx2 %1, %1, off\n\ts_waitcnt vmcnt(0)\n\t" GAPSTR "v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[1,0,1]"            \
                         : "+v"(acc), "+v"(aw)                                                                                                          \
                         : "v"(x)                                                                                                                       \
                         : "memory");                                                                                                                   \
            wv = __builtin_bit_cast(f32x2, aw);                                                                                                         \
        } else {                                                                                                                                        \
            asm volatile("global_load_dwordx2 %1, %2, off\n\ts_waitcnt vmcnt(0)\n\t" GAPSTR "v_pk_fma_f32 %0, %3, %1, %0 op_sel_hi:[1,0,1]"            \
                         : "+v"(acc), "=&v"(wv)                                                                                                         \
                         : "v"(p), "v"(x)                                                                                                               \
                         : "memory");                                                                                                                   \
        }                                                                                                                                               \
    } while (0)
template <int GAP, int OVL>
__global__ __launch_bounds__(1024) void victim_k(const f32x2* __restrict__ w, int nw, int iters, unsigned* counts) {
    const int g = threadIdx.x + blockIdx.x * 1024;
    __shared__ __attribute__((aligned(16))) float ldsbuf[1024 * 16];
    for (int i = threadIdx.x; i < 1024 * 16; i += 1024) ldsbuf[i] = 0.001f * i;
    __syncthreads();
    const unsigned ldsaddr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)(ldsbuf + threadIdx.x * 16);
    (void)ldsaddr;
    // the packed accumulator travels as a 64-bit integer: hipcc 7.2 extracted BOTH halves of a float2 asm operand from its low register in the
    // comparison below (it compared v2 with r0 AND r1) — the integer form compiles as written
    unsigned long long acc = 0ull;
    float r0 = 0.f, r1 = 0.f;
    unsigned lo = 0, hi = 0;
    for (int it = 0; it < iters; ++it) {
        // consecutive lanes read consecutive 16-byte units (a wave's load is 1 KiB contiguous, an L2 hit after the first pass: as the weights' loads in nat.hip)
        const f32x2* p = w + (size_t)((((it * 3 + blockIdx.x) * 1024 + threadIdx.x) * 2) % nw);
        const f32x2 x = {0.5f + 0.001f * (it & 63), -0.25f + 0.002f * (threadIdx.x & 31)};
        f32x2 wv;
        // the load's destination is written by the VMEM return path; the packed FMA is the first instruction behind the wait
        if (GAP == 0) VICTIM_ASM("");
        if (GAP == 1) VICTIM_ASM("s_nop 0\n\t");
        if (GAP == 2) VICTIM_ASM("s_nop 1\n\t");
        if (GAP == 3) VICTIM_ASM("s_nop 3\n\t");
        // the same arithmetic with scalar instructions on the operand as it is NOW (long after the load)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(wv));
        r0 = __builtin_fmaf(x.x, wv.x, r0);
        r1 = __builtin_fmaf(x.y, wv.x, r1);
        lo += (unsigned)(acc & 0xffffffffull) != __builtin_bit_cast(unsigned, r0);
        hi += (unsigned)(acc >> 32) != __builtin_bit_cast(unsigned, r1);
        if ((it & 15) == 15) {
            r0 *= 0.5f;
            r1 *= 0.5f;
        }
        acc = (unsigned long long)__builtin_bit_cast(unsigned, r0) | ((unsigned long long)__builtin_bit_cast(unsigned, r1) << 32);  // re-synchronise: one bad read is one count
    }
    if (lo) atomicAdd(&counts[0], lo);
    if (hi) atomicAdd(&counts[1], hi);
}

template <int GAP, int OVL>
static void run_victims(hipStream_t s, const f32x2* w, int nw, int iters, unsigned* counts, int ms, long* launches) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    *launches = 0;
    for (;;) {
        for (int q = 0; q < 32; ++q) hipLaunchKernelGGL((victim_k<GAP, OVL>), dim3(3), dim3(1024), 0, s, w, nw, iters, counts);
        *launches += 32;
        CK(hipStreamSynchronize(s));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > ms) break;
    }
}

int main(int argc, char** argv) {
    const int ms = argc > 1 ? atoi(argv[1]) : 1000, iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int nw = 1 << 20;
    f32x2* w;
    unsigned* counts;
    float* sink;
    int* stop;
    CK(hipMalloc(&w, (size_t)nw * sizeof(f32x2)));
    {
        f32x2* h = (f32x2*)malloc((size_t)nw * sizeof(f32x2));
        for (int i = 0; i < nw; ++i) h[i] = f32x2{0.001f * (i % 977) - 0.4f, 0.002f * (i % 313)};
        CK(hipMemcpy(w, h, (size_t)nw * sizeof(f32x2), hipMemcpyHostToDevice));
        free(h);
    }
    CK(hipMalloc(&counts, 16));
    CK(hipMalloc(&sink, 4));
    CK(hipHostMalloc(&stop, 4, hipHostMallocMapped));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const char* an[3] = {"nothing", "v_mfma_f32_32x32x16_bf16 stream", "v_mfma_f32_32x32x2_f32 stream"};
    const char* gn[4] = {"none", "s_nop 0", "s_nop 1", "s_nop 3"};
    printf("victim: 3 workgroups x 1024 threads, %d x (global_load_dwordx2 W ; s_waitcnt vmcnt(0) ; <gap> ; v_pk_fma_f32 acc, x, W, acc op_sel_hi:[1,0,1]) per lane and launch, ~%d ms per case\n", iters, ms);
    for (int ovl = 2; ovl >= 0; --ovl)
    for (int agg = 0; agg < 3; ++agg)
        for (int gap = 0; gap < 4; ++gap) {
            CK(hipMemset(counts, 0, 16));
            *stop = 0;
            if (agg) hipLaunchKernelGGL(aggressor_k, dim3(prop.multiProcessorCount * 2), dim3(256), 0, sa, agg, stop, sink);
            struct timespec ts = {0, 50 * 1000 * 1000};
            nanosleep(&ts, nullptr);
            long launches = 0;
            if (ovl == 2) {
                if (gap == 0) run_victims<0, 2>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 1) run_victims<1, 2>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 2) run_victims<2, 2>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 3) run_victims<3, 2>(sb, w, nw, iters, counts, ms, &launches);
            } else if (ovl) {
                if (gap == 0) run_victims<0, 1>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 1) run_victims<1, 1>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 2) run_victims<2, 1>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 3) run_victims<3, 1>(sb, w, nw, iters, counts, ms, &launches);
            } else {
                if (gap == 0) run_victims<0, 0>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 1) run_victims<1, 0>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 2) run_victims<2, 0>(sb, w, nw, iters, counts, ms, &launches);
                if (gap == 3) run_victims<3, 0>(sb, w, nw, iters, counts, ms, &launches);
            }
            *stop = 1;
            if (agg) CK(hipStreamSynchronize(sa));
            unsigned c[4];
            CK(hipMemcpy(c, counts, 16, hipMemcpyDeviceToHost));
            printf("dst %s | beside %-34s gap %-8s: %6ld victim launches, %10.3e packed FMAs, LOW halves wrong %9u, HIGH halves wrong %9u\n", ovl == 2 ? "x4 + 4 LDS reads" : (ovl ? "= address regs  " : "other regs      "), an[agg], gn[gap], launches,
                   (double)launches * 3 * 1024 * iters, c[0], c[1]);
            fflush(stdout);
        }
    return 0;
}
