// Kernel-development microbenchmark (NOT part of the product): what two waves on ONE SIMD of gfx950 cost each other when one
// streams v_mfma_f32_32x32x16_bf16 and the other issues VALU / LDS / VMEM work — the situation of the fused pair kernel's two
// co-resident workgroups (profiles/r01_j_kbench_findings.md: "next to an MFMA stream the other wave's VALU gets about one
// issue slot per MFMA").  One 512-thread workgroup per CU (LDS-padded): waves 0-3 take role A (MFMA), waves 4-7 role B; wave i
// and wave i + 4 land on the same SIMD (checked from HW_ID and reported).
//   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/coissue.hip -o tools/kbench/bin/coissue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

struct Rec {
    unsigned long long t0, t1;
    unsigned hwid, role;
};

// role A: ITERS x 8 MFMAs on 8 independent accumulators, NOP wait states after each (NOP < 0: none); PRIO: s_setprio
template <int NOP, int NOP2>
__device__ __forceinline__ void mfma_stream(int iters, float* sink) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.001f * (threadIdx.x & 31));
        b[e] = (__bf16)(0.002f * (threadIdx.x & 15));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            if (NOP >= 0) asm volatile("s_nop %0" ::"n"(NOP < 0 ? 0 : NOP));
            if (NOP2 >= 0) asm volatile("s_nop %0" ::"n"(NOP2 < 0 ? 0 : NOP2));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][15];
    if (s == 123.456f) *sink = s;
}

// role B bodies: `iters` x 32 instructions
__device__ __forceinline__ void valu_indep(int iters, float* sink) {  // 8 independent fma chains
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * threadIdx.x + j;
    const float m = 0.999f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 123.456f) *sink = s;
}
__device__ __forceinline__ void valu_dep(int iters, float* sink) {  // one dependent chain
    float v = 0.001f * threadIdx.x;
    const float m = 0.999f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(m), "v"(c));
    }
    if (v == 123.456f) *sink = v;
}
// the epilogue's mix: per 8 values 4 x (v_pk_mul_f32, 2 v_max_f32, v_cvt_pk_bf16_f32) + one ds_write_b128 = 17 instructions; x 2 = 34 ("32")
__device__ __forceinline__ void valu_epi(int iters, float* sink, unsigned char* lds) {
    f32x2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = f32x2{0.001f * threadIdx.x + j, 0.5f * j};
    const f32x2 k = {0.1f, 0.1f};
    unsigned addr = (threadIdx.x & 255) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            unsigned p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 m;
                float a0, a1;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(v[j]), "v"(k));
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(a0) : "v"(v[j].x), "v"(m.x));
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(a1) : "v"(v[j].y), "v"(m.y));
                asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[j]) : "v"(a0), "v"(a1));
            }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 q = {p[0], p[1], p[2], p[3]};
            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(q) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (v[0].x == 123.456f) *sink = v[0].x;
}
// LDS fragment reads only: 32 ds_read_b128 per iteration, waited in groups of 8
__device__ __forceinline__ void lds_reads(int iters, float* sink) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned addr = (threadIdx.x & 255) * 16;
    u32x4 q[8];
    unsigned s = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[j]) : "v"(addr), "n"(j * 4096));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) s ^= q[j][0];
        }
    }
    if (s == 0x12345u) *sink = 1.f;
}

// one instruction type at a time (32 per iteration), to find which ones an MFMA stream on the same SIMD blocks
template <int WHICH>
__device__ __forceinline__ void valu_one(int iters, float* sink, unsigned char* lds, const float* gmem) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x2 v[4], m[4];
    unsigned u[4];
    float a[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = f32x2{0.001f * threadIdx.x + j, 0.5f * j};
        m[j] = v[j];
        u[j] = threadIdx.x * 7 + j;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.25f * j + threadIdx.x;
    const f32x2 k = {0.1f, 0.1f};
    const float ks = 0.1f;
    unsigned addr = (threadIdx.x & 255) * 16;
    u32x4 q = {1u, 2u, 3u, 4u};
    u32x4 ld[4];
    const float* gp = gmem + (threadIdx.x & 255) * 4;
    const unsigned ldsbase = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds + (threadIdx.x >> 6) * 8192);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (WHICH == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m[j]) : "v"(v[j]), "v"(k));
                if (WHICH == 1) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[j]) : "v"(a[j + 4]), "v"(ks));
                if (WHICH == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[j]) : "v"(a[j]), "v"(a[j + 4]));
                if (WHICH == 3) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(q), "n"(j * 4096) : "memory");
                if (WHICH == 4) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[j]) : "v"(a[j + 4]), "v"(ks));
                if (WHICH == 5) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[j]) : "v"(u[(j + 1) & 3]), "v"(addr));
                if (WHICH == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 3]));
                if (WHICH == 7) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[j]), "+v"(u[(j + 1) & 3]));
                if (WHICH == 8) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[j]) : "v"(u[(j + 1) & 3]), "v"(addr) : );
                if (WHICH == 9) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ld[j]) : "v"(gp), "n"(j * 1024) : "memory");
                if (WHICH == 10) asm volatile("global_store_dwordx4 %0, %1, off offset:%2" ::"v"(gp), "v"(q), "n"(j * 1024) : "memory");
                if (WHICH == 11) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v[j]), "n"(j * 4096) : "memory");
                if (WHICH == 12) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[j]) : "v"(a[j + 4]), "v"(ks));
                if (WHICH == 13) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(m[j]) : "v"(v[j]), "v"(k));
                if (WHICH == 14) asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(m[j]) : "v"(v[j]), "v"(k));
                if (WHICH == 15) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(u[j]), "n"(j * 4096) : "memory");
                if (WHICH == 16) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[j]) : "v"(u[(j + 1) & 3]), "v"(addr), "v"(u[(j + 2) & 3]));
                if (WHICH == 17) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(a[j]) : "v"(a[j + 4]), "v"(ks));
                // LDS-DMA: 16 bytes per lane straight into LDS (no VGPR write-back); M0 = LDS base of this wave's 1 KiB piece
                if (WHICH == 18) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" ::"v"(gp), "s"(__builtin_amdgcn_readfirstlane(ldsbase + j * 4096)), "n"(j * 1024) : "memory");
                // 16 loads in flight, one wait (the staging pattern): 19 = into VGPRs, 20 = LDS-DMA
                if (WHICH == 19) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ld[j]) : "v"(gp + r * 4096), "n"(j * 1024) : "memory");
                if (WHICH == 20) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" ::"v"(gp + r * 4096), "s"(__builtin_amdgcn_readfirstlane(ldsbase + (r * 4 + j) * 1024)), "n"(j * 1024) : "memory");
                if (WHICH == 21) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j]) : "v"(addr), "n"(j * 4096));
            }
            if (WHICH == 9 || WHICH == 18) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (WHICH == 21) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((WHICH == 19 || WHICH == 20) && (r & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float sres = m[0].x + a[0] + __builtin_bit_cast(float, u[0]) + (WHICH == 9 ? __builtin_bit_cast(float, ld[0][0]) : 0.f);
    if (sres == 123.456f) *sink = sres;
}
static const char* one_names[] = {"v_pk_mul_f32", "v_max_f32", "v_cvt_pk_bf16_f32", "ds_write_b128", "v_mul_f32", "v_and_b32", "v_lshlrev_b32", "v_permlane32_swap",
                                  "v_cndmask_b32", "global_load_dwordx4 (L2 hits, waited per 4)", "global_store_dwordx4", "ds_write_b64", "v_add_f32", "v_pk_add_f32",
                                  "v_pk_fma_f32", "ds_write_b32", "v_perm_b32", "v_fma_f32", "global_load_lds_dwordx4 (LDS-DMA, waited per 4)",
                                  "global_load_dwordx4, 16 in flight per wait", "global_load_lds_dwordx4, 16 in flight per wait", "ds_read_b128 (waited per 4)"};
constexpr int N_ONE = 22;

template <int NOP, int NOP2>
__global__ __launch_bounds__(512) void coissue_k(Rec* out, float* sink, int mode_b, int prio_a, int prio_b, int iters_a, int iters_b, const float* gmem) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = threadIdx.x >> 6;
    const bool role_a = wave < 4;
    if (role_a) {
        if (prio_a == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_a == 3) __builtin_amdgcn_s_setprio(3);
    } else {
        if (prio_b == 1) __builtin_amdgcn_s_setprio(1);
        if (prio_b == 3) __builtin_amdgcn_s_setprio(3);
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role_a) {
        if (iters_a > 0) mfma_stream<NOP, NOP2>(iters_a, sink);
    } else if (iters_b > 0) {
        if (mode_b == 1) valu_indep(iters_b, sink);
        else if (mode_b == 2) valu_dep(iters_b, sink);
        else if (mode_b == 3) valu_epi(iters_b, sink, lds);
        else if (mode_b == 4) lds_reads(iters_b, sink);
        else if (mode_b == 5) mfma_stream<-1, -1>(iters_b / 8 > 0 ? iters_b / 8 : 1, sink);  // a second MFMA stream (loop beside loop)
        else if (mode_b >= 100) {
            switch (mode_b - 100) {
#define ONE(n) case n: valu_one<n>(iters_b, sink, lds, gmem); break;
                ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15) ONE(16) ONE(17) ONE(18) ONE(19) ONE(20) ONE(21)
#undef ONE
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        r.t0 = t0;
        r.t1 = t1;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r.role = role_a ? 0 : 1;
        out[blockIdx.x * 8 + wave] = r;
    }
}

// intra-wave overlap: every MFMA followed by FILL plain VALU instructions (the epilogue's kinds: v_mul, v_max, v_cvt_pk on 4 independent
// chains) and optionally one ds_read_b128 — what a software-pipelined loop (the other column half's epilogue under this half's MFMAs) issues
template <int FILL, int DSR>
__global__ __launch_bounds__(512) void filler_k(Rec* out, float* sink, int both, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.001f * (threadIdx.x & 31));
        b[e] = (__bf16)(0.002f * (threadIdx.x & 15));
    }
    float v[8];
    unsigned u[4] = {1u, 2u, 3u, 4u};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.25f * j + threadIdx.x;
    const float ks = 0.1f;
    const unsigned addr = (threadIdx.x & 255) * 16;
    u32x4 q;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4 || both) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
                if (DSR) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "n"(j * 4096));
#pragma unroll
                for (int f = 0; f < FILL; ++f) {
                    const int c = (j * FILL + f) & 3;
                    if ((f % 3) == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[c]) : "v"(v[c + 4]), "v"(ks));
                    if ((f % 3) == 1) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[c + 4]) : "v"(v[c]), "v"(v[c + 4]));
                    if ((f % 3) == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[c]) : "v"(v[c]), "v"(v[c + 4]));
                }
            }
            if (DSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sres = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v[0] + __builtin_bit_cast(float, u[0]) + (DSR ? __builtin_bit_cast(float, q[0]) : 0.f);
    if (sres == 123.456f) *sink = sres;
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        r.t0 = t0;
        r.t1 = t1;
        r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r.role = wave < 4 ? 0 : 1;
        out[blockIdx.x * 8 + wave] = r;
    }
}
// the chip's SUSTAINED bf16 MFMA rate (power-limited clock) as a function of the operand data: every SIMD runs two waves of back-to-back
// v_mfma_f32_32x32x16_bf16 on 8 accumulators; operands zero / one constant / pseudo-random per lane, refreshed from a small register pool
__global__ __launch_bounds__(512) void mfma_peak_k(float* sink, int iters, int data_mode, unsigned seed) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a[4], b[4];
    unsigned h = (threadIdx.x + blockIdx.x * 512u) * 2654435761u + seed;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u;
            const float ra = ((int)(h >> 8) % 2001 - 1000) * 1e-3f;
            h = h * 1664525u + 1013904223u;
            const float rb = ((int)(h >> 8) % 2001 - 1000) * 1e-3f;
            a[q][e] = (__bf16)(data_mode == 0 ? 0.f : data_mode == 1 ? 0.5f : ra);
            b[q][e] = (__bf16)(data_mode == 0 ? 0.f : data_mode == 1 ? 0.25f : rb);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[q]), "v"(b[(q + j) & 3]));
    }
    float sres = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sres += acc[j][0] + acc[j][7];
    if (sres == 123.456f) *sink = sres;
}

typedef void (*fill_t)(Rec*, float*, int, int);

typedef void (*kern_t)(Rec*, float*, int, int, int, int, int, const float*);
struct Variant {
    const char* name;
    kern_t k;
};

int main(int argc, char** argv) {
    const int nwg = 256, iters_a = argc > 1 ? atoi(argv[1]) : 400;  // 3200 MFMAs per wave
    Rec* d;
    float* sink;
    CK(hipMalloc(&d, nwg * 8 * sizeof(Rec)));
    CK(hipMalloc(&sink, 4));
    float* gmem;
    CK(hipMalloc(&gmem, 1 << 20));
    CK(hipMemset(gmem, 0, 1 << 20));
    const bool fine = argc > 2 && atoi(argv[2]) == 1;  // 1: one instruction type at a time instead of the mixes
    std::vector<Rec> h(nwg * 8);
    const Variant vs[] = {
        {"none", coissue_k<-1, -1>}, {"s_nop 0", coissue_k<0, -1>},  {"s_nop 1", coissue_k<1, -1>},  {"s_nop 3", coissue_k<3, -1>},
        {"s_nop 5", coissue_k<5, -1>}, {"s_nop 7", coissue_k<7, -1>}, {"s_nop 11", coissue_k<11, -1>}, {"s_nop 15", coissue_k<15, -1>},
        {"s_nop 15+7", coissue_k<15, 7>}, {"s_nop 15+15", coissue_k<15, 15>},
    };
    const char* bnames[] = {"idle", "VALU 8 independent chains", "VALU dependent chain", "epilogue mix (pk_mul,max,max,cvt_pk x4 + ds_write_b128)",
                            "ds_read_b128 x8 + wait", "second MFMA stream"};
    for (auto& v : vs) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    auto run = [&](const Variant& v, int mode_b, int pa, int pb, int ia, int ib, double& ca, double& cb, int& same_simd) {
        hipLaunchKernelGGL(v.k, dim3(nwg), dim3(512), 100 * 1024, 0, d, sink, mode_b, pa, pb, ia, ib, gmem);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(v.k, dim3(nwg), dim3(512), 100 * 1024, 0, d, sink, mode_b, pa, pb, ia, ib, gmem);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
        double sa = 0, sb = 0;
        same_simd = 0;
        for (int w = 0; w < nwg; ++w) {
            for (int i = 0; i < 4; ++i) {
                sa += (double)(h[w * 8 + i].t1 - h[w * 8 + i].t0);
                sb += (double)(h[w * 8 + 4 + i].t1 - h[w * 8 + 4 + i].t0);
                if (((h[w * 8 + i].hwid >> 4) & 3) == ((h[w * 8 + 4 + i].hwid >> 4) & 3)) same_simd++;
            }
        }
        ca = sa / (nwg * 4);
        cb = sb / (nwg * 4);
    };
    printf("s_memtime ticks (100 MHz constant clock on gfx950? reported as-is); per-instruction figures are ticks / instruction count\n");
    // calibrate: A alone, B alone
    double ca, cb;
    int ss;
    run(vs[0], 0, 0, 0, iters_a, 0, ca, cb, ss);
    const double a_alone = ca / (iters_a * 8.0);
    printf("A alone (no pacing): %.1f ticks = %.3f ticks / MFMA   [waves i and i+4 on the same SIMD: %d of %d]\n", ca, a_alone, ss, nwg * 4);
    for (auto& v : vs) {
        run(v, 0, 0, 0, iters_a, 0, ca, cb, ss);
        printf("  A alone, %-12s : %.3f ticks / MFMA (x%.3f)\n", v.name, ca / (iters_a * 8.0), ca / (iters_a * 8.0) / a_alone);
    }
    if (argc > 2 && atoi(argv[2]) == 3) {  // sustained MFMA rate vs operand data
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const char* names[] = {"all-zero operands", "one constant per operand", "pseudo-random operands in [-1, 1]"};
        const int iters = 20000;  // x 32 MFMAs per wave
        printf("sustained v_mfma_f32_32x32x16_bf16 rate, 256 workgroups x 8 waves (2 per SIMD), %d MFMAs per wave, three runs each:\n", iters * 32);
        for (int dm = 0; dm < 3; ++dm) {
            printf("  %-36s", names[dm]);
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(mfma_peak_k, dim3(nwg), dim3(512), 0, 0, sink, rep == 0 ? 2000 : iters, dm, 12345u + rep);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(mfma_peak_k, dim3(nwg), dim3(512), 0, 0, sink, iters, dm, 777u + rep);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double fl = (double)nwg * 8 * iters * 32.0 * 32768.0;
                printf("  %7.1f TF/s (%.2f of 2500; %.2f ms)", fl / ms * 1e-9, fl / ms * 1e-9 / 2500.0, ms);
            }
            printf("\n");
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 4) {  // a sustained MFMA-only load for power / clock sampling: coissue <iters> 4 <data mode 0|1|2> <seconds>
        const int dm = argc > 3 ? atoi(argv[3]) : 2;
        const double secs = argc > 4 ? atof(argv[4]) : 8.0;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int iters = 20000;
        double total_ms = 0.0;
        int n = 0;
        while (total_ms < secs * 1e3) {
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(mfma_peak_k, dim3(nwg), dim3(512), 0, 0, sink, iters, dm, 777u + n * 8 + r);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            n += 8;
        }
        const double fl = (double)nwg * 8 * iters * 32.0 * 32768.0 * n;
        printf("MFMA-only load, data mode %d: %d launches in %.2f s = %.1f TF/s (%.2f of 2500)\n", dm, n, total_ms * 1e-3, fl / total_ms * 1e-9, fl / total_ms * 1e-9 / 2500.0);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 2) {  // intra-wave fillers
        struct FV { const char* name; fill_t k; };
        const FV fv[] = {{"0", filler_k<0, 0>}, {"2", filler_k<2, 0>}, {"4", filler_k<4, 0>}, {"6", filler_k<6, 0>}, {"8", filler_k<8, 0>}, {"12", filler_k<12, 0>},
                         {"0+ds_read", filler_k<0, 1>}, {"4+ds_read", filler_k<4, 1>}, {"6+ds_read", filler_k<6, 1>}, {"8+ds_read", filler_k<8, 1>}};
        printf("cycles per MFMA of a wave whose every MFMA is followed by F plain VALU instructions (v_mul / v_max / v_cvt_pk, 4 chains) [+ one ds_read_b128]:\n");
        for (auto& f : fv) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(f.k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
            double r[2];
            for (int both = 0; both < 2; ++both) {
                for (int rep = 0; rep < 2; ++rep) {
                    hipLaunchKernelGGL(f.k, dim3(nwg), dim3(512), 100 * 1024, 0, d, sink, both, 800);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(h.data(), d, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
                double sa = 0;
                for (int w = 0; w < nwg; ++w)
                    for (int i = 0; i < (both ? 8 : 4); ++i) sa += (double)(h[w * 8 + i].t1 - h[w * 8 + i].t0);
                r[both] = sa / (nwg * (both ? 8 : 4)) / (800 * 4.0);
            }
            printf("  F = %-10s one wave per SIMD: %6.2f cycles / MFMA    two waves per SIMD (both the same stream): %6.2f cycles / MFMA per wave = %6.2f per SIMD\n", f.name, r[0], r[1], r[1] / 2);
        }
        return 0;
    }
    if (fine) {
        const int vsel[] = {0, 2, 5, 6};  // A unpaced, s_nop 1, s_nop 7, s_nop 11
        for (int w = 0; w < N_ONE; ++w) {
            int ib = 100;
            run(vs[0], 100 + w, 0, 0, 0, ib, ca, cb, ss);
            const double b_alone = cb / (ib * 32.0);
            ib = (int)(a_alone * iters_a * 8.0 / (b_alone * 32.0));
            if (ib < 1) ib = 1;
            run(vs[0], 100 + w, 0, 0, 0, ib, ca, cb, ss);
            const double b_alone_t = cb, a_alone_t = a_alone * iters_a * 8.0;
            printf("\nB = %-44s alone %6.2f ticks/instr |", one_names[w], b_alone);
            for (int vi : vsel)
                for (int pb = 0; pb <= 3; pb += 3) {
                    run(vs[vi], 100 + w, 0, pb, iters_a, ib, ca, cb, ss);
                    // overlap = how much of the shorter stream hid under the longer one: 1 = perfect, 0 = serial
                    const double both = ca > cb ? ca : cb, ser = a_alone_t + b_alone_t, ideal = a_alone_t > b_alone_t ? a_alone_t : b_alone_t;
                    printf("  [%s,pB=%d] A x%.2f B x%.2f ovl %.2f |", vs[vi].name, pb, ca / a_alone_t, cb / b_alone_t, (ser - both) / (ser - ideal));
                }
        }
        printf("\n");
        return 0;
    }
    for (int mb = 1; mb <= 5; ++mb) {
        // size B so that alone it takes about as long as A alone
        int ib = 200;
        run(vs[0], mb, 0, 0, 0, ib, ca, cb, ss);
        const double b_alone = cb / (ib * 32.0);
        ib = (int)(a_alone * iters_a * 8.0 / (b_alone * 32.0));
        if (ib < 1) ib = 1;
        run(vs[0], mb, 0, 0, 0, ib, ca, cb, ss);
        const double b_alone_t = cb;
        printf("\nB = %s: alone %.3f ticks / instr; %d iterations (%.0f ticks alone; A alone %.0f)\n", bnames[mb], b_alone, ib, b_alone_t, a_alone * iters_a * 8.0);
        for (auto& v : vs) {
            for (int pr = 0; pr < 3; ++pr) {
                const int pa = pr == 1 ? 3 : 0, pb = pr == 2 ? 3 : 0;
                run(v, mb, pa, pb, iters_a, ib, ca, cb, ss);
                printf("  A %-12s prio A=%d B=%d : A %.3f ticks/MFMA (x%.2f)   B %.3f ticks/instr (x%.2f)   both done after %.0f ticks (serial %.0f, ideal %.0f)\n", v.name, pa, pb,
                       ca / (iters_a * 8.0), ca / (iters_a * 8.0) / a_alone, cb / (ib * 32.0), cb / (ib * 32.0) / b_alone, ca > cb ? ca : cb,
                       a_alone * iters_a * 8.0 + b_alone_t, a_alone * iters_a * 8.0 > b_alone_t ? a_alone * iters_a * 8.0 : b_alone_t);
            }
        }
    }
    return 0;
}
