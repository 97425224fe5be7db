// Kernel-development microbenchmark (NOT part of the product): what does a barrier across the workgroups of a resident grid cost on MI355X?
// The NAT decoder is a chain of three dependent launches per frame (~16 us each, of which a third is matrix work): a persistent decoder kernel
// would replace launch boundaries by barriers among the 64 workgroups that share a sentence tile.  This measures that barrier — agent-scope
// release fence by every thread, workgroup barrier, one atomic arrive + spin per workgroup, acquire fence — with a coherence check (every
// workgroup writes a line per round and reads its neighbour's after the barrier), for groups of 64 (4 independent groups) and of 256.
//   build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/grid_barrier.hip -o tools/kbench/bin/grid_barrier
//   run:   tools/kbench/bin/grid_barrier [rounds=2000]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                                   \
        }                                                                                              \
    } while (0)

// one barrier among `members` workgroups sharing `ctr` (monotonic: round r waits for members * (r + 1)); bounded spin.
// MODE 0: no fences at all (arrive + spin only: the floor; data may be stale)
// MODE 1: agent-scope release fence by EVERY thread before, acquire fence by every thread after (what a compiler-level grid sync does)
// MODE 2: the fences by wave 0 only (the other waves' stores are ordered by the workgroup barrier + s_waitcnt vmcnt(0))
// MODE 3: no cache-wide fences: the payload is written / read with agent-scope relaxed ATOMIC stores / loads (sc1: through the L2s), s_waitcnt
//         vmcnt(0) before the workgroup barrier
// MODE 4: sc1 atomic stores (write-through), NO release fence; after the barrier wave 0 runs an acquire fence (buffer_inv sc1) and the payload is
//         read with PLAIN 16-byte loads (what a kernel that re-reads its neighbours' state many times wants: the reads may hit in L2)
// MODE 5: as 4 without the acquire fence (expected stale)
// In modes 4 and 5 every workgroup reads the payload of ALL members of its group (the LSTM step reads the whole state of its sentence tile).
template <int MODE>
__device__ __forceinline__ void group_barrier(unsigned* ctr, unsigned target, int* fail) {
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (MODE >= 2) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's stores have left
    __syncthreads();
    if (threadIdx.x < 64) {
        if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {
                    *fail = 1;
                    break;
                }
            }
        }
        if (MODE == 2 || MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int MODE>
__global__ __launch_bounds__(512) void barrier_k(unsigned* ctrs, float* slots, int members, int rounds, int payload, int* fail, unsigned* bad) {
    const int wg = blockIdx.x, grp = wg / members, me = wg % members;
    unsigned* ctr = ctrs + grp * 64;  // a counter per group, 256 bytes apart
    float* mine = slots + (size_t)wg * payload;
    float* nb = slots + (size_t)(grp * members + (me + 1) % members) * payload;
    unsigned wrong = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < payload; i += blockDim.x) {
            if (MODE >= 3) __hip_atomic_store(mine + i, (float)(r * 7 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[i] = (float)(r * 7 + i);
        }
        group_barrier<MODE>(ctr, (unsigned)members * (2 * r + 1), fail);
        if (MODE >= 4) {
            const float4* all = reinterpret_cast<const float4*>(slots + (size_t)grp * members * payload);
            for (int i = threadIdx.x; i < members * payload / 4; i += blockDim.x) {
                const float4 v = all[i];
                const int j = (4 * i) % payload;
                wrong += (v.x != (float)(r * 7 + j)) + (v.y != (float)(r * 7 + j + 1)) + (v.z != (float)(r * 7 + j + 2)) + (v.w != (float)(r * 7 + j + 3));
            }
        } else
        for (int i = threadIdx.x; i < payload; i += blockDim.x) {
            const float v = MODE == 3 ? __hip_atomic_load(nb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : nb[i];
            wrong += v != (float)(r * 7 + i);
        }
        group_barrier<MODE>(ctr, (unsigned)members * (2 * r + 2), fail);  // nobody overwrites before everybody has read
        if (*fail) break;
    }
    if (wrong) atomicAdd(bad, wrong);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned *ctrs, *bad;
    float* slots;
    int* fail;
    const int WGS = 256, PAYMAX = 8192;
    CK(hipMalloc(&ctrs, 4 * 64 * sizeof(unsigned)));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&fail, 4));
    CK(hipMalloc(&slots, (size_t)WGS * PAYMAX * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const void* kerns[6] = {reinterpret_cast<const void*>(&barrier_k<0>), reinterpret_cast<const void*>(&barrier_k<1>), reinterpret_cast<const void*>(&barrier_k<2>),
                            reinterpret_cast<const void*>(&barrier_k<3>), reinterpret_cast<const void*>(&barrier_k<4>), reinterpret_cast<const void*>(&barrier_k<5>)};
    const char* mnames[6] = {"no fences (floor, may read stale)", "fences by every thread", "fences by wave 0 only", "no fences, sc1 atomic stores/loads",
                             "sc1 stores, acquire fence by wave 0, plain loads of the whole group's payload", "sc1 stores, no fence, plain loads of the whole group's payload"};
    for (int mode = 0; mode < 6; ++mode)
    for (int members : {64, 256})
        for (int payload : {64, 512}) {
            CK(hipMemset(ctrs, 0, 4 * 64 * sizeof(unsigned)));
            CK(hipMemset(bad, 0, 4));
            CK(hipMemset(fail, 0, 4));
            int r = rounds;
            void* args[] = {&ctrs, &slots, (void*)&members, &r, (void*)&payload, &fail, &bad};
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel(kerns[mode], dim3(WGS), dim3(512), args, 0, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned hb = 0;
            int hf = 0;
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
            printf("[%s] groups of %3d workgroups (256 resident, 512 threads each), %5d floats written + read per workgroup and round: %.2f us per barrier (2 per round, %d rounds)%s, stale reads %u\n",
                   mnames[mode], members, payload, ms * 1e3 / (2.0 * rounds), rounds, hf ? "  SPIN LIMIT HIT" : "", hb);
        }
    return 0;
}
