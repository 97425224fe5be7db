// Kernel-development harness (NOT part of the product, never loaded by it): times one fused ResBlock-pair
// launch of the bf16 path in isolation, checks it against a naive on-device restatement of
// vietTTS/hifigan/model.py:45-50 on the same bf16 operands, and (built with -DVTTS_TIMELINE=1) reports
// where a workgroup's cycles go.  Build + run:  tools/kbench/build.sh && tools/kbench/kbench [C K dil B L reps impl]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <random>
#include <vector>

#include "vtts_internal.h"

using namespace vtts;
namespace vtts {
hipError_t launch_pair_p_bf16(int C, int K, const BConvArgs& a, hipStream_t s);
}

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

static unsigned short f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

__device__ __forceinline__ float d_bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ unsigned short d_f2bf(float f) {
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float d_lrelu(float v, float s) { return v >= 0.f ? v : v * s; }

// naive reference, one thread per (t, co): xt = bf16(lrelu(c1(bf16(lrelu(x))) + b1)); y = bf16(c2(xt) + b2 + x)
// W layout: [K][Cin][Cout] fp32 holding bf16-representable values.
__global__ void ref_c1_k(const unsigned short* x, const float* w1, const float* b1, unsigned short* xt, int L, int C, int K, int dil, int t_lo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const int co = i % C, t = t_lo + i / C;
    if (t < 0 || t >= L) return;
    const int pad = (K - 1) / 2 * dil;
    float acc = 0.f;
    for (int j = 0; j < K; ++j) {
        const int ts = t + j * dil - pad;
        if (ts < 0 || ts >= L) continue;
        for (int ci = 0; ci < C; ++ci) {
            const float xv = d_bf2f(d_f2bf(d_lrelu(d_bf2f(x[(size_t)ts * C + ci]), 0.1f)));
            acc = fmaf(w1[((size_t)j * C + ci) * C + co], xv, acc);
        }
    }
    xt[(size_t)t * C + co] = d_f2bf(d_lrelu(acc + b1[co], 0.1f));
}
__global__ void ref_c2_k(const unsigned short* x, const unsigned short* xt, const float* w2, const float* b2, float* y, int L, int C, int K, int t_lo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const int co = i % C, t = t_lo + i / C;
    if (t < 0 || t >= L) return;
    const int pad = (K - 1) / 2;
    float acc = 0.f;
    for (int j = 0; j < K; ++j) {
        const int ts = t + j - pad;
        if (ts < 0 || ts >= L) continue;
        for (int ci = 0; ci < C; ++ci) acc = fmaf(w2[((size_t)j * C + ci) * C + co], d_bf2f(xt[(size_t)ts * C + ci]), acc);
    }
    y[(size_t)(t - t_lo) * C + co] = acc + b2[co] + d_bf2f(x[(size_t)t * C + co]);
}

int main(int argc, char** argv) {
    int C = 128, K = 11, dil = 3, B = 16, L = 65536, reps = 5, impl = 0;
    if (argc > 1) C = atoi(argv[1]);
    if (argc > 2) K = atoi(argv[2]);
    if (argc > 3) dil = atoi(argv[3]);
    if (argc > 4) B = atoi(argv[4]);
    if (argc > 5) L = atoi(argv[5]);
    if (argc > 6) reps = atoi(argv[6]);
    int dflags = 0, stagger = 0;
    if (argc > 9) stagger = atoi(argv[9]);  // development builds with a start-offset switch (none in the tree now)
    if (argc > 8) dflags = atoi(argv[8]);  // timeline builds of kernels_bf16_rb.hip: experiment switches (results wrong)
    if (argc > 7) impl = atoi(argv[7]);  // kept for old command lines; there is one pair kernel (kernels_bf16_rbg.hip)
    const int acc_add = getenv("KB_ACC") ? atoi(getenv("KB_ACC")) : 0;  // 1: the chain's third launch (y = y + pair(x): MRF accumulate, model.py:118-120)
    printf("pair C=%d K=%d dil=%d B=%d L=%d impl=%d dflags=%d stagger=%d acc_add=%d\n", C, K, dil, B, L, impl, dflags, stagger, acc_add);
    // impl 0: second generation (two workgroups per CU); impl 1: persistent software-pipelined kernel (kernels_bf16_rbp.hip)
#ifdef KBENCH_NO_P
    auto launch = [&](const BConvArgs& aa) { return launch_pair_g_bf16(C, K, aa, 0); };
#else  // build with tools/kbench/experiments/kernels_bf16_rbp.hip on the command line
    auto launch = [&](const BConvArgs& aa) { return impl == 1 ? launch_pair_p_bf16(C, K, aa, 0) : launch_pair_g_bf16(C, K, aa, 0); };
#endif
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t n = (size_t)B * L * C;
    std::vector<unsigned short> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = f2bf(1.5f * nd(rng));
    const float ws = 1.0f / sqrtf((float)C * K);
    std::vector<float> w1((size_t)K * C * C), w2((size_t)K * C * C), bias(2 * C);
    for (auto& v : w1) v = bf2f(f2bf(ws * nd(rng)));
    for (auto& v : w2) v = bf2f(f2bf(ws * nd(rng)));
    for (auto& v : bias) v = 0.1f * nd(rng);
    const BPackGeom g = pair_g_pack_geom(C, K);
    const size_t pb = bf16_packed_bytes(g);
    std::vector<unsigned short> wp(pb);  // 2 * pb bytes
    bf16_pack(w1.data(), C, g, wp.data());
    bf16_pack(w2.data(), C, g, wp.data() + pb / 2);

    unsigned short *dx, *dy, *dwp, *dxt;
    float *dbias, *dw1, *dw2, *dyref;
    unsigned long long* ddbg;
    CK(hipMalloc(&dx, n * 2));
    CK(hipMalloc(&dy, n * 2));
    CK(hipMalloc(&dwp, 2 * pb));
    CK(hipMalloc(&dbias, bias.size() * 4));
    CK(hipMalloc(&dw1, w1.size() * 4));
    CK(hipMalloc(&dw2, w2.size() * 4));
    const int nwg_max = 1 << 20;
    CK(hipMalloc(&ddbg, (size_t)nwg_max * 16 * 8));
    CK(hipMemset(ddbg, 0, (size_t)nwg_max * 16 * 8));
    CK(hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwp, wp.data(), 2 * pb, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy, 0xff, n * 2));
    std::vector<unsigned short> hy0;
    if (acc_add) {  // the accumulator's previous content (last utterance only is checked; the others just need finite values)
        hy0.resize(n);
        for (size_t i = 0; i < n; ++i) hy0[i] = f2bf(0.7f * nd(rng));
        CK(hipMemcpy(dy, hy0.data(), n * 2, hipMemcpyHostToDevice));
    }

    BConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dx;
    a.wp = dwp;
    a.bias = dbias;
    a.y = dy;
    a.B = B;
    a.L = L;
    a.x_pitch = C;
    a.cin_real = C | ((dflags ? dflags : stagger) << 16);
    a.dil = dil;
    a.pad = (K - 1) / 2 * dil;
    a.slope_in = 0.1f;
    a.slope_out = 1.0f;
    a.acc_add = acc_add;
    a.div = 1.0f;
    a.dbg = nullptr;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(launch(a));
    CK(hipDeviceSynchronize());
    std::vector<float> times;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        CK(launch(a));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        times.push_back(ms);
    }
    std::sort(times.begin(), times.end());
    const double flops = 2.0 * 2.0 * (double)B * L * C * C * K;
    printf("time: min %.3f ms  median %.3f ms   -> %.1f TF/s (median), %.3f of 2500\n", times[0], times[times.size() / 2],
           flops / times[times.size() / 2] * 1e-9, flops / times[times.size() / 2] * 1e-9 / 2500.0);

    // ---- correctness on a window of the last utterance (start edge, an interior tile seam, end edge)
    if (acc_add) {  // the timed launches accumulated reps + 1 times: one launch on the original accumulator content
        CK(hipMemcpy(dy, hy0.data(), n * 2, hipMemcpyHostToDevice));
        CK(launch(a));
        CK(hipDeviceSynchronize());
    }
    {
        const int nwin = 3, wlen = 700;
        const int t_los[nwin] = {0, L / 2 - 350, L - wlen};
        CK(hipMalloc(&dxt, (size_t)L * C * 2));
        CK(hipMalloc(&dyref, (size_t)wlen * C * 4));
        const unsigned short* xb = dx + (size_t)(B - 1) * L * C;
        std::vector<float> yr((size_t)wlen * C);
        std::vector<unsigned short> yg((size_t)wlen * C);
        double worst = 0.0, ymax = 0.0;
        for (int wi = 0; wi < nwin; ++wi) {
            const int t_lo = t_los[wi];
            const int h = (K - 1) / 2;
            const int xt_lo = t_lo - h, xt_n = wlen + 2 * h;
            CK(hipMemset(dxt, 0, (size_t)L * C * 2));
            hipLaunchKernelGGL(ref_c1_k, dim3((xt_n * C + 255) / 256), dim3(256), 0, 0, xb, dw1, dbias, dxt, L, C, K, dil, xt_lo, xt_n);
            hipLaunchKernelGGL(ref_c2_k, dim3((wlen * C + 255) / 256), dim3(256), 0, 0, xb, dxt, dw2, dbias + C, dyref, L, C, K, t_lo, wlen);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(yr.data(), dyref, yr.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(yg.data(), dy + ((size_t)(B - 1) * L + t_lo) * C, yg.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < yr.size(); ++i) {
                if (acc_add) yr[i] += bf2f(hy0[((size_t)(B - 1) * L + t_lo) * C + i]);
                const double d = fabs((double)bf2f(yg[i]) - (double)yr[i]);
                // allowed: bf16 rounding of the output (2^-9 rel) + accumulation-order noise
                const double tol = fabs(yr[i]) * (1.0 / 256) + 2e-2;
                if (d / tol > worst) worst = d / tol;
                ymax = std::max(ymax, fabs((double)yr[i]));
            }
        }
        printf("check: worst err/tol = %.3f (|y|max %.2f) -> %s\n", worst, ymax, worst <= 1.0 ? "OK" : "MISMATCH");
    }

#if VTTS_TIMELINE
    {
        a.dbg = ddbg;
        CK(launch(a));
        CK(hipDeviceSynchronize());
        // number of workgroups: find the last non-zero record
        std::vector<unsigned long long> h((size_t)nwg_max * 16);
        CK(hipMemcpy(h.data(), ddbg, h.size() * 8, hipMemcpyDeviceToHost));
        int nwg = 0;
        for (int i = 0; i < nwg_max; ++i)
            if (h[(size_t)i * 16 + 6]) nwg = i + 1;
        auto seg = [&](const char* nm, int from, int to) {
            double sm = 0;
            int cnt = 0;
            for (int i = 0; i < nwg; ++i)
                if (h[(size_t)i * 16 + to] && h[(size_t)i * 16 + from]) {  // edge tiles skip some stamps
                    sm += (double)(h[(size_t)i * 16 + to] - h[(size_t)i * 16 + from]);
                    ++cnt;
                }
            printf("  %-34s %9.0f\n", nm, cnt ? sm / cnt : 0.0);
        };
        if (impl == 1) {  // persistent kernel: per-workgroup sums over its run of tiles
            double c1 = 0, e1 = 0, c2 = 0, nt = 0, tot = 0;
            for (int i = 0; i < nwg; ++i) { c1 += h[(size_t)i * 16 + 0]; e1 += h[(size_t)i * 16 + 1]; c2 += h[(size_t)i * 16 + 2]; nt += h[(size_t)i * 16 + 3]; tot += h[(size_t)i * 16 + 4]; }
            printf("persistent kernel, %d workgroups, %.1f tiles each: per tile  c1 loop %.0f  barrier+ep1+barrier %.0f  c2 loop %.0f  = %.0f ticks (loop total / tiles %.0f)\n",
                   nwg, nt / nwg, c1 / nt, e1 / nt, c2 / nt, (c1 + e1 + c2) / nt, tot / nt);
            return 0;
        }
        printf("timeline over %d workgroups (shader-clock ticks, mean per workgroup; thread 0's view):\n", nwg);
        if (impl == 20 || impl >= 30) {
            seg("stage_x + barrier", 0, 1);
            seg("   issue loads", 0, 7);
            seg("   wait vmcnt(0)", 7, 8);
            seg("   lrelu + ds_write", 8, 9);
            seg("   init_acc + barrier", 9, 1);
            seg("c1 main loop", 1, 2);
            seg("B2 + epilogue 1 + B3", 2, 3);
            seg("c2 main loop", 3, 4);
            seg("epilogue 2", 4, 6);
            seg("   issue residual loads", 4, 10);
            seg("   wait vmcnt(0)", 10, 11);
            seg("   add", 11, 12);
            seg("   pack, swap, store", 12, 6);
            double sp = 0, st = 0;
            if (impl == 20)
            for (int i = 0; i < nwg; ++i) { sp += (double)h[(size_t)i * 16 + 11]; st += (double)h[(size_t)i * 16 + 12]; }
            printf("  wave 0: failed ready-polls per workgroup %.1f, ticks spent spinning %.0f\n", sp / nwg, st / nwg);
        } else if (impl == 0) {
            seg("stage_x + slab 0 + barrier", 0, 1);
            seg("c1 main loop", 1, 2);
            seg("epilogue 1 + barrier", 2, 3);
            seg("c2 main loop", 3, 4);
            seg("ep2: acc -> LDS + barrier", 4, 5);
            seg("ep2: residual, store", 5, 6);
        } else {
            seg("stage_x: issue loads", 0, 7);
            seg("stage_x: wait vmcnt(0)", 7, 8);
            seg("stage_x: lrelu + ds_write", 8, 9);
            seg("stage_x: barrier", 9, 1);
            seg("c1 main loop", 1, 2);
            seg("epilogue 1 (+ issue residual)", 2, 10);
            seg("ep1 barrier (vmcnt(0))", 10, 3);
            seg("c2 main loop", 3, 4);
            seg("epilogue 2", 4, 6);
        }
        seg("total", 0, 6);
        // per-CU occupancy: group by (xcc, se, sh, cu)
        std::map<unsigned, std::vector<std::pair<unsigned long long, unsigned long long>>> percu;
        for (int i = 0; i < nwg; ++i) {
            const unsigned long long id = h[(size_t)i * 16 + 15];
            const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            percu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back({h[(size_t)i * 16 + 0], h[(size_t)i * 16 + 6]});
        }
        {  // wave-slot census of thread 0's wave: first generation vs the rest
            int first[16] = {0}, rest[16] = {0};
            for (int i = 0; i < nwg; ++i) ((i < 512) ? first : rest)[h[(size_t)i * 16 + 15] & 0xf]++;
            printf("HW_ID.wave_id of wave 0, workgroups 0..511:");
            for (int i = 0; i < 16; ++i) if (first[i]) printf("  slot%d:%d", i, first[i]);
            printf("   later:");
            for (int i = 0; i < 16; ++i) if (rest[i]) printf("  slot%d:%d", i, rest[i]);
            printf("\n");
        }
        {  // one CU's workgroups in start order: phase stamps relative to the kernel's first stamp (lockstep or not?)
            unsigned long long g0 = ~0ull;
            for (int i = 0; i < nwg; ++i) g0 = std::min(g0, h[(size_t)i * 16 + 0]);
            std::vector<int> ids;
            const unsigned key0 = percu.begin()->first;
            for (int i = 0; i < nwg; ++i) {
                const unsigned long long id = h[(size_t)i * 16 + 15];
                const unsigned hw = (unsigned)id, xcc = (unsigned)(id >> 32) & 0xf;
                const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                if (((xcc << 12) | (se << 8) | (sh << 4) | cu) == key0) ids.push_back(i);
            }
            std::sort(ids.begin(), ids.end(), [&](int x, int y) { return h[(size_t)x * 16] < h[(size_t)y * 16]; });
            printf("one CU, workgroups in start order (ticks since kernel start): wg slot | start X-staged c1-done xt-written c2-done end\n");
            for (size_t k = 0; k < ids.size() && k < 14; ++k) {
                const int i = ids[k];
                printf("  %6d s%llu |", i, h[(size_t)i * 16 + 15] & 0xf);
                const int st[6] = {0, 1, 2, 3, 4, 6};
                for (int q = 0; q < 6; ++q) printf(" %8llu", h[(size_t)i * 16 + st[q]] - g0);
                printf("\n");
            }
        }
        double busy = 0, span = 0, gaps = 0;
        long ngaps = 0;
        unsigned long long gmin = ~0ull, gmax = 0;
        for (auto& kv : percu) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end());
            for (size_t i = 0; i < v.size(); ++i) {
                busy += (double)(v[i].second - v[i].first);
                if (i) {
                    gaps += (double)v[i].first - (double)v[i - 1].second;
                    ngaps++;
                }
                gmin = std::min(gmin, v[i].first);
                gmax = std::max(gmax, v[i].second);
            }
            span += (double)(v.back().second - v.front().first);
        }
        printf("CUs seen: %zu; per-CU busy/span = %.3f; mean gap between consecutive workgroups on a CU = %.0f ticks; kernel span %llu ticks\n",
               percu.size(), busy / span, ngaps ? gaps / ngaps : 0.0, gmax - gmin);
    }
#endif
    return 0;
}
