// Fused ResBlock1 pair in bf16 with a dedicated weight-loader wave:   x' = c2(lrelu(c1(lrelu(x)))) + x
// (vietTTS/hifigan/model.py:45-50).
//
// Measured on MI355X (profiles/r01_e_*): in the barrier-per-slab kernels the MFMA main loops ran at 75 % of the
// matrix-pipe rate, and at 100 % with the per-slab s_barrier removed (a timing ablation) — every barrier drains the
// MFMA pipeline of all eight waves at once.  Here the main loops contain NO workgroup barrier:
//   * waves NCW .. NCW+NLW-1 only stream weight slabs L2 -> LDS by LDS-DMA into a ring of NBUF buffers and publishes
//     "slab p landed" by bumping ready[p % NBUF] after a counted s_waitcnt vmcnt (one wave cannot issue a 16 KiB slab
//     per slab period — measured — so NLW = 4 loader waves split every slab);
//   * the NCW consumer waves poll ready[] one slab ahead (a broadcast ds_read issued with the fragment reads, its
//     wait falls on an existing lgkmcnt wait), run the slab's MFMAs with a register-double-buffered fragment pipeline
//     that continues across slab boundaries, and bump done[s % NBUF] when their last ds_read of slab s is issued;
//   * the loader re-fills a buffer once done[] shows all consumers have left it.  LDS returns in order per CU, so
//     flag-after-data on the producer side and data-after-flag on the consumer side need no further fences.
// Waves therefore drift by up to NBUF - 1 slabs instead of meeting every 16 MFMAs.  s_barrier remains only at the
// three tile-level hand-offs (X tile staged / X tile dead / xt tile written).
//
// Tile: C output channels x N1 time steps per workgroup; xt = lrelu(c1(lrelu(x))) lives only in LDS (it overwrites
// the dead X tile); both epilogues work from the MFMA accumulator layout with v_permlane32_swap (16-byte accesses,
// no LDS transpose); accumulators start from the bias; the residual rows are requested before the c2 loop.
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

namespace vtts {

template <int C_, int KS_, int N1_, int WM_, int WN_, int CKC_, int TG_, int NBUF_, int MINWG_, int NLW_ = 4>
struct LTile {
    static constexpr int NLW = NLW_;                    // loader waves (each streams 1/NLW of every slab)
    static constexpr int C = C_, KS = KS_, N1 = N1_, WM = WM_, WN = WN_, CKC = CKC_, TG = TG_, NBUF = NBUF_, MINWG = MINWG_;
    static constexpr int NCW = WM * WN;                 // consumer (MFMA) waves
    static constexpr int CTHREADS = 64 * NCW, THREADS = CTHREADS + 64 * NLW;
    static constexpr int MR = C / WM / 32, NR = N1 / WN / 32;
    static constexpr int H2 = (KS - 1) / 2;             // c2 halo (rate 1); c1's is H2 * rate
    static constexpr int MAXDIL = 5;
    static constexpr int NT2 = N1 - 2 * H2;             // outputs per workgroup
    static constexpr int NCK = C / CKC;                 // weight-slab channel chunks
    static constexpr int SPR = C / 8, P = C * 2;        // 16-byte slots / bytes per tile row (X and xt tiles alike)
    static constexpr int ROWSX_MAX = N1 + 2 * H2 * MAXDIL;
    static constexpr int ROWST = N1 + 2 * H2;           // xt rows incl. the tail only discarded columns read
    static constexpr int KSTEPS = CKC / 16;
    static constexpr int NSL = (KS + TG - 1) / TG;
    static constexpr int NS1 = NCK * NSL, NSTOT = 2 * NS1;
    static constexpr int MB = C / 32;
    static constexpr int SLAB_BYTES = C * TG * CKC * 2;
    static constexpr int GPS = SLAB_BYTES / 1024;       // LDS-DMA wave-instructions (1 KiB each) per slab
    static constexpr int GPSL = GPS / NLW;              // ... per loader wave
    static constexpr int XPT = (ROWSX_MAX * SPR + CTHREADS - 1) / CTHREADS;
    static constexpr int SYNC_BYTES = 64;               // ready[8], done[8]
    static_assert(C % (WM * 32) == 0 && N1 % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(C % CKC == 0 && CKC % 32 == 0, "channel tiling (an even number of k-steps per tap)");
    static_assert(SLAB_BYTES % 1024 == 0 && GPS % NLW == 0 && NBUF >= 2 && NBUF <= 8 && (NBUF - 1) * GPSL <= 63, "ring geometry");
    static_assert(SPR == 4 || SPR == 8 || SPR == 16, "row pitch 64/128/256 B");
    static int lds_bytes(int dil) {
        const int rowsx = N1 + 2 * H2 * dil;
        return SYNC_BYTES + NBUF * SLAB_BYTES + (rowsx > ROWST ? rowsx : ROWST) * P;
    }
    static_assert(SYNC_BYTES + NBUF * SLAB_BYTES + ROWSX_MAX * P <= 160 * 1024, "LDS budget");
};

template <class T>
__global__ __launch_bounds__(T::THREADS, (T::MINWG * T::THREADS + 255) / 256) void resblock_pair_lw_bf16_k(BConvArgs a) {
    constexpr int C = T::C, CKC = T::CKC, KS = T::KS, N1 = T::N1, WN = T::WN, TG = T::TG, NBUF = T::NBUF, NCW = T::NCW;
    constexpr int CTHREADS = T::CTHREADS, MR = T::MR, NR = T::NR, H2 = T::H2, NT2 = T::NT2;
    constexpr int SPR = T::SPR, P = T::P, GPSL = T::GPSL, NLW = T::NLW;
    constexpr int KSTEPS = T::KSTEPS, NSL = T::NSL, NS1 = T::NS1, NSTOT = T::NSTOT, MB = T::MB, XPT = T::XPT;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    lds_u32* ready = (lds_u32*)lds;       // ready[b]: slabs published into ring buffer b so far
    lds_u32* done = (lds_u32*)lds + 8;    // done[b]: consumer-wave passes over ring buffer b so far
    unsigned char* ab = lds + T::SYNC_BYTES;                   // weight-slab ring
    unsigned char* xt = ab + NBUF * T::SLAB_BYTES;             // X tile, later the xt tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    if (tid < 16) reinterpret_cast<unsigned*>(lds)[tid] = 0u;
    __syncthreads();

    auto compiler_fence = []() { asm volatile("" ::: "memory"); };
    auto lds_load = [](lds_u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    // The loader's own flag traffic goes through inline asm: behind an LDS-DMA, hipcc orders every LDS access it can
    // see with s_waitcnt vmcnt(0), which would drain the ring at each poll / publish.
    auto lds_load_asm = [](lds_u32* p) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    };
    auto lds_store_asm = [](lds_u32* p, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(p), "v"(v) : "memory"); };

    // =============================================== loader waves ===============================================
    // Loader j streams pieces j, j + NLW, ... (1 KiB each) of every slab.  State machine: fill any free ring buffer
    // first; otherwise retire the oldest un-published slab (counted vmcnt, then ready[b] += 1); otherwise nap.
    if (wave >= NCW) {
        const int lw = wave - NCW;
        const uint4* __restrict__ wsl = reinterpret_cast<const uint4*>(a.wp);
        auto issue = [&](int s) {
            const uint4* src = wsl + (size_t)s * (T::SLAB_BYTES / 16) + lw * 64 + lane;
            unsigned char* dst = ab + (s % NBUF) * T::SLAB_BYTES + lw * 1024;
#pragma unroll
            for (int i = 0; i < GPSL; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + i * NLW * 64), (lds_ptr_t)(dst + i * NLW * 1024), 16, 0, 0);
        };
        auto wait_groups = [&](int younger) {  // all but the `younger` most recent slabs' pieces of this wave have landed
            switch (younger) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * T::GPSL) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * T::GPSL < 63 ? 2 * T::GPSL : 63) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * T::GPSL < 63 ? 3 * T::GPSL : 63) : "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * T::GPSL < 63 ? 4 * T::GPSL : 63) : "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * T::GPSL < 63 ? 5 * T::GPSL : 63) : "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * T::GPSL < 63 ? 6 * T::GPSL : 63) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * T::GPSL < 63 ? 7 * T::GPSL : 63) : "memory"); break;
            }
        };
        int issued = 0, published = 0;
        int barriers = 0;  // B1 (X staged), B2 (X dead), B3 (xt written): the loaders only have to arrive
#pragma unroll
        for (int s = 0; s < (NBUF < NSTOT ? NBUF : NSTOT); ++s) issue(s);
        issued = NBUF < NSTOT ? NBUF : NSTOT;
        __builtin_amdgcn_s_barrier();  // B1
        barriers = 1;
        while (published < NSTOT) {
            if (issued < NSTOT) {
                const unsigned need = (unsigned)(NCW * (issued / NBUF));
                if (lds_load_asm(&done[issued % NBUF]) >= need) {
                    compiler_fence();
                    issue(issued);
                    ++issued;
                    continue;
                }
            }
            if (published < issued) {
                wait_groups(issued - published - 1);
                if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(&ready[published % NBUF]), "v"(1u) : "memory");
                ++published;
                continue;
            }
            // nothing to fill, nothing in flight: the buffer wanted next is still being read
            if (barriers == 1 && published >= NS1 && issued - NBUF >= NS1) {  // ... by c2, which starts behind B2 / B3
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
                barriers = 3;
                continue;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        for (; barriers < 3; ++barriers) __builtin_amdgcn_s_barrier();
        return;
    }

    // ============================================== consumer waves ==============================================
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int t0 = blockIdx.x * NT2;         // first output time step of this workgroup
    const int b = blockIdx.z;
    const int L = a.L;
    const int dil = a.dil;
    const int h1 = H2 * dil;                 // c1's symmetric pad (model.py:8-10)
    const int rowsx = N1 + 2 * h1;           // X rows: times t0 - H2 - h1 ... t0 - H2 - h1 + rowsx - 1
    const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + (size_t)b * L * C;
    [[maybe_unused]] const int wg_lin = blockIdx.z * gridDim.x + blockIdx.x;
    VTTS_TL_ID(a, wg_lin);
    VTTS_TL(a, wg_lin, 0);

    // Accumulators start from the bias (row = channel 32*mr + 8*rq + 4*lh + i of this wave's m-block, r = 4*rq + i)
    f32x16 acc[MR][NR];
    float4 bq[MR][4];
    auto load_bias = [&](const float* __restrict__ bias) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) bq[mr][rq] = *reinterpret_cast<const float4*>(bias + wm * (C / T::WM) + mr * 32 + 8 * rq + 4 * lh);
    };
    auto init_acc = [&]() {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    acc[mr][nr][4 * rq + 0] = bq[mr][rq].x;
                    acc[mr][nr][4 * rq + 1] = bq[mr][rq].y;
                    acc[mr][nr][4 * rq + 2] = bq[mr][rq].z;
                    acc[mr][nr][4 * rq + 3] = bq[mr][rq].w;
                }
    };
    load_bias(a.bias);

    // ---------------- X tile: LeakyReLU + zero padding in registers, swizzled ds_write_b128 ----------------
    {
        uint4 v[XPT];
        bool okx[XPT];
        const int nunits = rowsx * SPR;
        const int tx0 = t0 - H2 - h1;
        // unconditional loads from clamped addresses, masked afterwards: a load under a per-element branch makes
        // hipcc wait for each one before issuing the next
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * CTHREADS;
            const int row = u / SPR, c = u % SPR;
            const int t = tx0 + row;
            okx[i] = u < nunits && t >= 0 && t < L;
            const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
            v[i] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + c * 8);
        }
        auto act2 = [](unsigned u) { return pack_bf16x2(lrelu01(bf16_lo(u)), lrelu01(bf16_hi(u))); };  // LRELU_SLOPE, model.py:5,46
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            if (!okx[i]) v[i] = make_uint4(0u, 0u, 0u, 0u);
            v[i].x = act2(v[i].x);
            v[i].y = act2(v[i].y);
            v[i].z = act2(v[i].z);
            v[i].w = act2(v[i].w);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * CTHREADS;
            const int row = u / SPR, c = u % SPR;
            if (u < nunits) *reinterpret_cast<uint4*>(xt + row * P + ((c ^ swz_of<SPR>(row)) << 4)) = v[i];
        }
    }
    init_acc();
    __syncthreads();  // B1
    VTTS_TL(a, wg_lin, 1);

    // ---- main-loop machinery -------------------------------------------------------------------------------------
    constexpr int TAILT = KS % TG;
    const int rowbase0 = wn * (N1 / WN) + l31;
    struct SlabPos {
        const unsigned char* abuf;  // this lane's A fragments of the slab
        int row0;                   // tile row of output column (this lane, nr = 0) at the slab's first tap
        int slot0;                  // 16-byte slot of the slab's first k-step in a tile row
    };
    auto slab_pos = [&](int sg, int dl) {
        const int sp = sg < NS1 ? sg : sg - NS1;
        const int ck = sp / NSL, sl = sp - ck * NSL;
        return SlabPos{ab + (sg % NBUF) * T::SLAB_BYTES + (size_t)(wm * MR) * 1024 + lane * 16, rowbase0 + sl * TG * dl, ck * (CKC / 8) + lh};
    };
    bf16x8 af[2][MR], bf[2][NR];
    auto load_frags = [&](const SlabPos& sp, int q, int dl, int par) {
        const int tj = q / KSTEPS, ks = q % KSTEPS;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int row = sp.row0 + tj * dl + nr * 32;
            bf[par][nr] = *reinterpret_cast<const bf16x8*>(xt + row * P + (((sp.slot0 + ks * 2) ^ swz_of<SPR>(row)) << 4));
        }
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) af[par][mr] = *reinterpret_cast<const bf16x8*>(sp.abuf + (q * MB + mr) * 1024);
    };
#if VTTS_TIMELINE
    unsigned long long spin_ticks = 0;
    unsigned spins = 0;
#endif
    auto wait_ready = [&](int sg, unsigned seen) {
        const unsigned need = (unsigned)(NLW * (sg / NBUF + 1));
#if VTTS_TIMELINE
        if (seen < need) {
            const unsigned long long t_in = __builtin_amdgcn_s_memtime();
            while (seen < need) {
                __builtin_amdgcn_s_sleep(1);
                seen = lds_load(&ready[sg % NBUF]);
                ++spins;
            }
            spin_ticks += __builtin_amdgcn_s_memtime() - t_in;
        }
#else
        while (seen < need) {
            __builtin_amdgcn_s_sleep(1);
            seen = lds_load(&ready[sg % NBUF]);
        }
#endif
        compiler_fence();
    };
    // MFMAs of slab sg (NTAPS taps x KSTEPS k-steps).  The ds_reads of step q+1 go out ahead of the MFMAs of step q;
    // the last step looks ahead into slab sg+1 (polled two steps earlier) when the phase has one.
    auto mma_slab = [&](const SlabPos& cur, const SlabPos& nxt, auto ntaps_tag, int sg, bool has_next, int dl) {
        constexpr int NQ = decltype(ntaps_tag)::value * KSTEPS;
        static_assert(NQ % 2 == 0, "fragment parity must return to 0 at a slab boundary");
        unsigned seen = 0u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q == NQ - 2 && has_next) seen = lds_load(&ready[(sg + 1) % NBUF]);
            if (q + 1 < NQ) {
                load_frags(cur, q + 1, dl, (q + 1) & 1);
            } else if (has_next) {
                wait_ready(sg + 1, seen);
                load_frags(nxt, 0, dl, 0);
            }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1][mr], bf[q & 1][nr], acc[mr][nr], 0, 0, 0);
            if (q + 1 < NQ) __builtin_amdgcn_sched_group_barrier(0x100, MR + NR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MR * NR, 0);
        }
        // every ds_read of slab sg by this wave is in the LDS queue: hand the buffer back
        compiler_fence();
        if (lane == 0) __hip_atomic_fetch_add(&done[sg % NBUF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        compiler_fence();
    };
    auto run_phase = [&](int s_begin, int s_end, int dl) {
        wait_ready(s_begin, 0u);
        SlabPos cur = slab_pos(s_begin, dl);
        load_frags(cur, 0, dl, 0);
#pragma nounroll
        for (int sg = s_begin; sg < s_end; ++sg) {
            const bool has_next = sg + 1 < s_end;
            const SlabPos nxt = has_next ? slab_pos(sg + 1, dl) : cur;
            const int sp = sg < NS1 ? sg : sg - NS1;
            if (TAILT != 0 && (sp % NSL) == NSL - 1) mma_slab(cur, nxt, std::integral_constant<int, (TAILT ? TAILT : TG)>{}, sg, has_next, dl);
            else mma_slab(cur, nxt, std::integral_constant<int, TG>{}, sg, has_next, dl);
            cur = nxt;
        }
    };

    // ---------------- phase 1: xt = c1(lrelu(x)); column n <-> xt time t0 - H2 + n; tap j reads X row n + j*dil ----------------
    run_phase(0, NS1, dil);
    VTTS_TL(a, wg_lin, 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // B2: every wave is done reading the X tile
    compiler_fence();

    // A lane's accumulators for one 32x32 block: column (time) l31, rows (channels) 8*rq + 4*lh + i, r = 4*rq + i.
    // (lo, hi) of two packed dwords: after swapping across the wave halves, lh = 0 owns channels 16p .. 16p+7 and
    // lh = 1 owns 16p+8 .. 16p+15 of rq pair p, as [P'0 P'1 Q'0 Q'1].
    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };

    // ---------------- epilogue 1: LeakyReLU(0.1), bf16, zero outside [0, L) -> xt tile in LDS ----------------
    load_bias(a.bias + C);  // c2's bias: lands while epilogue 1 runs
    {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cb = wm * (C / T::WM) + mr * 32 + 16 * p;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int row = wn * (N1 / WN) + nr * 32 + l31;
                    const int tt = t0 - H2 + row;
                    const bool ok = tt >= 0 && tt < L;
                    const int r0 = 8 * p;
                    unsigned p0 = pack_bf16x2(lrelu01(acc[mr][nr][r0 + 0]), lrelu01(acc[mr][nr][r0 + 1]));
                    unsigned p1 = pack_bf16x2(lrelu01(acc[mr][nr][r0 + 2]), lrelu01(acc[mr][nr][r0 + 3]));
                    unsigned q0 = pack_bf16x2(lrelu01(acc[mr][nr][r0 + 4]), lrelu01(acc[mr][nr][r0 + 5]));
                    unsigned q1 = pack_bf16x2(lrelu01(acc[mr][nr][r0 + 6]), lrelu01(acc[mr][nr][r0 + 7]));
                    if (!ok) p0 = p1 = q0 = q1 = 0u;  // c2's own zero padding applies to xt
                    swap_pair(p0, q0);
                    swap_pair(p1, q1);
                    const int slot = (cb >> 3) + lh;
                    *reinterpret_cast<uint4*>(xt + row * P + ((slot ^ swz_of<SPR>(row)) << 4)) = make_uint4(p0, p1, q0, q1);
                }
            }
        }
        // rows N1 .. N1 + 2*H2 - 1 are only read by the discarded output columns: keep them finite
        for (int u = tid; u < 2 * H2 * SPR; u += CTHREADS) {
            const int row = N1 + u / SPR, c = u % SPR;
            *reinterpret_cast<uint4*>(xt + row * P + (c << 4)) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    init_acc();

    // residual rows (raw x) in the swapped accumulator layout, requested now, used after the c2 main loop
    uint4 resv[MR][2][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = t0 + wn * (N1 / WN) + nr * 32 + l31;
                const int tc = t < L ? t : L - 1;  // rows past the end are never stored: any in-bounds address will do
                resv[mr][p][nr] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + wm * (C / T::WM) + mr * 32 + 16 * p + 8 * lh);
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's xt rows are in LDS
    __builtin_amdgcn_s_barrier();                        // B3
    compiler_fence();
    VTTS_TL(a, wg_lin, 3);

    // ---------------- phase 2: c2 over the xt tile (rate 1): column n <-> time t0 + n, tap j reads xt row n + j ----------------
    run_phase(NS1, NSTOT, 1);
    VTTS_TL(a, wg_lin, 4);

    // ---------------- epilogue 2: + x [MRF accumulate / mean] [consumer's LeakyReLU] -> bf16, 16-byte stores ----------------
    {
        const float s_out = a.slope_out;
        const float dv = a.div;
        unsigned short* __restrict__ yg = static_cast<unsigned short*>(a.y) + (size_t)b * L * C;
        // pass A: x = xt + x  (model.py:50) — un-swap each residual chunk into the accumulator layout and add it in place
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    uint4 r = resv[mr][p][nr];
                    swap_pair(r.x, r.z);
                    swap_pair(r.y, r.w);
                    const int r0 = 8 * p;
                    acc[mr][nr][r0 + 0] += bf16_lo(r.x); acc[mr][nr][r0 + 1] += bf16_hi(r.x);
                    acc[mr][nr][r0 + 2] += bf16_lo(r.y); acc[mr][nr][r0 + 3] += bf16_hi(r.y);
                    acc[mr][nr][r0 + 4] += bf16_lo(r.z); acc[mr][nr][r0 + 5] += bf16_hi(r.z);
                    acc[mr][nr][r0 + 6] += bf16_lo(r.w); acc[mr][nr][r0 + 7] += bf16_hi(r.w);
                }
        if (a.acc_add != 0) {  // MRF  xs += rb(x)  (model.py:118-120): all requests first, one wait (the residual registers are free now)
            uint4 accv[MR][2][NR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int t = t0 + wn * (N1 / WN) + nr * 32 + l31;
                        const int tc = t < L ? t : L - 1;
                        accv[mr][p][nr] = *reinterpret_cast<const uint4*>(yg + (size_t)tc * C + wm * (C / T::WM) + mr * 32 + 16 * p + 8 * lh);
                    }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        uint4 o = accv[mr][p][nr];
                        swap_pair(o.x, o.z);
                        swap_pair(o.y, o.w);
                        const int r0 = 8 * p;
                        acc[mr][nr][r0 + 0] = bf16_lo(o.x) + acc[mr][nr][r0 + 0]; acc[mr][nr][r0 + 1] = bf16_hi(o.x) + acc[mr][nr][r0 + 1];
                        acc[mr][nr][r0 + 2] = bf16_lo(o.y) + acc[mr][nr][r0 + 2]; acc[mr][nr][r0 + 3] = bf16_hi(o.y) + acc[mr][nr][r0 + 3];
                        acc[mr][nr][r0 + 4] = bf16_lo(o.z) + acc[mr][nr][r0 + 4]; acc[mr][nr][r0 + 5] = bf16_hi(o.z) + acc[mr][nr][r0 + 5];
                        acc[mr][nr][r0 + 6] = bf16_lo(o.w) + acc[mr][nr][r0 + 6]; acc[mr][nr][r0 + 7] = bf16_hi(o.w) + acc[mr][nr][r0 + 7];
                    }
        }
        // pass B: mean / consumer's activation / bf16 / swap into 8 consecutive channels per lane / store
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cb = wm * (C / T::WM) + mr * 32 + 16 * p;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int row = wn * (N1 / WN) + nr * 32 + l31;
                    const int t = t0 + row;
                    const bool ok = row < NT2 && t < L;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[mr][nr][8 * p + e];
                    if (dv != 1.0f) {  // x = xs / num_kernels  (model.py:121)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] / dv;
                    }
                    if (s_out != 1.0f) {  // the (only) consumer's LeakyReLU, applied once by the producer
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = lrelu_f(v[e], s_out);
                    }
                    unsigned p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
                    unsigned q0 = pack_bf16x2(v[4], v[5]), q1 = pack_bf16x2(v[6], v[7]);
                    swap_pair(p0, q0);
                    swap_pair(p1, q1);
                    if (ok) *reinterpret_cast<uint4*>(yg + (size_t)t * C + cb + 8 * lh) = make_uint4(p0, p1, q0, q1);
                }
            }
        }
    }
    VTTS_TL(a, wg_lin, 6);
#if VTTS_TIMELINE
    if (a.dbg && tid == 0) {
        a.dbg[(size_t)wg_lin * 16 + 11] = spins;
        a.dbg[(size_t)wg_lin * 16 + 12] = spin_ticks;
    }
#endif
}

// ---- tile table -------------------------------------------------------------------------------------
//                                       C   KS   N1  WM WN CKC TG NBUF MINWG
template <int KS> using L128 = LTile<128, KS, 256, 2, 4, 64, 1, 5, 1, 4>;
template <int KS> using L64 = LTile<64, KS, 512, 1, 8, 64, 2, 5, 1, 4>;

template <class T>
static hipError_t launch_l(const BConvArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_lw_bf16_k<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           T::lds_bytes(T::MAXDIL));
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (a.dil < 1 || a.dil > T::MAXDIL) return hipErrorInvalidValue;
    dim3 grid((a.L + T::NT2 - 1) / T::NT2, 1, a.B);
    hipLaunchKernelGGL(resblock_pair_lw_bf16_k<T>, grid, dim3(T::THREADS), T::lds_bytes(a.dil), s, a);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_l_ks(const BConvArgs& a, int K, hipStream_t s) {
    switch (K) {
        case 3: return launch_l<TT<3>>(a, s);
        case 7: return launch_l<TT<7>>(a, s);
        case 11: return launch_l<TT<11>>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pair_lw_bf16(int C, int K, const BConvArgs& a, hipStream_t s) {
    switch (C) {
        case 128: return launch_l_ks<L128>(a, K, s);
        case 64: return launch_l_ks<L64>(a, K, s);
    }
    return hipErrorInvalidValue;
}

BPackGeom pair_lw_pack_geom(int C, int K) {
    switch (C) {
        case 128: return BPackGeom{128, 64, 128, K, 128, 1};
        case 64: return BPackGeom{64, 64, 64, K, 64, 2};
    }
    return BPackGeom{0, 0, 0, 0, 0, 0};
}

}  // namespace vtts
