// Fused ResBlock1 pair in bf16, second generation:   x' = c2(lrelu(c1(lrelu(x)))) + x   (vietTTS/hifigan/model.py:45-50)
//
// Same dataflow as kernels_bf16_pair.hip (xt = lrelu(c1(lrelu(x))) never leaves the CU; both convolutions share one
// stream of LDS-DMA'd weight slabs), re-tiled after the per-workgroup timeline of profiles/r01_e_*: with one
// 8-wave workgroup per CU, 28-60 % of a workgroup's cycles were its own un-overlapped tile staging and epilogues.
//   * 4-wave workgroups sized to <= 80 KiB of LDS, so that TWO are resident per CU (2 waves per SIMD, 256 VGPRs
//     each): one workgroup's staging / epilogue phases run beside the other's MFMA phase.
//       C = 128: 128 x 128 tile (2 x 2 waves of 64 x 64), C = 64: 64 x 256 (1 x 4 waves of 64 x 64),
//       C = 32: 32 x 512 (1 x 4 waves of 32 x 128).
//   * the X tile holds exactly the rows the layer's rate needs (N1 + (K-1)*rate), not the worst case.
//   * both epilogues work from the MFMA accumulator layout, no fp32 LDS transpose: a lane owns, for one time
//     step, 4 x 4 consecutive channels; one v_permlane32_swap per packed dword pair turns that into 2 x 8
//     consecutive channels = 16-byte LDS / global accesses (32 contiguous bytes per row per wave-instruction).
//   * the residual rows (raw x, L2-resident: this workgroup staged them a few microseconds earlier) are
//     requested before the c2 main loop and consumed after it.
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

namespace vtts {

template <int C_, int KS_, int N1_, int WM_, int WN_, int CKC_, int TG_, int MINWG_, int PIPE_ = 0, int NBUF_ = 2>
struct RTile {
    static constexpr int PIPE = PIPE_ % 10;             // 0 compiler's order, 1 reads one step ahead, 2 1:1 interleave, 3 reads TWO steps ahead
    static constexpr int PD = (PIPE_ % 10 >= 3) ? 2 : 1; // fragment prefetch distance in k-steps
    static constexpr int NFB = (PIPE_ % 10 >= 3) ? 4 : 2; // fragment register buffers
    static constexpr int ABL = PIPE_ / 10;  // kbench timing ablations (wrong results): 1 no ring barrier, 2 no DMA, 3 no B reads, 4 no A reads, 5 no MFMA
    static constexpr int C = C_, KS = KS_, N1 = N1_, WM = WM_, WN = WN_, CKC = CKC_, TG = TG_, MINWG = MINWG_;
    static constexpr int THREADS = 64 * WM * WN;
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
    static constexpr int SLAB_UNITS = SLAB_BYTES / 16;
    static constexpr int APT = (SLAB_UNITS + THREADS - 1) / THREADS;
    static constexpr int XPT = (ROWSX_MAX * SPR + THREADS - 1) / THREADS;
    static constexpr int NBUF = NBUF_;                  // weight-slab ring: NBUF - 1 slabs in flight
    static_assert(NBUF >= 2 && ((CKC / 16) % 2 == 0), "ring");
    static_assert(C % (WM * 32) == 0 && N1 % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(C % CKC == 0 && CKC % 16 == 0, "channel tiling");
    static_assert(SLAB_UNITS % 64 == 0, "slab = whole wave-instructions of LDS-DMA");
    static_assert(SPR == 4 || SPR == 8 || SPR == 16, "row pitch 64/128/256 B");
    static int lds_bytes(int dil) {
        const int rowsx = N1 + 2 * H2 * dil;
        const int ra = (rowsx > ROWST ? rowsx : ROWST) * P;
        return NBUF * SLAB_BYTES + ra;
    }
    static_assert(NBUF * SLAB_BYTES + ROWSX_MAX * P <= 160 * 1024, "LDS budget");
};

template <class T>
__global__ __launch_bounds__(T::THREADS, (T::MINWG * T::THREADS + 255) / 256) void resblock_pair2_bf16_k(BConvArgs a) {
    constexpr int C = T::C, CKC = T::CKC, KS = T::KS, N1 = T::N1, WN = T::WN, TG = T::TG;
    constexpr int THREADS = T::THREADS, MR = T::MR, NR = T::NR, H2 = T::H2, NT2 = T::NT2;
    constexpr int NCK = T::NCK, SPR = T::SPR, P = T::P;
    constexpr int KSTEPS = T::KSTEPS, NSL = T::NSL, NSTOT = T::NSTOT, MB = T::MB, APT = T::APT, XPT = T::XPT;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* ab = lds;                              // weight-slab ring
    unsigned char* xt = lds + T::NBUF * T::SLAB_BYTES;    // X tile, later the xt tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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

    const uint4* __restrict__ wsl = reinterpret_cast<const uint4*>(a.wp);
    const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + (size_t)b * L * C;
    [[maybe_unused]] const int wg_lin = blockIdx.z * gridDim.x + blockIdx.x;
    VTTS_TL_ID(a, wg_lin);

#if VTTS_TIMELINE
    const int dflags = a.cin_real >> 16;  // kbench experiments: 1 no X loads, 2 no residual loads, 4 no stores, 8 no slab DMA
#else
    constexpr int dflags = 0;
#endif

    // Two workgroups share a CU and, started together, stay in lockstep (both in their MFMA phase, then both in
    // their staging / epilogue phase).  The workgroup that finds its first wave in a non-zero wave slot of its SIMD
    // is the CU's second one: in the launch's first generation it starts `a.x_pitch` shader cycles late, and every
    // later workgroup inherits the offset from the one it replaces.
    if (T::MINWG > 1 && a.x_pitch > 0 && wg_lin < 2 * 256) {
        unsigned* flag = reinterpret_cast<unsigned*>(xt);
        if (tid == 0) *flag = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 4) /* HW_ID.wave_id */;
        __syncthreads();
        const unsigned slot = *flag;
        __syncthreads();
        if (slot != 0) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime() + (unsigned long long)a.x_pitch;
            while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(32);
        }
    }

    VTTS_TL(a, wg_lin, 0);
    // Accumulators start from the bias (row = channel 32*mr + 8*rq + 4*lh + i of this wave's m-block, r = 4*rq + i):
    // the bias rows are requested a phase ahead of the MFMAs that consume them.
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

    // weight slab s -> ring buffer `buf` by LDS-DMA (destination = wave-uniform base + lane*16 = the fragment-ordered image)
    // one 1 KiB piece (this wave's i-th) of slab s
    auto issue_piece = [&](int s, int i) {
        const int u0 = wave * 64 + i * THREADS;
        if (APT * THREADS == T::SLAB_UNITS || u0 < T::SLAB_UNITS)
            glds16_asm(wsl + (size_t)s * T::SLAB_UNITS + u0 + lane,
                       __builtin_amdgcn_readfirstlane(lds_addr_of(ab + (s % T::NBUF) * T::SLAB_BYTES + (size_t)u0 * 16)));
    };
    auto issue_slab = [&](int s) {
        const uint4* src = wsl + (size_t)s * T::SLAB_UNITS;
        unsigned char* dst = ab + (s % T::NBUF) * T::SLAB_BYTES + (size_t)(wave * 64) * 16;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int u0 = wave * 64 + i * THREADS;
            if ((APT * THREADS == T::SLAB_UNITS || u0 < T::SLAB_UNITS) && !((dflags & 8) && s >= 2)) {
                if constexpr (T::PIPE >= 3)
                    glds16_asm(src + u0 + lane, __builtin_amdgcn_readfirstlane(lds_addr_of(dst + (size_t)i * THREADS * 16)));
                else
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + u0 + lane), (lds_ptr_t)(dst + (size_t)i * THREADS * 16), 16, 0, 0);
            }
        }
    };

    // ---------------- X tile: LeakyReLU + zero padding in registers, swizzled ds_write_b128 ----------------
#pragma unroll
    for (int i = 0; i < T::NBUF - 1; ++i)
        if (i < NSTOT) issue_slab(i);
    {
        uint4 v[XPT];
        const int nunits = rowsx * SPR;
        const int tx0 = t0 - H2 - h1;
        // unconditional loads from clamped addresses, masked afterwards: a load under a per-element branch makes
        // hipcc wait for each one before issuing the next
        bool okx[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * THREADS;
            const int row = u / SPR, c = u % SPR;
            const int t = tx0 + row;
            okx[i] = u < nunits && t >= 0 && t < L;
            const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
            v[i] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + c * 8);
        }
        if (dflags & 1) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) okx[i] = false;
        }
#if VTTS_TIMELINE
        VTTS_TL(a, wg_lin, 7);
        if (a.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VTTS_TL(a, wg_lin, 8);
#endif
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
            const int u = tid + i * THREADS;
            const int row = u / SPR, c = u % SPR;
            if (u < nunits) *reinterpret_cast<uint4*>(xt + row * P + ((c ^ swz_of<SPR>(row)) << 4)) = v[i];
        }
    }
    init_acc();
    VTTS_TL(a, wg_lin, 9);
    if constexpr (T::PIPE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the asm LDS-DMAs of the first slabs
    __syncthreads();
    VTTS_TL(a, wg_lin, 1);

    // ---- main-loop machinery -------------------------------------------------------------------------------------
    // Weight slabs stream through a ring of NBUF LDS buffers, D = NBUF - 1 slabs ahead of the MFMAs (an LDS-DMA takes
    // about a microsecond from issue to landed, several slabs' worth of MFMAs).  Per slab: issue the DMA of slab s+D
    // into the buffer slab s-1 was read from, run the slab's MFMAs, then a counted s_waitcnt vmcnt + raw s_barrier
    // that makes slab s+2 visible — one slab more than the next iteration needs, so that the fragment pipeline
    // (ds_reads of step q+1 issued ahead of the MFMAs of step q) runs THROUGH the barrier into the next slab.
    constexpr int NBUF = T::NBUF, D = NBUF - 1, NS1 = T::NS1;
    constexpr bool AHEAD = D >= 2;  // with one slab in flight the next slab is not visible before the barrier: no look-ahead
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
    constexpr int PD = T::PD, NFB = T::NFB;
    bf16x8 af[NFB][MR], bf[NFB][NR];
    auto load_frags = [&](const SlabPos& sp, int q, int dl, int par) {
        const int tj = q / KSTEPS, ks = q % KSTEPS;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int row = sp.row0 + tj * dl + nr * 32;
            if (T::ABL != 3) bf[par][nr] = *reinterpret_cast<const bf16x8*>(xt + row * P + (((sp.slot0 + ks * 2) ^ swz_of<SPR>(row)) << 4));
            else bf[par][nr] = bf[(par + NFB - 1) % NFB][nr];
        }
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            if (T::ABL != 4) af[par][mr] = *reinterpret_cast<const bf16x8*>(sp.abuf + (q * MB + mr) * 1024);
            else af[par][mr] = af[(par + NFB - 1) % NFB][mr];
        }
    };
    // MFMAs of one slab (NTAPS taps x KSTEPS k-steps); the last step's look-ahead reads the first fragments of `nxt`
    // PIPE 4: the slab's own share of the weight stream (slab sg_issue, < 0 = none) is issued INSIDE the MFMA sequence,
    // the two waves of a SIMD (w and w + NCW/2) in different halves of it: issued in one burst behind the barrier, the
    // 1 KiB LDS-DMAs of all waves held every MFMA pipe idle for their whole issue time (measured).
    const int q_issue0 = (wave >= (T::WM * WN) / 2) ? 1 : 0;
    auto mma_slab = [&](const SlabPos& cur, const SlabPos& nxt, auto ntaps_tag, int dl, int sg_issue) {
        constexpr int NQ = decltype(ntaps_tag)::value * KSTEPS;
        static_assert(!AHEAD || NQ % NFB == 0, "fragment buffer index must return to 0 at a slab boundary");
        static_assert(NQ >= PD, "slab shorter than the prefetch distance");
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if constexpr (T::PIPE == 4) {
#pragma unroll
                for (int i = 0; i < APT; ++i)
                    if (sg_issue >= 0 && q == (i < NQ / 2 ? i : NQ / 2 - 1) + q_issue0 * (NQ / 2)) issue_piece(sg_issue, i);
            }
            if (q + PD < NQ) load_frags(cur, q + PD, dl, (q + PD) % NFB);
            else if (AHEAD) load_frags(nxt, q + PD - NQ, dl, (q + PD) % NFB);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    if (T::ABL != 5) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q % NFB][mr], bf[q % NFB][nr], acc[mr][nr], 0, 0, 0);
                    else asm volatile("" ::"v"(af[q % NFB][mr]), "v"(bf[q % NFB][nr]));
                }
            if constexpr (T::PIPE == 1 || T::PIPE == 3 || T::PIPE == 4) {
                if (q + PD < NQ || AHEAD) __builtin_amdgcn_sched_group_barrier(0x100, MR + NR, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MR * NR, 0);
            } else if constexpr (T::PIPE == 2) {
#pragma unroll
                for (int i = 0; i < MR * NR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < MR + NR && (q + 1 < NQ || AHEAD)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
    };
    // end of slab sg: every LDS-DMA except the `groups` most recent slabs has landed for this wave; then the barrier
    // s_waitcnt with only the vmcnt field set (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]);
    // the builtin (unlike inline asm) keeps hipcc's own lgkmcnt bookkeeping exact across the wait
    auto ring_barrier = [&](int groups) {
        // drain this wave's look-ahead ds_reads here, under the slab's last MFMAs, so that the loop carries no
        // outstanding LDS read: hipcc otherwise starts every iteration with a conservative s_waitcnt lgkmcnt(0)
        // placed right AFTER the first reads of the new slab
#define VTTS_VMCNT_IMM(n) (((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))
        if (groups <= 0) __builtin_amdgcn_s_waitcnt(VTTS_VMCNT_IMM(0));
        else if (groups == 1) __builtin_amdgcn_s_waitcnt(VTTS_VMCNT_IMM(APT < 63 ? APT : 63));
        else if (groups == 2) __builtin_amdgcn_s_waitcnt(VTTS_VMCNT_IMM(2 * APT < 63 ? 2 * APT : 63));
        else __builtin_amdgcn_s_waitcnt(VTTS_VMCNT_IMM(3 * APT < 63 ? 3 * APT : 63));
#undef VTTS_VMCNT_IMM
        if (T::ABL != 1) __builtin_amdgcn_s_barrier();
    };
    auto run_phase = [&](int s_begin, int s_end, int dl) {
        SlabPos cur = slab_pos(s_begin, dl);
        if (AHEAD) {
#pragma unroll
            for (int q = 0; q < PD; ++q) load_frags(cur, q, dl, q);
        }
#pragma nounroll
        for (int sg = s_begin; sg < s_end; ++sg) {
            const int sg_issue = (sg + D < NSTOT && T::ABL != 2) ? sg + D : -1;
            if (T::PIPE != 4 && sg_issue >= 0) issue_slab(sg_issue);
            if (!AHEAD) {
#pragma unroll
                for (int q = 0; q < PD; ++q) load_frags(cur, q, dl, q);
            }
            const SlabPos nxt = (sg + 1 < s_end) ? slab_pos(sg + 1, dl) : cur;
            const int sp = sg < NS1 ? sg : sg - NS1;
            if (TAILT != 0 && (sp % NSL) == NSL - 1) mma_slab(cur, nxt, std::integral_constant<int, (TAILT ? TAILT : TG)>{}, dl, sg_issue);
            else mma_slab(cur, nxt, std::integral_constant<int, TG>{}, dl, sg_issue);
            // slabs sg+3 .. min(sg+D, NSTOT-1) may stay in flight
            int groups = NSTOT - 3 - sg;
            if (groups > D - 2) groups = D - 2;
            if (D < 2) groups = 0;
            if (sg + 1 < NSTOT) ring_barrier(groups);
            cur = nxt;
        }
    };

    // ---------------- phase 1: xt = c1(lrelu(x)); column n <-> xt time t0 - H2 + n; tap j reads X row n + j*dil ----------------
    run_phase(0, NS1, dil);
    VTTS_TL(a, wg_lin, 2);

    // A lane's accumulators for one 32x32 block: column (time) l31, rows (channels) 8*rq + 4*lh + i, r = 4*rq + i.
    // (lo, hi) of two packed dwords: after swapping across the wave halves, lh = 0 owns channels 16p .. 16p+7 and
    // lh = 1 owns 16p+8 .. 16p+15 of rq pair p, as [P'0 P'1 Q'0 Q'1].
    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };

    // ---------------- epilogue 1: bias, LeakyReLU(0.1), bf16, zero outside [0, L) -> xt tile in LDS ----------------
    // (all waves are past the barrier that ended the last c1 slab: the X tile is dead)
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
        for (int u = tid; u < 2 * H2 * SPR; u += THREADS) {
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
                const int row = wn * (N1 / WN) + nr * 32 + l31;
                const int t = t0 + row;
                const int ch = wm * (C / T::WM) + mr * 32 + 16 * p + 8 * lh;
                const int tc = t < L ? t : L - 1;  // rows past the end are never stored: any in-bounds address will do
                resv[mr][p][nr] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + ch);
            }
    VTTS_TL(a, wg_lin, 10);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's xt rows are in LDS; the weight ring stays in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    VTTS_TL(a, wg_lin, 3);

    // ---------------- phase 2: c2 over the xt tile (rate 1): column n <-> time t0 + n, tap j reads xt row n + j ----------------
    run_phase(NS1, NSTOT, 1);
    VTTS_TL(a, wg_lin, 4);

    // ---------------- epilogue 2: + bias + x [MRF accumulate / mean] [consumer's LeakyReLU] -> bf16, 16-byte stores ----------------
    {
        const float s_out = a.slope_out;
        const float dv = a.div;
        unsigned short* __restrict__ yg = static_cast<unsigned short*>(a.y) + (size_t)b * L * C;
        uint4 accv[MR][2][NR];
        const bool acc_add = a.acc_add != 0;
        if (acc_add) {  // MRF accumulator rows: all requests first, one wait
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
        }
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
                    const int r0 = 8 * p;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[mr][nr][r0 + e];
                    {  // x = xt + x  (model.py:50): un-swap the residual chunk into the accumulator layout
                        uint4 r = resv[mr][p][nr];
                        swap_pair(r.x, r.z);
                        swap_pair(r.y, r.w);
                        v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
                        v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
                    }
                    const size_t g = (size_t)t * C + cb + 8 * lh;
                    if (acc_add) {  // MRF  xs += rb(x)  (model.py:118-120)
                        uint4 o = accv[mr][p][nr];
                        swap_pair(o.x, o.z);
                        swap_pair(o.y, o.w);
                        v[0] = bf16_lo(o.x) + v[0]; v[1] = bf16_hi(o.x) + v[1]; v[2] = bf16_lo(o.y) + v[2]; v[3] = bf16_hi(o.y) + v[3];
                        v[4] = bf16_lo(o.z) + v[4]; v[5] = bf16_hi(o.z) + v[5]; v[6] = bf16_lo(o.w) + v[6]; v[7] = bf16_hi(o.w) + v[7];
                    }
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
                    if (ok && !(dflags & 4)) *reinterpret_cast<uint4*>(yg + g) = make_uint4(p0, p1, q0, q1);
                }
            }
        }
    }
    VTTS_TL(a, wg_lin, 5);
    VTTS_TL(a, wg_lin, 6);
}

// ---- tile table -------------------------------------------------------------------------------------
//                                      C   KS   N1  WM WN CKC TG            MINWG
template <int KS> using R128 = RTile<128, KS, 128, 2, 2, 64, 1, 2>;
template <int KS> using R64 = RTile<64, KS, 256, 1, 4, 64, 2, 2>;
template <int KS> using R32 = RTile<32, KS, 512, 1, 4, 32, KS, 2>;
// development variants (tools/kbench): the first-generation geometry with the new epilogues
template <int KS> using R128W = RTile<128, KS, 256, 2, 4, 128, 1, 1>;
template <int KS> using R64W = RTile<64, KS, 512, 1, 8, 64, (KS < 4 ? KS : 4), 1>;
template <int KS> using R32W = RTile<32, KS, 512, 1, 8, 32, KS, 2>;
template <int KS> using R128P1 = RTile<128, KS, 128, 2, 2, 64, 1, 2, 1>;
template <int KS> using R128WP1 = RTile<128, KS, 256, 2, 4, 128, 1, 1, 1>;
template <int KS> using R128W4 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 0, 4>;
template <int KS> using R128W5 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 0, 5>;
template <int KS> using R128W5P1 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 1, 5>;
template <int KS> using R128W5P2 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 2, 5>;
template <int KS> using R128W5P3 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 3, 5>;
template <int KS> using R128W4P3 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 3, 4>;
template <int KS> using R128W5P4 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 4, 5>;
template <int KS> using R128W4P4 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 4, 4>;
template <int KS> using R128W5A1 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 11, 5>;
template <int KS> using R128W5A2 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 21, 5>;
template <int KS> using R128W5A3 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 31, 5>;
template <int KS> using R128W5A4 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 41, 5>;
template <int KS> using R128W5A5 = RTile<128, KS, 256, 2, 4, 64, 1, 1, 51, 5>;

template <class T>
static hipError_t launch_r(const BConvArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair2_bf16_k<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           T::lds_bytes(T::MAXDIL));
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (a.dil < 1 || a.dil > T::MAXDIL) return hipErrorInvalidValue;
    dim3 grid((a.L + T::NT2 - 1) / T::NT2, 1, a.B);
    hipLaunchKernelGGL(resblock_pair2_bf16_k<T>, grid, dim3(T::THREADS), T::lds_bytes(a.dil), s, a);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_r_ks(const BConvArgs& a, int K, hipStream_t s) {
    switch (K) {
        case 3: return launch_r<TT<3>>(a, s);
        case 7: return launch_r<TT<7>>(a, s);
        case 11: return launch_r<TT<11>>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pair2_bf16(int C, int K, int variant, const BConvArgs& a, hipStream_t s) {
    if (variant == 0) switch (C) {
            case 128: return launch_r_ks<R128>(a, K, s);
            case 64: return launch_r_ks<R64>(a, K, s);
            case 32: return launch_r_ks<R32>(a, K, s);
        }
    if (variant == 1) switch (C) {
            case 128: return launch_r_ks<R128W>(a, K, s);
            case 64: return launch_r_ks<R64W>(a, K, s);
            case 32: return launch_r_ks<R32W>(a, K, s);
        }
    if (C == 128) switch (variant) {
            case 2: return launch_r_ks<R128P1>(a, K, s);
            case 3: return launch_r_ks<R128WP1>(a, K, s);
            case 4: return launch_r_ks<R128W4>(a, K, s);
            case 5: return launch_r_ks<R128W5>(a, K, s);
            case 6: return launch_r_ks<R128W5P1>(a, K, s);
            case 7: return launch_r_ks<R128W5P2>(a, K, s);
            case 8: return launch_r_ks<R128W5P3>(a, K, s);
            case 9: return launch_r_ks<R128W4P3>(a, K, s);
            case 16: return launch_r_ks<R128W5P4>(a, K, s);
            case 17: return launch_r_ks<R128W4P4>(a, K, s);
            case 11: return launch_r_ks<R128W5A1>(a, K, s);
            case 12: return launch_r_ks<R128W5A2>(a, K, s);
            case 13: return launch_r_ks<R128W5A3>(a, K, s);
            case 14: return launch_r_ks<R128W5A4>(a, K, s);
            case 15: return launch_r_ks<R128W5A5>(a, K, s);
        }
    return hipErrorInvalidValue;
}

BPackGeom pair2_pack_geom(int C, int K, int variant) {
    if (C == 128 && variant == 2) variant = 0;
    if (C == 128 && variant == 3) variant = 1;
    if (C == 128 && variant >= 4) return BPackGeom{128, 64, 128, K, 128, 1};
    if (variant == 0) switch (C) {
            case 128: return BPackGeom{128, 64, 128, K, 128, 1};
            case 64: return BPackGeom{64, 64, 64, K, 64, 2};
            case 32: return BPackGeom{32, 32, 32, K, 32, K};
        }
    if (variant == 1) switch (C) {
            case 128: return BPackGeom{128, 128, 128, K, 128, 1};
            case 64: return BPackGeom{64, 64, 64, K, 64, K < 4 ? K : 4};
            case 32: return BPackGeom{32, 32, 32, K, 32, K};
        }
    return BPackGeom{0, 0, 0, 0, 0, 0};
}

}  // namespace vtts
