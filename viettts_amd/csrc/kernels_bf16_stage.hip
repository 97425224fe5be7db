// The generator's LAST stage as one kernel, bf16, C = 32 (round 6):
//     x0 = ups_3 output window                                                   (vietTTS/hifigan/model.py:112-113)
//     for j in (k = 3, 7, 11):  xs (+)= ResBlock1_j(x0)                          (model.py:44-51, :115-120)
//     x = xs / 3;  x = leaky_relu(x, 0.01);  wav = tanh(conv_post(x))            (model.py:121-124)
// One workgroup = a window of W time steps x 32 channels that stays in LDS / registers from the stage input to the waveform samples.  What
// leaves the CU: W - 2 M fp32 samples per window, and the running MRF sum once (bf16, across the third ResBlock: the registers are needed there).
//
// Against three launches (whole-ResBlock kernels k = 3, 7; three pair launches at k = 11 with conv_post in the last: kernels_bf16_rbk.hip,
// kernels_bf16_rbg.hip) the stage input is read once instead of three times, two of the three read-modify-write passes of the MRF accumulator
// and the k = 11 pairs' two HBM round trips of x' are gone, and so are five launches.  Arithmetic: the SAME operations in the SAME order per
// element as that path (accumulators start from the bias block, k-steps tap-major, x' rounded to bf16 where the pair path stores it, LeakyReLU of
// the rounded value, MRF sum rounded to bf16 after every addition, mean as v * (1 / 3), conv_post as the streaming kernel's fmaf chain) — the
// samples are BIT-IDENTICAL to it (tests/test_gpu_bf16.py::test_stage_kernel_is_bit_identical).
//
// Structure (what differs from resblock_bf16_k):
//   * wave tile 32 x 128 = four 32-column blocks; every convolution runs in TWO parts: the wave's INTERIOR blocks (1, 2), whose operand rows all
//     lie in the wave's own 128 columns (the largest tap offset is H * dil <= 25 < 32), start right after the wave's own epilogue, BEFORE the
//     workgroup barrier; the EDGE blocks (0, 3), which read the neighbours' rows, follow it.  Every barrier has MFMA work on both sides, so a
//     wave that arrives early waits under its own matrix work, and the epilogue of the interior blocks (registers no MFMA of the edge part
//     touches) is issued between the edge part's MFMAs;
//   * a convolution's A operand is register-resident (k = 11: 88 VGPRs); while the edge part runs, the NEXT convolution's fragment q is
//     loaded into the register fragment q has just been read from for the last time: no weight ring, no exposed L2 latency, no extra registers;
//   * LDS addresses: one per tap (the lane's row + (tap - H) * dil in the 16-row-blocked tile, bf16_common.h::tile_off), recomputed per
//     convolution; blocks and k-steps are immediate offsets (+32 rows = +2048 bytes, k-step = +512);
//   * the running x (residual) and the MRF sum live in registers in the accumulator layout, packed bf16, as resblock_bf16_k holds x;
//   * the tile is staged FROM the residual registers (x0 is loaded once per ResBlock in the accumulator layout: no separate staging path);
//   * the last epilogue leaves its rows in tile A, and after one barrier every thread forms output samples with conv_post_bf16_k's fmaf chain.
// Window margin: M = max over the ResBlocks of H * (d0 + d1 + d2) + 3 H, + 3 for conv_post = 63 for V1; W = 512: 386 samples per window.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

#ifndef VTTS_ST_PF  // kernel-development switches (tools/kbench/cc_one.sh, build.py --define)
#define VTTS_ST_PF 1
#endif
#ifndef VTTS_ST_SPLIT  // 1 = every convolution in an interior and an edge part around the barrier (two accumulator blocks); 0 = resblock_bf16_k's order (four)
#define VTTS_ST_SPLIT 1
#endif
#ifndef VTTS_ST_NRES   // resident A fragments of a convolution with more than VTTS_ST_NRES_ALL k-steps
#define VTTS_ST_NRES 12
#define VTTS_ST_NRES_ALL 14
#endif

namespace vtts {

// kernel-development builds (-DVTTS_TIMELINE=1, tools/st_timeline.py): thread 0 of every workgroup stamps the shader clock at the phase boundaries
// into BStageArgs::dbg[wg * 128 + i]: 0 start, 1 staged; ResBlock rb, pair pr: 2 + 24 rb + 8 pr + {0 c1 interior, 1 barrier, 2 c1 edge (+ fill), 3 epilogue 1 (0, 3),
// 4 c2 interior, 5 barrier, 6 c2 edge (+ fill), 7 epilogue 2 (0, 3)}; 126 tail done
#if VTTS_TIMELINE
#define ST_TL(i)                                                                                                             \
    do {                                                                                                                     \
        if (a.dbg && threadIdx.x == 0) a.dbg[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 128 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ST_TL(i) do { } while (0)
#endif

constexpr int ST_XCD_MIN_TILES = 192;  // XCD-aware window order from this many windows per utterance slot on (as resblock_pair_g_bf16_k)

template <int K0_, int K1_, int K2_, int W_, int WN_, int MINWG_>
struct StTile {
    static constexpr int C = 32, K0 = K0_, K1 = K1_, K2 = K2_, W = W_, WN = WN_, MINWG = MINWG_;
    static constexpr int THREADS = 64 * WN, NR = W / WN / 32;
    static_assert(NR == 4 && W % (WN * 128) == 0, "wave tile 32 x 128: blocks 1, 2 interior, blocks 0, 3 edge");
    static constexpr int KMAX = K0 > K1 ? (K0 > K2 ? K0 : K2) : (K1 > K2 ? K1 : K2);
    static constexpr int HMAX = (KMAX - 1) / 2, MAXDIL = 5, GUARD = HMAX * MAXDIL;
    static_assert(GUARD < 32, "an interior block's taps stay inside the wave's own columns");
    static constexpr int ROWS = W + 2 * GUARD, SPR = 4, P = 64;
    static constexpr int TILE_BYTES = tile_rows16(ROWS) * P;
    static constexpr int NQMAX = 2 * KMAX;  // k-steps per convolution (two per tap at C = 32)
    // A fragments resident in registers per convolution: all of them up to k = 7 (56 VGPRs); at k = 11 the first 12 of 22, the rest through a 4-slot ring
    // (88 resident ones leave hipcc's allocator no room at two waves per SIMD: it spills, and a spill reload's s_waitcnt drains the weight loads in flight)
    static constexpr int nres(int ks) { return 2 * ks <= VTTS_ST_NRES_ALL ? 2 * ks : VTTS_ST_NRES; }
    static constexpr int NRESMAX = nres(K0) > nres(K1) ? (nres(K0) > nres(K2) ? nres(K0) : nres(K2)) : (nres(K1) > nres(K2) ? nres(K1) : nres(K2));
    static constexpr int BIAS_FLOATS = 18 * C, POST_K = 7, POST_FLOATS = POST_K * C;
    static constexpr int LDS_BYTES = 2 * TILE_BYTES + BIAS_FLOATS * 4 + POST_FLOATS * 4;
    static_assert(LDS_BYTES * MINWG <= 160 * 1024, "LDS");
};

template <class T>
__global__ __launch_bounds__(T::THREADS, (T::MINWG * T::THREADS + 255) / 256) void stage_bf16_k(BStageArgs a) {
    constexpr int C = T::C, W = T::W, GUARD = T::GUARD, THREADS = T::THREADS, KMAX = T::KMAX;
    constexpr int BLK = 2048, KST = 512;  // bytes between a lane's rows r and r + 32 / between a tap's two k-steps (slot + 2) in the blocked tile

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const tA = lds;                   // lrelu(x): c1's input; at the very end the stage's output rows (conv_post's input)
    unsigned char* const tT = lds + T::TILE_BYTES;   // lrelu(c1(.)): c2's input
    float* const sbias = reinterpret_cast<float*>(lds + 2 * T::TILE_BYTES);  // [3][6][C]
    float* const spost = sbias + T::BIAS_FLOATS;                              // conv_post's [7][C] (Haiku [K][Cin][1])

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = tid >> 6;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int Lp = a.L;                                                          // rows allocated per utterance
    const int L = a.lens ? min(max(a.lens[b], 0) * a.len_mul, a.L) : a.L;      // valid rows of this utterance (ragged batch: the rest reads as zero padding)
    const int M = a.margin;                                                      // invalid rows per side after the three ResBlocks and conv_post
    const int NT = W - 2 * M;                                                    // samples per window
    // XCD-aware window order (resblock_pair_g_bf16_k): an XCD takes a contiguous, balanced eighth of the utterance's valid windows
    const int ntv = (L + NT - 1) / NT, rx = (int)((blockIdx.x + b) & 7), lox = (rx * ntv) >> 3, hix = ((rx + 1) * ntv) >> 3;
    const bool xmap = (a.L + NT - 1) / NT >= ST_XCD_MIN_TILES;
    const int tile = xmap ? lox + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (xmap && tile >= hix) return;
    const int t0 = tile * NT;  // first sample of this window
    if (t0 >= L) return;
    const int tw = t0 - M;     // time of window row 0
    ST_TL(0);
    const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + (size_t)b * Lp * C;
    unsigned short* __restrict__ sg = static_cast<unsigned short*>(a.s) + (size_t)b * Lp * C;
    const int R0 = wn * 128 + l31;  // window row of this lane's column in block 0 (block nr: + 32 nr)
    const bool interior = tw >= 0 && tw + W <= L;  // every row of the window lies inside the utterance (workgroup-uniform)
    const int waddr = tile_off<4>(GUARD + R0, lh);  // the lane's own row, slot lh: block nr + nr * BLK, 16-channel group p + p * KST

    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };
    auto act2 = [](unsigned u) { return lrelu01_pack(bf16_lo(u), bf16_hi(u)); };  // LRELU_SLOPE, model.py:5

    for (int u = tid; u < T::BIAS_FLOATS; u += THREADS) sbias[u] = a.bias[u / (6 * C)][u % (6 * C)];
    for (int u = tid; u < T::POST_FLOATS; u += THREADS) spost[u] = a.post_w[u];
    // guard rows of both tiles = 0 (never written again)
    for (int u = tid; u < 2 * 2 * GUARD * 4; u += THREADS) {
        const int which = u / (2 * GUARD * 4), v = u % (2 * GUARD * 4);
        const int gr = v % (2 * GUARD), c = v / (2 * GUARD);
        const int row = gr < GUARD ? gr : W + gr;
        *reinterpret_cast<uint4*>((which ? tT : tA) + tile_off<4>(row, c)) = make_uint4(0u, 0u, 0u, 0u);
    }

    uint4 xr[4][2];   // the running x of this lane's outputs: accumulator layout, packed bf16 (block nr, channels 16 p + {4 lh + 0..3, 8 + 4 lh + 0..3})
    uint4 ms[4][2];   // the MRF sum, same layout
    f32x16 acc[VTTS_ST_SPLIT ? 2 : 4];
    auto ai = [](int nr) { return VTTS_ST_SPLIT ? ((nr == 0 || nr == 1) ? 0 : 1) : nr; };  // accumulator of block nr (split: blocks 1, 2 and then 0, 3 share two)
    bf16x8 aw[T::NRESMAX];  // the current convolution's resident A fragments (k-steps 0 .. NRES - 1)
    bf16x8 ar[4];           // ... and a ring for the k-steps behind them (k = 11 only), three steps ahead
    int ta[KMAX];           // per tap: byte offset of (the lane's row + (tap - H) * dil, slot lh) in a tile

    auto time_ok = [&](int nr) {
        const int t = tw + R0 + 32 * nr;
        return t >= 0 && t < L;
    };
    // stage input rows of block nr -> xr (accumulator layout); rows outside the utterance are the reference's zero padding
    auto load_x0 = [&](auto int_tag, int nr) {
        constexpr bool INT = decltype(int_tag)::value;
        const int t = tw + R0 + 32 * nr;
        const int tc = INT ? t : (t < 0 ? 0 : (t >= L ? L - 1 : t));
#pragma unroll
        for (int p = 0; p < 2; ++p) xr[nr][p] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + 16 * p + 8 * lh);
    };
    auto fix_x0 = [&](auto int_tag, int nr) {  // (after the loads have been requested for every block: mask, un-swap)
        constexpr bool INT = decltype(int_tag)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint4 r = xr[nr][p];
            if (!INT && !time_ok(nr)) r = make_uint4(0u, 0u, 0u, 0u);
            swap_pair(r.x, r.z);  // 8 consecutive channels per lane -> the accumulator layout's 4 + 4
            swap_pair(r.y, r.w);
            xr[nr][p] = r;
        }
    };
    // 8 values of block nr (two bf16 pairs of each channel quad) -> tile row: swap back to 8 consecutive channels per lane, one 16-byte write
    auto write_tile = [&](auto int_tag, unsigned char* tile, int nr, unsigned p0, unsigned p1, unsigned q0, unsigned q1, int p) {
        constexpr bool INT = decltype(int_tag)::value;
        if (!INT && !time_ok(nr)) p0 = p1 = q0 = q1 = 0u;
        swap_pair(p0, q0);
        swap_pair(p1, q1);
        *reinterpret_cast<uint4*>(tile + waddr + nr * BLK + p * KST) = make_uint4(p0, p1, q0, q1);
    };
    auto stage_from_xr = [&](auto int_tag, int nr) {  // tile A rows of block nr = lrelu(x0)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint4 r = xr[nr][p];
            write_tile(int_tag, tA, nr, act2(r.x), act2(r.y), act2(r.z), act2(r.w), p);
        }
    };
    auto set_taps = [&](auto ks_tag, int dl) {
        constexpr int KS = decltype(ks_tag)::value, H = (KS - 1) / 2;
#pragma unroll
        for (int tp = 0; tp < KS; ++tp) ta[tp] = tile_off<4>(GUARD + R0 + (tp - H) * dl, lh);
    };
    // A fragments: buffer loads — the lane's 16 bytes in a VGPR offset that never changes, the convolution's offset in an SGPR, the k-step an
    // immediate / SGPR addition: no per-load address arithmetic on the VALU (kernels_bf16_rbg.hip: VTTS_LEAN)
    struct WSrc {
        __amdgpu_buffer_rsrc_t rs;
        int soff;  // byte offset of the convolution's fragment 0
    };
    const unsigned a_voff = (unsigned)lane * 16;
    const __amdgpu_buffer_rsrc_t rs_w[3] = {
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp[0]), 0, 6 * 2 * T::K0 * 1024, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp[1]), 0, 6 * 2 * T::K1 * 1024, 0x00020000),
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp[2]), 0, 6 * 2 * T::K2 * 1024, 0x00020000)};
    auto wfrag = [&](int rb, int conv, int nqt) { return WSrc{rs_w[rb], conv * nqt * 1024}; };  // ResBlock rb's convolution conv (nqt k-steps of 1 KiB)
    auto wload = [&](const WSrc& w, int q) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w.rs, a_voff, w.soff + q * 1024, 0)); };
    // acc[blocks] = bias + W (*) tile over the convolution's 2 KS k-steps.  MODE 0: the interior blocks 1, 2; MODE 1: the edge blocks 0, 3; MODE 2: all
    // four (VTTS_ST_SPLIT = 0).  A operand of k-step q: aw[q] (resident) for q < NRES, else the ring, requested three steps ahead from `cur`.
    // NEXTN > 0: the NEXT convolution's resident fragments stream into aw[q] behind this part's last reads of it.  The instruction order is FIXED
    // by scheduling barriers: hipcc left to itself hoists a reload above the MFMAs that still read the register (and then needs a second one: 140
    // registers of fragments instead of 88), and sinks the look-ahead LDS reads to their uses.
    auto mfma_part = [&](auto ks_tag, auto mode_tag, auto nextn_tag, const unsigned char* __restrict__ tile, int conv, WSrc cur, WSrc nextw) {
        constexpr int KS = decltype(ks_tag)::value, NQT = 2 * KS, MODE = decltype(mode_tag)::value, NEXTN = decltype(nextn_tag)::value;
        constexpr int NRES = T::nres(KS), NB = MODE == 2 ? 4 : 2;
        constexpr int BLKS[4] = {MODE == 0 ? 1 : 0, MODE == 0 ? 2 : (MODE == 1 ? 3 : 1), 2, 3};
        // the accumulators start from the bias: the LDS copy lands in the first block's registers, the other blocks' first MFMAs take them as their C
        // operand (issued before the first block's own first MFMA overwrites them): no bias block, no moves
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 bv = *reinterpret_cast<const float4*>(sbias + conv * C + 8 * rq + 4 * lh);
            acc[ai(BLKS[0])][4 * rq + 0] = bv.x; acc[ai(BLKS[0])][4 * rq + 1] = bv.y; acc[ai(BLKS[0])][4 * rq + 2] = bv.z; acc[ai(BLKS[0])][4 * rq + 3] = bv.w;
        }
        constexpr int PF = VTTS_ST_PF < NQT ? VTTS_ST_PF : NQT - 1;  // k-steps a B fragment is requested ahead of its MFMAs
        bf16x8 bf[PF + 1][NB];
        auto load_b = [&](int q, int slot) {
            const int tp = q >> 1, ks = q & 1;
#pragma unroll
            for (int i = 0; i < NB; ++i) bf[slot][i] = *reinterpret_cast<const bf16x8*>(tile + ta[tp] + BLKS[i] * BLK + ks * KST);
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) load_b(q, q);
        if constexpr (NRES < NQT && NRES < 3) {
#pragma unroll
            for (int q = NRES; q < 3 && q < NQT; ++q) ar[(q - NRES) % 4] = wload(cur, q);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQT; ++q) {
            if (q + 3 >= NRES && q + 3 < NQT) ar[(q + 3 - NRES) % 4] = wload(cur, q + 3);
            if (q + PF < NQT) load_b(q + PF, (q + PF) % (PF + 1));
            const bf16x8 af = q < NRES ? aw[q < NRES ? q : 0] : ar[(q - NRES) % 4];
#pragma unroll
            for (int i = NB - 1; i >= 0; --i)
                acc[ai(BLKS[i])] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[q % (PF + 1)][i], q == 0 ? acc[ai(BLKS[0])] : acc[ai(BLKS[i])], 0, 0, 0);
            if (q < NEXTN && q < NRES) {
                __builtin_amdgcn_sched_barrier(0);  // the reload FOLLOWS the MFMAs that read aw[q] for the last time
                aw[q] = wload(nextw, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NEXTN > NRES) {
#pragma unroll
            for (int q = NRES; q < NEXTN; ++q) aw[q] = wload(nextw, q);
        }
    };
    // The epilogues between convolutions, per (block, 16-channel group p): four quarter units (two accumulator registers -> one packed bf16 pair) and
    // a write unit (lane exchange + one 16-byte LDS write).  KIND 0 = epilogue 1: xt = lrelu(c1(.)) -> tile T; KIND 1 = epilogue 2 between pairs:
    // x = c2 + x (model.py:50) rounded to bf16 as the pair path stores it, kept as the next residual, lrelu(x') -> tile A.
    auto ep_block = [&](auto int_tag, auto kind_tag, int nr) {
        constexpr int KIND = decltype(kind_tag)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r0 = 8 * p;
            const f32x16& c = acc[ai(nr)];
            if constexpr (KIND == 0) {
                write_tile(int_tag, tT, nr, lrelu01_pack(c[r0 + 0], c[r0 + 1]), lrelu01_pack(c[r0 + 2], c[r0 + 3]), lrelu01_pack(c[r0 + 4], c[r0 + 5]),
                           lrelu01_pack(c[r0 + 6], c[r0 + 7]), p);
            } else {
                const uint4 o = xr[nr][p];
                uint4 r;
                r.x = pack_bf16x2(vadd_raw(bf16_lo(o.x), c[r0 + 0]), vadd_raw(bf16_hi(o.x), c[r0 + 1]));
                r.y = pack_bf16x2(vadd_raw(bf16_lo(o.y), c[r0 + 2]), vadd_raw(bf16_hi(o.y), c[r0 + 3]));
                r.z = pack_bf16x2(vadd_raw(bf16_lo(o.z), c[r0 + 4]), vadd_raw(bf16_hi(o.z), c[r0 + 5]));
                r.w = pack_bf16x2(vadd_raw(bf16_lo(o.w), c[r0 + 6]), vadd_raw(bf16_hi(o.w), c[r0 + 7]));
                xr[nr][p] = r;
                write_tile(int_tag, tA, nr, act2(r.x), act2(r.y), act2(r.z), act2(r.w), p);  // lrelu of the ROUNDED x', as the pair path's staging does
            }
        }
    };
    auto add_res = [&](int nr) {  // x = c2 + x  (model.py:50)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r0 = 8 * p;
            const uint4 r = xr[nr][p];
            f32x16& c = acc[ai(nr)];
            c[r0 + 0] = vadd_raw(bf16_lo(r.x), c[r0 + 0]); c[r0 + 1] = vadd_raw(bf16_hi(r.x), c[r0 + 1]);
            c[r0 + 2] = vadd_raw(bf16_lo(r.y), c[r0 + 2]); c[r0 + 3] = vadd_raw(bf16_hi(r.y), c[r0 + 3]);
            c[r0 + 4] = vadd_raw(bf16_lo(r.z), c[r0 + 4]); c[r0 + 5] = vadd_raw(bf16_hi(r.z), c[r0 + 5]);
            c[r0 + 6] = vadd_raw(bf16_lo(r.w), c[r0 + 6]); c[r0 + 7] = vadd_raw(bf16_hi(r.w), c[r0 + 7]);
        }
    };
    auto add_ms = [&](int nr) {  // xs += rb(x)  (model.py:118-120)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r0 = 8 * p;
            const uint4 o = ms[nr][p];
            f32x16& c = acc[ai(nr)];
            c[r0 + 0] = vadd_raw(bf16_lo(o.x), c[r0 + 0]); c[r0 + 1] = vadd_raw(bf16_hi(o.x), c[r0 + 1]);
            c[r0 + 2] = vadd_raw(bf16_lo(o.y), c[r0 + 2]); c[r0 + 3] = vadd_raw(bf16_hi(o.y), c[r0 + 3]);
            c[r0 + 4] = vadd_raw(bf16_lo(o.z), c[r0 + 4]); c[r0 + 5] = vadd_raw(bf16_hi(o.z), c[r0 + 5]);
            c[r0 + 6] = vadd_raw(bf16_lo(o.w), c[r0 + 6]); c[r0 + 7] = vadd_raw(bf16_hi(o.w), c[r0 + 7]);
        }
    };
    auto pack_ms = [&](int nr) {  // the sum rounded to bf16, as the accumulator tensor of the launch-per-ResBlock path holds it
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r0 = 8 * p;
            const f32x16& c = acc[ai(nr)];
            ms[nr][p] = make_uint4(pack_bf16x2(c[r0 + 0], c[r0 + 1]), pack_bf16x2(c[r0 + 2], c[r0 + 3]), pack_bf16x2(c[r0 + 4], c[r0 + 5]), pack_bf16x2(c[r0 + 6], c[r0 + 7]));
        }
    };
    // rows of the window that later phases read back: the samples' rows and conv_post's halo (the 6 rows shared with a neighbour window hold the same bits there)
    auto keep_row = [&](int nr) {
        const int n = R0 + 32 * nr, t = tw + n;
        return n >= M - 3 && n < W - M + 3 && t >= 0 && t < L;
    };
    auto spill_ms = [&](int nr) {  // the sum leaves for the scratch tensor across the third ResBlock (its registers hold that convolution's weights)
        const int t = tw + R0 + 32 * nr;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint4 o = ms[nr][p];
            swap_pair(o.x, o.z);
            swap_pair(o.y, o.w);
            if (keep_row(nr)) *reinterpret_cast<uint4*>(sg + (size_t)t * C + 16 * p + 8 * lh) = o;
        }
    };
    auto fill_ms = [&](int nr) {
        const int t = tw + R0 + 32 * nr;
        const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
#pragma unroll
        for (int p = 0; p < 2; ++p) ms[nr][p] = *reinterpret_cast<const uint4*>(sg + (size_t)tc * C + 16 * p + 8 * lh);
    };
    auto fix_ms = [&](int nr) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint4 o = ms[nr][p];
            swap_pair(o.x, o.z);
            swap_pair(o.y, o.w);
            ms[nr][p] = o;
        }
    };
    const float rdv = mrf_recip(a.div);
    const float s_out = a.slope_out;
    // the stage's output rows of block nr: x = xs / num_kernels, the tail's LeakyReLU (model.py:121-122), bf16 -> tile A (conv_post's input; zero outside the utterance)
    auto ep_out = [&](int nr) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r0 = 8 * p;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = lrelu_f(vmul_raw(acc[ai(nr)][r0 + e], rdv), s_out);
            write_tile(std::false_type{}, tA, nr, pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]), p);
        }
    };

    // One ResBlock (model.py:44-51) + its share of the MRF bookkeeping.  RB = 0, 1, 2; KS its kernel size; KN the next ResBlock's (0: none)
    auto run_rb = [&](auto int_tag, auto rb_tag, auto ks_tag, auto kn_tag) {
        constexpr int RB = decltype(rb_tag)::value, KS = decltype(ks_tag)::value, KN = decltype(kn_tag)::value, NQT = 2 * KS;
        using NQ = std::integral_constant<int, T::nres(KS)>;                 // resident fragments the next convolution of this ResBlock needs
        using NN = std::integral_constant<int, KN ? T::nres(KN) : 0>;         // ... the next ResBlock's first convolution
        using Z = std::integral_constant<int, 0>;
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        using K0t = std::integral_constant<int, 0>;
        using K1t = std::integral_constant<int, 1>;
#pragma nounroll
        for (int pr = 0; pr < 3; ++pr) {
            const int dl = a.dils[RB][pr];
            const int cv = 6 * RB + 2 * pr;
            const int tl0 = 2 + 24 * RB + 8 * pr;
            (void)tl0;
            const WSrc w1 = wfrag(RB, 2 * pr, NQT), w2 = wfrag(RB, 2 * pr + 1, NQT);
            const WSrc wn = pr < 2 ? wfrag(RB, 2 * pr + 2, NQT) : wfrag(RB + 1 < 3 ? RB + 1 : 2, 0, 2 * KN);
            auto last = [&](int nr) {
                add_res(nr);
                if constexpr (RB == 0) {
                    pack_ms(nr);  // xs = rb_0(x)
                } else if constexpr (RB == 1) {
                    add_ms(nr);
                    pack_ms(nr);
                    spill_ms(nr);
                } else {
                    fix_ms(nr);
                    add_ms(nr);
                    ep_out(nr);
                }
                if constexpr (RB < 2) load_x0(int_tag, nr);  // the next ResBlock starts from the stage input again
            };
            auto restage = [&](int nr) {
                fix_x0(int_tag, nr);
                stage_from_xr(int_tag, nr);
            };
            if constexpr (VTTS_ST_SPLIT) {
                // interior part -> its epilogue (rows only this wave reads: no barrier needed) -> barrier -> edge part -> its epilogue
                // ---- c1 over A ----
                set_taps(ks_tag, dl);
                mfma_part(ks_tag, M0{}, Z{}, tA, cv, w1, w1);
                ST_TL(tl0 + 0);
                ep_block(int_tag, K0t{}, 1);
                ep_block(int_tag, K0t{}, 2);
                __syncthreads();  // the neighbours' edge rows of A are written; every wave is done reading T's edge rows
                ST_TL(tl0 + 1);
                mfma_part(ks_tag, M1{}, NQ{}, tA, cv, w1, w2);
                ST_TL(tl0 + 2);
                ep_block(int_tag, K0t{}, 0);
                ep_block(int_tag, K0t{}, 3);
                ST_TL(tl0 + 3);
                // ---- c2 over T (rate 1) ----
                set_taps(ks_tag, 1);
                mfma_part(ks_tag, M0{}, Z{}, tT, cv + 1, w2, w2);
                ST_TL(tl0 + 4);
                if (pr < 2) {
                    ep_block(int_tag, K1t{}, 1);
                    ep_block(int_tag, K1t{}, 2);
                    __syncthreads();  // the neighbours' edge rows of T are written; every wave is done reading A's edge rows
                    ST_TL(tl0 + 5);
                    mfma_part(ks_tag, M1{}, NQ{}, tT, cv + 1, w2, wn);
                    ST_TL(tl0 + 6);
                    ep_block(int_tag, K1t{}, 0);
                    ep_block(int_tag, K1t{}, 3);
                    ST_TL(tl0 + 7);
                } else {
                    if constexpr (RB == 2) {  // the sum comes back for the last addition
                        fill_ms(1);
                        fill_ms(2);
                    }
                    last(1);
                    last(2);
                    __syncthreads();
                    ST_TL(tl0 + 5);
                    if constexpr (RB == 2) {
                        fill_ms(0);
                        fill_ms(3);
                    }
                    mfma_part(ks_tag, M1{}, NN{}, tT, cv + 1, w2, wn);
                    ST_TL(tl0 + 6);
                    if constexpr (RB < 2) {
                        restage(1);
                        restage(2);
                    }
                    last(0);
                    last(3);
                    if constexpr (RB < 2) {
                        restage(0);
                        restage(3);
                    }
                    ST_TL(tl0 + 7);
                }
            } else {
                // the whole-ResBlock kernel's order (resblock_bf16_k): barrier -> all four blocks -> epilogue
                set_taps(ks_tag, dl);
                ST_TL(tl0 + 0);
                __syncthreads();  // A written; every wave is done reading T
                ST_TL(tl0 + 1);
                mfma_part(ks_tag, M2{}, NQ{}, tA, cv, w1, w2);
                ST_TL(tl0 + 2);
#pragma unroll
                for (int nr = 0; nr < 4; ++nr) ep_block(int_tag, K0t{}, nr);
                ST_TL(tl0 + 3);
                set_taps(ks_tag, 1);
                ST_TL(tl0 + 4);
                __syncthreads();  // T written; every wave is done reading A
                ST_TL(tl0 + 5);
                if (pr < 2) {
                    mfma_part(ks_tag, M2{}, NQ{}, tT, cv + 1, w2, wn);
                    ST_TL(tl0 + 6);
#pragma unroll
                    for (int nr = 0; nr < 4; ++nr) ep_block(int_tag, K1t{}, nr);
                } else {
                    mfma_part(ks_tag, M2{}, NN{}, tT, cv + 1, w2, wn);
                    ST_TL(tl0 + 6);
                    if constexpr (RB == 2) {
#pragma unroll
                        for (int nr = 0; nr < 4; ++nr) fill_ms(nr);
                    }
#pragma unroll
                    for (int nr = 0; nr < 4; ++nr) last(nr);
                    if constexpr (RB < 2) {
#pragma unroll
                        for (int nr = 0; nr < 4; ++nr) restage(nr);
                    }
                }
                ST_TL(tl0 + 7);
            }
        }
    };

    auto body = [&](auto int_tag) {
        // the first convolution's fragments and the stage input, all requests first
        {
            const WSrc w0 = wfrag(0, 0, 2 * T::K0);
#pragma unroll
            for (int q = 0; q < T::nres(T::K0); ++q) aw[q] = wload(w0, q);
        }
#pragma unroll
        for (int nr = 0; nr < 4; ++nr) load_x0(int_tag, nr);
#pragma unroll
        for (int nr = 0; nr < 4; ++nr) {
            fix_x0(int_tag, nr);
            stage_from_xr(int_tag, nr);
        }
        __syncthreads();  // biases, conv_post's weights and the guard rows are in LDS (tile A's rows are ordered by the first convolution's own barrier)
        ST_TL(1);
        run_rb(int_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, T::K0>{}, std::integral_constant<int, T::K1>{});
        run_rb(int_tag, std::integral_constant<int, 1>{}, std::integral_constant<int, T::K1>{}, std::integral_constant<int, T::K2>{});
        run_rb(int_tag, std::integral_constant<int, 2>{}, std::integral_constant<int, T::K2>{}, std::integral_constant<int, 0>{});
    };
    if (interior) body(std::true_type{});
    else body(std::false_type{});

    // ---------------- tanh(conv_post(.)) over the rows just written (model.py:123-124); conv_post_bf16_k's arithmetic, in its order ----------------
    __syncthreads();
    const float pb = a.post_b[0];
    for (int n = tid; n < NT; n += THREADS) {
        const int t = t0 + n;  // output sample: reads window rows M + n - 3 .. M + n + 3
        if (t >= L) break;
        float accp = 0.f;
#pragma unroll
        for (int j = 0; j < T::POST_K; ++j) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint4 v = *reinterpret_cast<const uint4*>(tA + tile_off<4>(GUARD + M + n - 3 + j, c));
                const float* w = spost + j * C + c * 8;
                accp = fmaf(w[0], bf16_lo(v.x), accp); accp = fmaf(w[1], bf16_hi(v.x), accp);
                accp = fmaf(w[2], bf16_lo(v.y), accp); accp = fmaf(w[3], bf16_hi(v.y), accp);
                accp = fmaf(w[4], bf16_lo(v.z), accp); accp = fmaf(w[5], bf16_hi(v.z), accp);
                accp = fmaf(w[6], bf16_lo(v.w), accp); accp = fmaf(w[7], bf16_hi(v.w), accp);
            }
        }
        a.wav[(size_t)b * Lp + t] = tanhf(accp + pb);
    }
    ST_TL(126);
}

// window geometry (kernel-development switch, tools/ab_bench.sh): 512 steps x 4 waves x 2 workgroups per CU, or 1024 x 8 x 1
#ifndef VTTS_ST_W
#define VTTS_ST_W 512
#define VTTS_ST_WN 4
#define VTTS_ST_WG 2
#endif
using ST32 = StTile<3, 7, 11, VTTS_ST_W, VTTS_ST_WN, VTTS_ST_WG>;

// the kernel exists for V1's last stage: C = 32, ResBlock1 with kernel sizes (3, 7, 11), rates <= 5, conv_post 32 -> 1, k = 7
bool stage_bf16_supported(int C, int nk, const int* ks, const int (*dils)[3], int post_cin, int post_cout, int post_k) {
    if (C != 32 || nk != 3 || ks[0] != ST32::K0 || ks[1] != ST32::K1 || ks[2] != ST32::K2) return false;
    if (post_cin != 32 || post_cout != 1 || post_k != ST32::POST_K) return false;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i)
            if (dils[j][i] < 1 || dils[j][i] > ST32::MAXDIL) return false;
    return stage_margin(ks, dils) * 2 + 128 <= ST32::W;
}
int stage_margin(const int* ks, const int (*dils)[3]) {
    int m = 0;
    for (int j = 0; j < 3; ++j) {
        const int H = (ks[j] - 1) / 2, mj = H * (dils[j][0] + dils[j][1] + dils[j][2]) + 3 * H;
        m = mj > m ? mj : m;
    }
    return m + 3;
}

hipError_t launch_stage_bf16(const BStageArgs& a, hipStream_t s) {
    using T = ST32;
    static DynLdsOnce once;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&stage_bf16_k<T>), T::LDS_BYTES, once); e != hipSuccess) return e;
    const int NT = T::W - 2 * a.margin;
    if (NT < 128) return hipErrorInvalidValue;
    dim3 grid((a.L + NT - 1) / NT, 1, a.B);
    if ((int)grid.x >= ST_XCD_MIN_TILES) grid.x = (grid.x + 7) / 8 * 8;
#if VTTS_TIMELINE  // kernel-development builds only: VTTS_ST_TL=<file> dumps the workgroups' phase stamps of every launch (the last one wins), synchronously
    static const char* const tl = getenv("VTTS_ST_TL");
    if (tl) {
        static unsigned long long* buf = nullptr;
        static size_t buf_bytes = 0;
        const size_t bytes = (size_t)grid.x * grid.z * 128 * 8;
        if (bytes > buf_bytes) {
            if (buf) (void)hipFree(buf);
            buf_bytes = hipMalloc(&buf, bytes) == hipSuccess ? bytes : 0;
            if (!buf_bytes) buf = nullptr;
        }
        if (buf) {
            (void)hipMemsetAsync(buf, 0, bytes, s);
            BStageArgs b = a;
            b.dbg = buf;
            hipLaunchKernelGGL(stage_bf16_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, b);
            (void)hipStreamSynchronize(s);
            unsigned long long* host = (unsigned long long*)malloc(bytes);
            (void)hipMemcpy(host, buf, bytes, hipMemcpyDeviceToHost);
            if (FILE* f = fopen(tl, "wb")) {
                fwrite(host, 1, bytes, f);
                fclose(f);
            }
            free(host);
            return hipGetLastError();
        }
    }
#endif
    hipLaunchKernelGGL(stage_bf16_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

const char* stage_kernel_name() { return "stage_bf16_k<StTile<3, 7, 11,"; }

}  // namespace vtts
