// Whole ResBlock1 in one kernel on the bf16 matrix pipe with SPLIT operands ("bf16x3"), fp32 channel-major activations:
//     for rate in (d0, d1, d2):  x = c2(lrelu(c1(lrelu(x)))) + x            (vietTTS/hifigan/model.py:44-51)
// followed by the MRF bookkeeping of model.py:115-121 (store / accumulate / accumulate-and-divide) — the split engine's answer to its narrow
// stages (round 5).
//
// Why (profiles/r04_i_bf16x3_pmc.md): at C = 32 / 64 a split-operand PAIR launch moves 2.1x its algorithmic bytes (x in, the fp32 residual
// again, x' out: 12 bytes per element and pair) for 3 x 2 * 2 * C * k bf16 flops per element — at C = 32, k = 3 a 512-column tile is 4.6k
// matrix-pipe cycles against ~15k cycles of the CU's share of HBM — and sits at MfmaUtil 0.13-0.31; stages 3 + 4 took 53 of the pass's 130 ms
// for a third of the FLOPs.  Fused per ResBlock the running x never leaves the CU between the three pairs: one read of the stage input
// (+ margin), one write (or read-modify-write) of the MRF accumulator — 8-12 bytes per element and RESBLOCK.
//
// Structure (kernels_bf16_rbk.hip's, re-done for fp32 in HBM and two-term operands):
//   * one workgroup = a window of W time steps x all C channels; every intermediate tensor lives at the SAME window coordinates (row r <-> time
//     tw + r), convolutions are centred (output row r reads input rows r + (j - H) * rate);
//   * the residual of an output element is always held by the same lane: the running x stays in REGISTERS, in fp32, in the MFMA accumulator
//     layout (16 * MR * NR values per lane) — exactly the fp32 value the pair-by-pair path would have stored to and re-read from HBM;
//   * ONE LDS tile, as a HI and a LO bf16 plane (channels-last rows, bf16_common.h: tile_off) of W + 2 * GUARD rows with zeroed guard rows: it holds
//     lrelu(x) (c1's B operand), then lrelu(c1(.)) (c2's), then lrelu(x') — a convolution's input is dead once its MFMA loop is over (the residual
//     is in registers), so each epilogue overwrites it between two workgroup barriers.  Half the LDS of separate A / T tiles: TWO workgroups of
//     four waves per CU on 512-step windows, one's staging / epilogues (VALU + memory) under the other's MFMA loops (the first cut — four planes,
//     one 8-wave workgroup per CU, every phase workgroup-wide — measured 3.72 / 6.02 / 8.75 ms per C = 32 ResBlock at k = 3 / 7 / 11);
//   * staging transposes: lane <-> time step (a wave reads 64 consecutive floats of one channel), 8 channels per unit, LeakyReLU, split, two
//     16-byte LDS writes (kernels_x3.hip: stage_x);
//   * MFMA loops: kernels_x3.hip's — per k-step A fragments hi / lo by buffer loads through a register ring, B fragments hi / lo one k-step ahead,
//     three MFMAs per operand pair, small terms first, look-ahead loads pinned between the MFMAs;
//   * rows whose dependency cone left the window are garbage after each convolution; the margin grows to M = H * (d0 + d1 + d2) + 3 H per side
//     (12 / 36 / 60 for k = 3 / 7 / 11); only the W - 2 M centre rows are stored;
//   * zero padding: every convolution of the reference pads ITS input with zeros outside [0, L): every tile write masks rows outside the
//     utterance (ragged batches: L = the utterance's valid length).
// Arithmetic: the SAME sequence of operations per output element as three launches of resblock_pair_x3_k (the bias is the C operand of every
// block's first MFMA in BOTH kernels — the accumulators start from it, nothing is added afterwards —, k-steps in tap-major order, the three terms
// in the same order, fp32 residual add, the same masks) — the results are BIT-IDENTICAL to the pair path
// (tests/test_gpu_x3.py::test_x3_whole_resblock_equals_the_pair_path), so every parity test of the split engine covers this kernel.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"
#include "device_common.h"

namespace vtts {

constexpr int RX_XCD_MIN_TILES = 64;

struct RbArgsX3 {
    ConvArgs a;            // x (stage input, raw), y (ResBlock output / MRF accumulator), B, L, lens / len_mul, slope_in, acc_mode, div, zrev
    int dils[3];           // the three pairs' rates
    const void* w[6];      // c1_0, c2_0, c1_1, c2_1, c1_2, c2_2: each [hi fragments][lo fragments] (kernels_x3.hip: pair_x3_pack)
    const float* bias[6];
    unsigned long long* dbg;  // kernel-development builds only (-DVTTS_TIMELINE=1, VTTS_RX_TL=<file>): per-workgroup s_memtime stamps at the phase boundaries; nullptr otherwise
};
#if VTTS_TIMELINE
#define RX_TL(i)                                                                                                   \
    do {                                                                                                           \
        if (p.dbg && threadIdx.x == 0) p.dbg[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 24 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RX_TL(i) do { } while (0)
#endif

template <int C_, int KS_, int W_, int WM_, int WN_, int WPS_, int KSX1_ = C_ / 16>
struct RXTile {
    static constexpr int C = C_, KS = KS_, W = W_, WM = WM_, WN = WN_, WPS = WPS_;
    static constexpr int THREADS = 64 * WM * WN, NWAVES = WM * WN;
    static constexpr int MR = C / WM / 32, NR = W / WN / 32;
    static constexpr int H = (KS - 1) / 2, MAXDIL = 5;
    static constexpr int GUARD = H * MAXDIL;               // rows a centred tap can reach beyond the window
    static constexpr int ROWS = W + 2 * GUARD;
    static constexpr int SPR = C / 8, P = C * 2;           // 16-byte slots / bytes per tile row
    static constexpr int KSTEPS = C / 16, MB = C / 32;
    static constexpr int NSTEPS = KS * KSTEPS;             // k-steps per convolution
    // k-steps per CHANNEL CHUNK of a c1 pass: the pair kernel this one must reproduce bit for bit stages C = 128 at k = 3 in two 64-channel chunks
    // (kernels_x3.hip: X128two), i.e. its c1 sums run chunk-major (channels 0-63 over all taps, then 64-127); c2 passes are tap-major everywhere
    static constexpr int KSX1 = KSX1_;
    static constexpr int PA = 1, RA = PA + 1;              // A-fragment ring: k-steps ahead / slots
    static constexpr int PLANE = tile_rows16(ROWS) * P;
    static constexpr int FS = W + 4;                       // row stride (floats) of the fp32 transposition area [C][FS] that shares the planes' LDS at both ends of the kernel
    static constexpr int AREA_BYTES = 2 * PLANE > C * FS * 4 ? 2 * PLANE : C * FS * 4;  // the planes, or the transposition area
    static constexpr int BIAS_BYTES = 6 * C * 4;            // the six convolutions' biases, staged once behind the planes (16 per-lane global loads per epilogue
                                                            // were a vector-memory instruction apiece, on the port the partner workgroup's MFMAs starve)
    static constexpr int LDS_BYTES = AREA_BYTES + BIAS_BYTES;
    static constexpr size_t CONV_BYTES = (size_t)KS * C * C * 2;  // one plane (hi or lo) of one convolution
    static_assert(C % (WM * 32) == 0 && W % (WN * 32) == 0 && W % 64 == 0, "window / wave tiling");
    static_assert(KSTEPS % RA == 0 && KSTEPS % 2 == 0 && KSX1 % RA == 0 && KSX1 % 2 == 0 && KSTEPS % KSX1 == 0, "ring slot / B parity are compile-time positions in a tap");
    static_assert(LDS_BYTES <= 160 * 1024 && 2 * LDS_BYTES <= 160 * 1024, "LDS budget: two workgroups per CU");
};

template <class T>
__global__ __launch_bounds__(T::THREADS, T::WPS) void resblock_x3_k(RbArgsX3 p) {
    constexpr int C = T::C, KS = T::KS, W = T::W, WN = T::WN, MR = T::MR, NR = T::NR, H = T::H, GUARD = T::GUARD;
    constexpr int SPR = T::SPR, KSTEPS = T::KSTEPS, MB = T::MB, NSTEPS = T::NSTEPS, PA = T::PA, RA = T::RA, THREADS = T::THREADS, NWAVES = T::NWAVES;
    const ConvArgs& a = p.a;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const aHi = lds;                   // the tile: lrelu(x), then lrelu(c1(.)), then lrelu(x') ... (HI terms)
    unsigned char* const aLo = lds + T::PLANE;        // (LO terms)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, lh = lane >> 5;
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int LP = a.L;              // row pitch of x / y
    const int L = valid_len(a, b);   // this utterance's columns (ragged batches; == LP otherwise)
    const int d0 = p.dils[0], d1 = p.dils[1], d2 = p.dils[2];
    const int M = H * (d0 + d1 + d2) + 3 * H;  // invalid margin per side after the three pairs
    const int NT = W - 2 * M;                  // outputs per workgroup
    int tile = blockIdx.x;
    if (gridDim.x >= RX_XCD_MIN_TILES) {  // XCD-aware tile order: XCD blockIdx.x % 8 takes a contiguous, balanced eighth of the utterance's valid windows
        const int nt = (L + NT - 1) / NT, r = (int)((blockIdx.x + blockIdx.z) & 7), lo = (r * nt) >> 3, hi = ((r + 1) * nt) >> 3;
        tile = lo + (int)(blockIdx.x >> 3);
        if (tile >= hi) return;
    }
    const int t0 = tile * NT;  // first output time step
    if (t0 >= L) return;
    // (A start-up skew of the launch's second batch of workgroups — so that the two workgroups of a CU alternate MFMA loops and epilogues — was
    //  measured in round 5 and changed nothing: they de-phase by themselves, profiles/r05_c_x3_rb_findings.md.)
    const int tw = t0 - M;     // time of window row 0
    const bool interior = tw >= 0 && tw + W <= L;  // every row of the window lies inside the utterance: no clamps, no zero-padding masks (workgroup-uniform)
    RX_TL(0);
#if VTTS_TIMELINE
    if (p.dbg && threadIdx.x == 0) p.dbg[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 24 + 23] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    const float* __restrict__ xb = a.x + (long)b * C * LP;
    float* yb = a.y + (long)b * C * LP;
    float* const sbias = reinterpret_cast<float*>(lds + T::AREA_BYTES);  // [6][C]; written here, read from epilogue 1 of pair 0 on (two barriers later)
    for (int u = tid; u < 6 * C; u += THREADS) sbias[u] = p.bias[u / C][u % C];
    const int cb0 = wm * (C / T::WM);  // this wave's first output channel
    const float slope = a.slope_in;    // LRELU_SLOPE, both activations of every pair (model.py:46,48)

    auto split2 = [](float v0, float v1, unsigned& hi, unsigned& lo) {  // two bf16 terms of two fp32 values (kernels_x3.hip)
        hi = pack_bf16x2(v0, v1);
        lo = pack_bf16x2(v0 - bf16_lo(hi), v1 - bf16_hi(hi));
    };
    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };

    // ---- the running x of this lane's outputs, fp32, in the accumulator layout: column n = col0 + 32 nr (time tw + n), channel cb0 + 32 mr +
    //      (r & 3) + 8 (r >> 2) + 4 lh; a half-wave reads 32 consecutive floats of one channel.  The waves' (channel block, column range) tiles cover
    //      the window exactly once, so these registers are also what the tile is STAGED from (write_tile below: the first cut read x twice) ----
    const int col0 = wn * (W / WN) + l31;
    float xr[MR][NR][16];
    // (A 16-byte form of these loads — rows through an fp32 transposition area in LDS, as the stores at the kernel's end — measured SLOWER: two more
    //  barriers and an LDS round trip in front of the first MFMA, 10k -> 25k cycles of a 76k-cycle window at k = 3; gpurun_out/r05_x3ab4.)
    if (interior) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[mr][nr][r] = xb[(long)(cb0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + tw + col0 + nr * 32];
    } else {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = tw + col0 + nr * 32;
                const bool ok = t >= 0 && t < L;
                const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = xb[(long)(cb0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + tc];
                    xr[mr][nr][r] = ok ? v : 0.0f;
                }
            }
    }
    // ---- guard rows of both planes = 0 (never written again) ----
    for (int u = tid; u < 2 * GUARD * SPR; u += THREADS) {
        const int gr = u % (2 * GUARD), c = u / (2 * GUARD);
        const int row = gr < GUARD ? gr : W + gr;  // [0, GUARD) and [GUARD + W, ROWS)
        const int off = tile_off<SPR>(row, c);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(aHi + off) = z;
        *reinterpret_cast<uint4*>(aLo + off) = z;
    }

    f32x16 acc[MR][NR];

    // ---- one convolution over a tile pair: acc += W (*) tile, centred taps: output column n reads tile row GUARD + n + (tap - H) * dl ----
    // A fragment (plane, tap, ks, mr): 16 bytes per lane at  w + plane * CONV_BYTES + (((tap * KSTEPS + ks) * MB + wm * MR + mr) * 64 + lane) * 16
    // B fragment (plane, tap, ks, nr): 16-byte slot 2 ks + lh of that row
    // The weight stream is ONE sequence of 6 * NSTEPS k-steps through a register ring that runs on across the epilogues: a convolution's last look-ahead
    // loads are the NEXT convolution's first fragments, in flight while the epilogue runs (the first cut started every convolution with an exposed L2
    // round trip: six per window).  NSTEPS is a multiple of RA, so every convolution starts at ring slot 0.
    bf16x8 af[RA][MR][2];
    const unsigned a_voff = (unsigned)((wm * MR) * 64 + lane) * 16;
    auto load_a_from = [&](const void* w, int sc, int slot) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, (int)(2 * T::CONV_BYTES), 0x00020000);
        const int soff = (sc * MB) * 1024;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            af[slot][mr][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff + mr * 1024, soff, 0));
            af[slot][mr][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff + mr * 1024, soff + (int)T::CONV_BYTES, 0));
        }
    };
#pragma unroll
    for (int s = 0; s < PA; ++s) load_a_from(p.w[0], s, s % RA);  // the stream's first fragments, under the loads of x

    // The accumulators START FROM THE BIAS: the very first MFMA of every 32 x 32 block takes a 16-register bias block (read from the LDS copy) as its C
    // operand — no accumulator initialisation (64 v_mov per convolution and wave) and no bias add in the epilogues (64 more), on a kernel whose epilogues
    // are VALU-bound.  resblock_pair_x3_k starts its accumulators from the same values (kernels_x3.hip: init_acc), so the two stay bit-identical.
    auto conv_phase = [&](auto nks_tag, const float* __restrict__ bias_lds, const void* __restrict__ w, const void* __restrict__ wnext, int dl, const unsigned char* __restrict__ thi, const unsigned char* __restrict__ tlo) {
        constexpr int NKS = decltype(nks_tag)::value;  // k-steps per channel chunk: flat step s = (chunk * KS + tap) * NKS + i  <->  k-step chunk * NKS + i of tap `tap`
        f32x16 bblk[MR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 bv = *reinterpret_cast<const float4*>(bias_lds + cb0 + mr * 32 + 8 * rq + 4 * lh);
                bblk[mr][4 * rq + 0] = bv.x;
                bblk[mr][4 * rq + 1] = bv.y;
                bblk[mr][4 * rq + 2] = bv.z;
                bblk[mr][4 * rq + 3] = bv.w;
            }
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, (int)(2 * T::CONV_BYTES), 0x00020000);
        const auto rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wnext ? wnext : w), 0, (int)(2 * T::CONV_BYTES), 0x00020000);
        bf16x8 bf[2][NR][2];
        auto load_a = [&](int s, int slot) {  // flat step s = tap * KSTEPS + ks; past the end: the next convolution's first steps (the last convolution re-reads its last step, never used)
            const bool nxt = s >= NSTEPS;
            const int sf = !nxt ? s : (wnext ? s - NSTEPS : NSTEPS - 1);
            // (the next convolution's first PA steps: chunk 0, tap 0 in either order, PA < NKS)
            const int chunk = nxt ? 0 : sf / (KS * NKS), rem = nxt ? sf : sf - chunk * (KS * NKS), tap = rem / NKS, ks = chunk * NKS + rem - tap * NKS;
            const int soff = ((tap * KSTEPS + ks) * MB) * 1024;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                af[slot][mr][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(nxt ? rs_n : rs_w, a_voff + mr * 1024, soff, 0));
                af[slot][mr][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(nxt ? rs_n : rs_w, a_voff + mr * 1024, soff + (int)T::CONV_BYTES, 0));
            }
        };
        auto load_b = [&](int s, int par) {
            const int sf = s < NSTEPS ? s : NSTEPS - 1;
            const int chunk = sf / (KS * NKS), rem = sf - chunk * (KS * NKS), tap = rem / NKS, ks = chunk * NKS + rem - tap * NKS;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int off = tile_off<SPR>(GUARD + col0 + nr * 32 + (tap - H) * dl, ks * 2 + lh);
                bf[par][nr][0] = *reinterpret_cast<const bf16x8*>(thi + off);
                bf[par][nr][1] = *reinterpret_cast<const bf16x8*>(tlo + off);
            }
        };
        load_b(0, 0);  // (the first PA A fragments are already in the ring)
        auto tap_iter = [&](int s0, auto first_tag) {  // one (chunk, tap): NKS k-steps; first_tag: the phase's very first step (C operand = the bias block)
            constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
            for (int i = 0; i < NKS; ++i) {
                load_a(s0 + i + PA, (i + PA) % RA);
                load_b(s0 + i + 1, (i + 1) & 1);
                const int sl = i % RA, par = i & 1;
                // small terms first (the order of resblock_pair_x3_k: bit-identical sums)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sl][mr][1], bf[par][nr][0], (FIRST && i == 0) ? bblk[mr] : acc[mr][nr], 0, 0, 0);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sl][mr][0], bf[par][nr][1], acc[mr][nr], 0, 0, 0);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sl][mr][0], bf[par][nr][0], acc[mr][nr], 0, 0, 0);
                // keep hipcc from sinking the look-ahead loads to their uses; spread them between this step's MFMAs (kernels_x3.hip)
                constexpr int NMF = 3 * MR * NR, NA = 2 * MR, NB = 2 * NR;
                int done = 0;
#pragma unroll
                for (int m = 0; m < NMF; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    const int upto = (m + 1) * (NA + NB) / NMF;
                    for (; done < upto; ++done) {
                        if (done < NA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
        };
        tap_iter(0, std::true_type{});
#pragma unroll 1
        for (int s0 = NKS; s0 < NSTEPS; s0 += NKS) tap_iter(s0, std::false_type{});
    };

    // this lane's 16 values of block (mr, nr), already biased / activated / masked by the caller -> split -> tile rows (8 consecutive channels per
    // lane after the exchange across the wave halves, as resblock_pair_x3_k's epilogue 1)
    auto write_tile = [&](unsigned char* thi, unsigned char* tlo, int mr, int nr, const float (&v)[16]) {
        const int row = GUARD + col0 + nr * 32;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            unsigned hp0, hp1, hq0, hq1, lp0, lp1, lq0, lq1;
            split2(v[8 * pp + 0], v[8 * pp + 1], hp0, lp0);
            split2(v[8 * pp + 2], v[8 * pp + 3], hp1, lp1);
            split2(v[8 * pp + 4], v[8 * pp + 5], hq0, lq0);
            split2(v[8 * pp + 6], v[8 * pp + 7], hq1, lq1);
            swap_pair(hp0, hq0);
            swap_pair(hp1, hq1);
            swap_pair(lp0, lq0);
            swap_pair(lp1, lq1);
            const int off = tile_off<SPR>(row, ((cb0 + mr * 32 + 16 * pp) >> 3) + lh);
            *reinterpret_cast<uint4*>(thi + off) = make_uint4(hp0, hp1, hq0, hq1);
            *reinterpret_cast<uint4*>(tlo + off) = make_uint4(lp0, lp1, lq0, lq1);
        }
    };

    // ---- stage the tile = lrelu(x) from the residual registers (zero outside the utterance: xr is) ----
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = lrelu_f(xr[mr][nr][r], slope);
            write_tile(aHi, aLo, mr, nr, v);
        }
    RX_TL(1);
    __syncthreads();  // tile staged, guard rows zeroed
    RX_TL(2);

#pragma unroll 1
    for (int pr = 0; pr < 3; ++pr) {
        const int dl = pr == 0 ? d0 : (pr == 1 ? d1 : d2);
        const float* const b1 = sbias + (2 * pr) * C;
        const float* const b2 = sbias + (2 * pr + 1) * C;
        // ---- c1 over the tile ----
        conv_phase(std::integral_constant<int, T::KSX1>{}, b1, p.w[2 * pr], p.w[2 * pr + 1], dl, aHi, aLo);
        RX_TL(3 + 6 * pr);
        __syncthreads();  // every wave is done reading lrelu(x)
        RX_TL(4 + 6 * pr);
        // ---- xt = lrelu(c1 + b1), zero outside [0, L) (c2's own zero padding applies to xt) -> the tile ----
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = lrelu_f(acc[mr][nr][r], slope);  // (b1 is in the sum already) max(v, slope v): the compare-and-select form's value for every finite v (bf16_common.h)
                if (!interior) {  // (workgroup-uniform branch: all but an utterance's first and last windows skip the 16 v_cndmask per block — 19 cycles each alone, profiles/r03_a_coissue_findings.md)
                    const int t = tw + col0 + nr * 32;
                    const bool ok = t >= 0 && t < L;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (!ok) v[r] = 0.0f;
                }
                write_tile(aHi, aLo, mr, nr, v);
            }
        }
        __syncthreads();  // xt written
        RX_TL(5 + 6 * pr);
        // ---- c2 over the tile (rate 1) ----
        conv_phase(std::integral_constant<int, KSTEPS>{}, b2, p.w[2 * pr + 1], pr < 2 ? p.w[2 * pr + 2] : nullptr, 1, aHi, aLo);
        RX_TL(6 + 6 * pr);
        // ---- x = (c2 + b2) + x  (model.py:50), in fp32 as the pair path's epilogue 2 ----
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[mr][nr][r] = acc[mr][nr][r] + xr[mr][nr][r];  // (b2 is in the sum already)
        if (pr < 2) {
            __syncthreads();  // every wave is done reading xt
            RX_TL(7 + 6 * pr);
            // the next pair's c1 input: lrelu(x') of the value the pair path would have stored and re-read, zero outside the utterance
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = lrelu_f(xr[mr][nr][r], slope);
                    if (!interior) {
                        const int t = tw + col0 + nr * 32;
                        const bool ok = t >= 0 && t < L;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (!ok) v[r] = 0.0f;
                    }
                    write_tile(aHi, aLo, mr, nr, v);
                }
            __syncthreads();  // lrelu(x') written
            RX_TL(8 + 6 * pr);
        }
    }

    // ---------------- MRF bookkeeping + store of the W - 2 M centre rows (device_common.h: epilogue_store's operations, in its order) ----------------
    const int mode = a.acc_mode;
    const float dv = a.div;
    // The stores leave in 16-byte accesses through an fp32 transposition area [C][W + 4] in the planes' LDS (dead by now): beside the partner
    // workgroup's MFMA stream a wave's vector-memory instructions only issue into idle matrix-pipe time (profiles/r03_a_coissue_findings.md), and the
    // per-lane dword form — 64 stores (+ 64 loads of the MRF accumulator) per lane — took 15k of a 76k-cycle window at k = 3 (r05 timelines; 10k this
    // way).  Needs every boundary on a multiple of 4 columns (the V1 shapes are); the dword form below stays as the general path.
    float* const fs = reinterpret_cast<float*>(lds);
    constexpr int FS = T::FS;
    const bool vec4 = ((M | L | LP) & 3) == 0;  // (then t0 = tile * (W - 2 M) is a multiple of 4 too)
    if (vec4) {
        __syncthreads();  // every wave is done reading the tile: the area is the fp32 transposition area again
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) fs[(cb0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * FS + col0 + nr * 32] = xr[mr][nr][r];
        __syncthreads();
        const int NT4 = NT / 4, total = C * NT4;
        constexpr int UB = 4;  // 16-byte units per round trip: the accumulator's values of a batch are requested before its first store (y may alias nothing else, but the compiler cannot know)
        for (int u0 = tid; u0 < total; u0 += UB * THREADS) {
            float4 yv[UB];
            float* yp[UB];
            bool live[UB];
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                const int u = u0 + i * THREADS;
                const int uc = u < total ? u : total - 1;
                const int ch = uc / NT4, q = uc - ch * NT4;
                const int t = t0 + 4 * q;
                live[i] = u < total && t < L;
                yp[i] = yb + (long)ch * LP + (live[i] ? t : t0);
                yv[i] = mode != ACC_STORE ? *reinterpret_cast<const float4*>(yp[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                const int u = u0 + i * THREADS;
                const int uc = u < total ? u : total - 1;
                const int ch = uc / NT4, q = uc - ch * NT4;
                const float4 v = *reinterpret_cast<const float4*>(fs + ch * FS + M + 4 * q);
                float4 o = v;
                if (mode == ACC_ADD) o = make_float4(yv[i].x + v.x, yv[i].y + v.y, yv[i].z + v.z, yv[i].w + v.w);
                else if (mode == ACC_MEAN) o = make_float4((yv[i].x + v.x) / dv, (yv[i].y + v.y) / dv, (yv[i].z + v.z) / dv, (yv[i].w + v.w) / dv);
                if (live[i]) *reinterpret_cast<float4*>(yp[i]) = o;
            }
        }
        RX_TL(22);
        return;
    }
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int n = col0 + nr * 32;
            const int t = tw + n;
            const bool ok = n >= M && n < W - M && t < L;  // (n >= M implies t >= t0 >= 0)
            const int tc = ok ? t : 0;
            float yv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[r] = mode != ACC_STORE ? yb[(long)(cb0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + tc] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = xr[mr][nr][r];
                if (mode == ACC_ADD) v = yv[r] + v;
                else if (mode == ACC_MEAN) v = (yv[r] + v) / dv;
                if (ok) yb[(long)(cb0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + tc] = v;
            }
        }
    RX_TL(22);
}

// ---- tile table ------------------------------------------------------------------------------------
//                                           C   KS   W   WM WN WPS
#ifndef VTTS_RX32_W  // C = 32: 512-step windows on four waves (wave tile 32 x 128), 74 KB of LDS: two workgroups per CU
#define VTTS_RX32_W 512
#define VTTS_RX32_WN 4
#define VTTS_RX32_WPS 2
#endif
#ifndef VTTS_RX64_W  // C = 64: 256-step windows on four waves (wave tile 32 x 128), 74 KB: two workgroups per CU
#define VTTS_RX64_W 256
#define VTTS_RX64_WM 2
#define VTTS_RX64_WN 2
#define VTTS_RX64_WPS 2
#endif
template <int KS> using RX32 = RXTile<32, KS, VTTS_RX32_W, 1, VTTS_RX32_WN, VTTS_RX32_WPS>;
template <int KS> using RX64 = RXTile<64, KS, VTTS_RX64_W, VTTS_RX64_WM, VTTS_RX64_WN, VTTS_RX64_WPS>;
template <int KS> using RX128 = RXTile<128, KS, 128, 4, 1, 2, 4>;  // C = 128, k = 3: 128-step windows (19 % margin), four waves = four 32-channel blocks, 74 KB: two workgroups per CU

template <class T>
static hipError_t launch_rx(const RbArgsX3& p, hipStream_t s) {
    static DynLdsOnce once;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&resblock_x3_k<T>), T::LDS_BYTES, once); e != hipSuccess) return e;
    int dsum = 0;
    for (int i = 0; i < 3; ++i) {
        if (p.dils[i] < 1 || p.dils[i] > T::MAXDIL) return hipErrorInvalidValue;
        dsum += p.dils[i];
    }
    const int NT = T::W - 2 * (T::H * dsum + 3 * T::H);
    if (NT < 32) return hipErrorInvalidValue;
    dim3 grid((p.a.L + NT - 1) / NT, 1, p.a.B);
    if ((int)grid.x >= RX_XCD_MIN_TILES) grid.x = (grid.x + 7) / 8 * 8;
    hipLaunchKernelGGL(resblock_x3_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, p);
    return hipGetLastError();
}

// where the kernel exists: C = 32 (k = 3, 7, 11), C = 64 (k = 3, 7: at k = 11 two workgroups' planes of a 256-step window pass the LDS budget) and C = 128 (k = 3)
bool resblock_x3_supported(int C, int K, const int* dils, int L) {
    for (int i = 0; i < 3; ++i)
        if (dils[i] < 1 || dils[i] > 5) return false;
    if (L < 1 || (long)C * L >= (1l << 31)) return false;
    if (C == 32) return K == 3 || K == 7 || K == 11;
    if (C == 64) return K == 3 || K == 7;
    if (C == 128) return K == 3;
    return false;
}
// ... and where it is the faster choice than three pair launches
// per ResBlock at 64 x 1024 frames (rocprofv3, gpurun_out/r05_x3ab7 vs r05_x3ab): C = 32: k = 3 2.65 vs 6.03 ms, k = 7 4.90 vs 8.31, k = 11 7.56 vs 9.77;
// C = 64: k = 3 3.79 vs 6.71, k = 7 9.17 vs 9.62; C = 128, k = 3: 7.21 vs 8.96
bool resblock_x3_preferred(int C, int K) { return C == 32 || (C == 64 && K <= 7) || (C == 128 && K == 3); }

// a = the ResBlock's ConvArgs (x = stage input, y = output with acc_mode / div, B, L, lens, slope_in, zrev); w[q] / bias[q] = the six convolutions
hipError_t launch_resblock_x3(const ConvArgs& a, const int* dils, const void* const* w, const float* const* bias, hipStream_t s) {
    RbArgsX3 p;
    p.a = a;
    p.dbg = nullptr;
#if VTTS_TIMELINE  // kernel-development builds only (--define VTTS_TIMELINE=1; tools/rx_timeline.py): not in the shipped library
    static const char* const tl = getenv("VTTS_RX_TL");  // read once
    if (tl) {  // kernel-development: stamps of ONE class (VTTS_RX_TL_K, default 3; C = VTTS_RX_TL_C, default 32), last launch wins; dumped right away (synchronous)
        const int kk = getenv("VTTS_RX_TL_K") ? atoi(getenv("VTTS_RX_TL_K")) : 3, cc = getenv("VTTS_RX_TL_C") ? atoi(getenv("VTTS_RX_TL_C")) : 32;
        if (a.K == kk && a.Cin == cc && a.Cin <= 64) {
            static unsigned long long* buf = nullptr;
            static size_t buf_bytes = 0;
            const size_t nwg = (size_t)((a.L + 31) / 32 + 8) * a.B, bytes = nwg * 24 * 8;
            if (bytes > buf_bytes) {  // a later launch of the class may be larger than the first
                if (buf) (void)hipFree(buf);
                buf_bytes = hipMalloc(&buf, bytes) == hipSuccess ? bytes : 0;
                if (!buf_bytes) buf = nullptr;
            }
            if (buf) {
                (void)hipMemsetAsync(buf, 0, bytes, s);
                p.dbg = buf;
                for (int i = 0; i < 3; ++i) p.dils[i] = dils[i];
                for (int q = 0; q < 6; ++q) {
                    p.w[q] = w[q];
                    p.bias[q] = bias[q];
                }
                hipError_t e = a.Cin == 32 ? (a.K == 3 ? launch_rx<RX32<3>>(p, s) : a.K == 7 ? launch_rx<RX32<7>>(p, s) : launch_rx<RX32<11>>(p, s))
                                           : (a.K == 3 ? launch_rx<RX64<3>>(p, s) : launch_rx<RX64<7>>(p, s));
                (void)hipStreamSynchronize(s);
                unsigned long long* host = (unsigned long long*)malloc(bytes);
                (void)hipMemcpy(host, buf, bytes, hipMemcpyDeviceToHost);
                if (FILE* f = fopen(tl, "wb")) {
                    fwrite(host, 1, bytes, f);
                    fclose(f);
                }
                free(host);
                return e;
            }
        }
    }
#endif
    for (int i = 0; i < 3; ++i) p.dils[i] = dils[i];
    for (int q = 0; q < 6; ++q) {
        p.w[q] = w[q];
        p.bias[q] = bias[q];
    }
    if (a.Cin == 32) switch (a.K) {
            case 3: return launch_rx<RX32<3>>(p, s);
            case 7: return launch_rx<RX32<7>>(p, s);
            case 11: return launch_rx<RX32<11>>(p, s);
        }
    if (a.Cin == 64) switch (a.K) {
            case 3: return launch_rx<RX64<3>>(p, s);
            case 7: return launch_rx<RX64<7>>(p, s);
        }
    if (a.Cin == 128 && a.K == 3) return launch_rx<RX128<3>>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace vtts
