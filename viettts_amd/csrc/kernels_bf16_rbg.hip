// Fused ResBlock1 pair in bf16, weights straight from L2 into registers:   x' = c2(lrelu(c1(lrelu(x)))) + x
// (vietTTS/hifigan/model.py:45-50).
//
// What the per-workgroup timelines of the LDS-staged generations showed (tools/kbench, profiles/r01_e_*): the MFMA
// main loops run at the matrix-pipe rate only when nothing in them synchronises the workgroup; every scheme that
// streams the weight slabs through LDS needs such a synchronisation per slab (s_barrier, or ready/done flags that a
// loader wave cannot serve fast enough) and lost 25-45 % of the loop to it.  This generation has NO shared weight
// staging at all:
//   * A operands (weights): host-packed in MFMA A-fragment order, so one global_load_dwordx4 per lane is one whole
//     fragment (1 KiB per wave, perfectly coalesced, L2/L1-resident: every workgroup reads the same 0.1-0.7 MB).  Each
//     wave loads the fragments of ITS m-blocks PA k-steps ahead into a register ring.  No LDS, no barrier, no flag.
//   * B operands (activations): the X tile (later the xt tile) in LDS, staged once per tile as before; each wave reads
//     its fragments one k-step ahead.
//   * wave tile 64 x 128 (MR = 2, NR = 4): per 8 MFMAs 2 KiB of A through the vector-memory path (half its 64 B/clk)
//     and 4 KiB of B through LDS (a quarter of its rate) — half the LDS traffic of the 64 x 64 wave tiles.
//   * LDS holds only the activation tile (<= 78 KiB), so two 4-wave workgroups share a CU; nothing couples their phases.
//     C = 256: 128-column tiles, the X tile staged 128 input channels at a time (2 x 45 KiB), the xt tile 69 KiB.
// s_barrier remains at the three tile-level hand-offs (X staged / X dead / xt written).
// Tried and measured no better (tools/kbench, round 1; profiles/r01_e_*, r01_j_kbench_findings.md): 32 x 128 wave tiles with
// three workgroups per CU (LDS read traffic per MFMA doubles), a 7-deep weight ring, 8-wave workgroups, a forced start offset
// between a CU's two workgroups (they drift to anti-phase by themselves), random start delays that desynchronise the CUs,
// s_setprio in the loops or in the staging / epilogue phases, a channel-blocked global layout for the epilogue's accesses
// (emulated: +1.5 %), the raw residual rows kept in LDS at C = 32 (-29 % HBM bytes, 0.8 % slower).  What did help late in the
// round: one look-ahead load after each MFMA instead of grouped issue (+2.1 %), LeakyReLU as packed multiply + raw v_max
// and the bias block as the first MFMA's C operand (-30 % VALU instructions per tile, +0.7 %).
// Round 2: lean loop addressing (VTTS_LEAN: weight fragments by buffer loads with the k-step in an SGPR offset, one row
// address + swizzle term per tap and v_xad_u32 per k-step for the LDS fragments: 43 -> 21 VALU per 64 MFMAs, 2-5 % per pair
// launch, profiles/r02_a_pair_kernel_findings.md).  The same treatment of the staging loads and of epilogue 2's row accesses
// (buffer loads / stores with SGPR row offsets, -200 VALU per tile) measured no faster.  A persistent, software-pipelined
// one-workgroup-per-CU variant (tools/kbench/experiments/kernels_bf16_rbp.hip) is correct and 15 % SLOWER: see the same file.
// Round 3 (profiles/r03_a_coissue_findings.md, r03_c_narrow_stage_findings.md): 16-row blocked LDS tiles at C = 32 / 64, register-resident
// weights and LDS-resident raw residual rows at C = 32, the MRF mean as a reciprocal multiply, no packed-f32 VALU.  Built, bit-correct and
// measured no faster, then removed again (the code is in the history up to commit a56dd21): the residual / MRF rows added by the matrix
// cores through identity A fragments (VTTS_RES_MFMA), an MFMA stream paced by s_nop (VTTS_PACE_NOP), column-half software pipelining of
// the epilogues, conv_post fused into the last pair launch.
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

#ifndef VTTS_XCD_MAP  // XCD-aware tile order (A/B switch, tools/kbench): see resblock_pair_g_bf16_k
#define VTTS_XCD_MAP 1
#endif

#ifndef VTTS_RAWRES  // C = 32: the tile's raw rows kept in LDS as the residual (A/B switch)
#define VTTS_RAWRES 1
#endif
#ifndef VTTS_WREG  // C = 32: the convolution's weights register-resident for the whole phase (A/B switch, tools/kbench)
#define VTTS_WREG 1
#endif
#ifndef VTTS_LEAN  // lean loop addressing (buffer loads with SGPR offsets, per-tap swizzle terms): A/B switch, tools/kbench
#define VTTS_LEAN 1
#endif

namespace vtts {

constexpr int XCD_MAP_MIN_TILES = 192;  // XCD-aware tile order from this many tiles per utterance slot on (resblock_pair_g_bf16_k)

template <int C_, int KS_, int N1_, int WM_, int WN_, int PA_, int MINWG_, int XC_ = C_>
struct GTile {
    static constexpr int C = C_, KS = KS_, N1 = N1_, WM = WM_, WN = WN_, PA = PA_, MINWG = MINWG_;
    static constexpr int XC = XC_, NXC = C / XC;        // the X tile is staged XC input channels at a time (C = 256: 2 x 128)
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int MR = C / WM / 32, NR = N1 / WN / 32;
    static constexpr int H2 = (KS - 1) / 2;             // c2 halo (rate 1); c1's is H2 * rate
    static constexpr int MAXDIL = 5;
    static constexpr int NT2 = N1 - 2 * H2;             // outputs per workgroup
    static constexpr int SPR1 = XC / 8, P1 = XC * 2;    // X tile (one channel chunk): 16-byte slots / bytes per row
    static constexpr int SPR2 = C / 8, P2 = C * 2;      // xt tile (all channels)
    static constexpr int ROWSX_MAX = N1 + 2 * H2 * MAXDIL;
    static constexpr int ROWST = N1 + 2 * H2;           // xt rows incl. the tail only discarded columns read
    static constexpr int KSTEPS = C / 16;               // k-steps per tap
    static constexpr int KSX = XC / 16;                 // ... of one X channel chunk
    static constexpr int NQT = KS * KSTEPS;             // k-steps per convolution
    static constexpr int MB = C / 32;
    static constexpr int RA = PA + 1;                   // A-fragment register ring (slots)
    static constexpr bool UNROLL_ALL = (KSX % RA) != 0;  // else: loop over taps, one tap's k-steps per iteration
    // C = 32: a convolution's whole A operand is NQT * MR fragments = at most 88 VGPRs per lane (k = 11), so it is loaded ONCE per phase,
    // ahead of it, and the MFMA loop carries no vector-memory instruction at all (see conv_phase_wreg)
    static constexpr bool WREG = VTTS_WREG && UNROLL_ALL && NXC == 1 && NQT * MR * 4 <= 96;
    static constexpr int XPT = (ROWSX_MAX * SPR1 + THREADS - 1) / THREADS;
    static constexpr size_t CONV_BYTES = (size_t)KS * C * C * 2;  // packed weights of one convolution
    static_assert(C % (WM * 32) == 0 && N1 % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(C % XC == 0 && (NXC == 1 || !UNROLL_ALL), "channel chunking");
    static_assert(SPR1 == 4 || SPR1 == 8 || SPR1 == 16, "X row pitch 64..256 B");
    static_assert(SPR2 == 4 || SPR2 == 8 || SPR2 == 16 || SPR2 == 32, "xt row pitch 64..512 B");
    // C = 32: the pair is bound by the CU's share of HBM (100 KB per tile at ~8 B per tick: profiles/r03_c_narrow_stage_findings.md), and a third
    // of those bytes is the residual's second read of rows the staging pass has just had in registers: it keeps them, raw, in a second LDS
    // region (32 KB; 69 KB with the X tile, still two workgroups per CU) and epilogue 2 adds them from there
    static constexpr bool RAWRES = VTTS_RAWRES && NXC == 1 && C == 32;
    static constexpr int RAW_BYTES = RAWRES ? tile_rows16(N1) * P1 : 0;
    static __host__ __device__ constexpr int tile_bytes(int dil) {  // rows in multiples of 16 (tile_off's blocks of 16 rows at C = 32 / 64)
        const int bx = tile_rows16(N1 + 2 * H2 * dil) * P1, bt = tile_rows16(ROWST) * P2;
        return bx > bt ? bx : bt;
    }
    // TAIL (GTail below): the generator's tail — conv_post (32 -> 1, k = 7) + tanh — run by this kernel on its own output rows (stage 4's last pair)
    static constexpr bool TAIL = false;
    static constexpr int TAIL_KS = 7, TAIL_H = 3;
    static constexpr int TAIL_BYTES = 0;
    static int lds_bytes(int dil) { return tile_bytes(dil) + RAW_BYTES; }
    static_assert(tile_rows16(ROWSX_MAX) * P1 + RAW_BYTES <= 80 * 1024 || !RAWRES, "two workgroups per CU");
    static_assert(tile_rows16(ROWSX_MAX) * P1 <= 160 * 1024 && tile_rows16(ROWST) * P2 <= 160 * 1024, "LDS budget");
};

// The stage-4 tail (round 5; asked for since round 2): the LAST pair launch of the generator's last stage — the one whose epilogue already forms the MRF
// mean and applies the tail's LeakyReLU(0.01) (model.py:121-122) — keeps its bf16 output rows in LDS (in place of the raw residual rows it has just
// consumed there), and after one barrier computes conv_post + tanh (model.py:123-124) from them: the stage output (1.07 GB at 64 x 1024 frames) is never
// written, conv_post_bf16_k's launch and its re-read are gone.  A tile then yields NT2 - 6 samples (conv_post's halo of 3 per side comes out of the tile's
// own rows: its stride shrinks by 6 and tile 0 starts at time -3, whose rows are the reference's zero padding).  The values conv_post sees are the bf16 values
// the un-fused path stores, its fmaf chain runs in conv_post_bf16_k's order (tap-major, channels ascending): bit-identical samples
// (tests/test_gpu_bf16.py::test_stage4_tail_fusion_is_bit_identical).
template <class B>
struct GTail : B {
    static constexpr bool TAIL = true;
    static constexpr int TAIL_BYTES = 1024;  // conv_post's 7 x 32 fp32 weights behind the residual region
    static int lds_bytes(int dil) { return B::tile_bytes(dil) + B::RAW_BYTES + TAIL_BYTES; }
    static_assert(B::RAWRES && B::C == 32 && B::WM == 1, "the tail lives on the C = 32 tile with the LDS-resident residual rows");
    static_assert(B::tile_bytes(B::MAXDIL) + B::RAW_BYTES + TAIL_BYTES <= 80 * 1024, "two workgroups per CU");
};

template <class T>
__global__ __launch_bounds__(T::THREADS, (T::MINWG * T::THREADS + 255) / 256) void resblock_pair_g_bf16_k(BConvArgs a) {
    constexpr int C = T::C, KS = T::KS, N1 = T::N1, WN = T::WN, PA = T::PA, RA = T::RA;
    constexpr int THREADS = T::THREADS, MR = T::MR, NR = T::NR, H2 = T::H2, NT2 = T::NT2;
    constexpr int SPR1 = T::SPR1, P1 = T::P1, SPR2 = T::SPR2, P2 = T::P2, XC = T::XC, NXC = T::NXC, KSX = T::KSX;
    constexpr int KSTEPS = T::KSTEPS, NQT = T::NQT, MB = T::MB, XPT = T::XPT;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xt = lds;  // X tile, later the xt tile
    [[maybe_unused]] unsigned char* const xraw = lds + (T::RAWRES ? T::tile_bytes(a.dil) : 0);  // T::RAWRES: raw rows t0 .. t0 + N1 - 1 (the residual)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int Lp = a.L;                                      // rows allocated per utterance
    const int L = a.lens ? min(max(a.lens[b], 0) * a.len_mul, a.L) : a.L;  // valid rows of this utterance, clamped to its slot (ragged batch: the rest reads as zero padding)
#if VTTS_XCD_MAP
    // Workgroups go to the 8 XCDs round-robin in launch order (workgroup w -> XCD w % 8, each with its own L2), so consecutive
    // blockIdx.x would put every tile's neighbours — whose halo rows it shares — on OTHER L2s.  gridDim.x is padded to a
    // multiple of 8 and an XCD takes a contiguous eighth of THIS utterance's valid tiles, in order (a tile's left halo was staged by
    // the same XCD's previous workgroup moments ago).  Eighths of the VALID tiles, balanced to within one tile and rotated from
    // utterance to utterance: with ceil(nt / 8) per XCD a batch of short utterances (nt = 9: 2, 2, 2, 2, 1, 0, 0, 0) left the
    // same XCDs idle for every utterance (+4 % on the 256-sentence pipeline).
    // Only launches whose utterance slots hold at least XCD_MAP_MIN_TILES tiles do this (the launcher pads gridDim.x for exactly those):
    // with a few tiles per eighth the saving is small, and a padded grid sends tile t of EVERY utterance to XCD t % 8, which on the
    // 256-sentence pipeline (15-140 tiles per utterance and stage) left the last XCDs a tile short per utterance: 2-4 % slower.
    constexpr int TS = T::TAIL ? NT2 - 2 * T::TAIL_H : NT2;  // tile stride (TAIL: conv_post's halo comes out of the tile's own rows)
    const bool xmap = (a.L + TS - 1) / TS >= XCD_MAP_MIN_TILES;
    const int nt = (L + TS - 1) / TS, r = (int)((blockIdx.x + b) & 7), lo = (r * nt) >> 3, hi = ((r + 1) * nt) >> 3;
    const int tile = xmap ? lo + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (xmap && tile >= hi) return;
#else
    constexpr int TS = T::TAIL ? NT2 - 2 * T::TAIL_H : NT2;
    const int tile = blockIdx.x;
#endif
    const int t0 = tile * TS - (T::TAIL ? T::TAIL_H : 0);   // first output time step (row 0 of c2's output) of this workgroup
    if (t0 + (T::TAIL ? T::TAIL_H : 0) >= L) return;        // a tile past this utterance's end: nothing reads its rows
    const int dil = a.dil;
    const int h1 = H2 * dil;                 // c1's symmetric pad (model.py:8-10)
    const int rowsx = N1 + 2 * h1;           // X rows: times t0 - H2 - h1 ... t0 - H2 - h1 + rowsx - 1
    const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + (size_t)b * Lp * C;
    [[maybe_unused]] const int wg_lin = blockIdx.z * gridDim.x + blockIdx.x;
    VTTS_TL_ID(a, wg_lin);
    VTTS_TL(a, wg_lin, 0);

    // Accumulators start from the bias: the first k-step's MFMAs take a 16-register bias block (row = channel
    // 32*mr + 8*rq + 4*lh + i of this wave's m-block, r = 4*rq + i; the same for every column block) as their C operand, so
    // there is no accumulator initialisation at all (it was 128 v_mov per convolution and wave, on the issue port the
    // co-resident workgroup's MFMAs need).
    f32x16 acc[MR][NR];
    f32x16 bblk[MR];
    auto load_bias = [&](const float* __restrict__ bias) {  // requested a phase ahead of the MFMAs that consume it
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const float* __restrict__ bp = bias + wm * (C / T::WM) + mr * 32 + 4 * lh;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(bp), q1 = *reinterpret_cast<const f32x4*>(bp + 8);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(bp + 16), q3 = *reinterpret_cast<const f32x4*>(bp + 24);
            const f32x8 lo = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7);
            bblk[mr] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        }
    };
    load_bias(a.bias);
    if constexpr (T::TAIL) {  // conv_post's weights (Haiku [K][Cin][1] fp32) behind the residual region; visible after the staging barrier
        float* tw = reinterpret_cast<float*>(lds + T::tile_bytes(dil) + T::RAW_BYTES);
        for (int i = tid; i < T::TAIL_KS * C; i += THREADS) tw[i] = a.tail_wf[i];
    }

    // ---------------- C = 32: register-resident weights ----------------
    // What the per-workgroup timelines show at C = 32 (tools/kbench, gpurun_out/r03_exp8/timeline.txt): the loops take 11.4k + 7.5k
    // cycles for 2 x 2.8k cycles of MFMA — they crawl whenever the co-resident workgroup is in a memory phase, because every k-step
    // (only 4 MFMAs = 128 cycles at one m-block) waits for a weight fragment that queues behind the partner's 36 KB of staging loads
    // in the CU's one vector-memory pipeline; a 3-step ring is 384 cycles of cover against 5k-cycle bursts.  All of a convolution's
    // fragments fit in registers here, so they are requested in ONE burst a phase ahead (c1's before the X tile is staged, c2's
    // before epilogue 1) and the loop is ds_read + MFMA only.
    constexpr int NQW = T::WREG ? NQT : 1;
    bf16x8 aw[NQW][MR];
    auto load_w_all = [&](const unsigned char* __restrict__ wconv) {
        if constexpr (T::WREG) {
            const uint4* __restrict__ ap = reinterpret_cast<const uint4*>(wconv) + (size_t)(wm * MR) * 64 + lane;
#pragma unroll
            for (int q = 0; q < NQT; ++q)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) aw[q][mr] = __builtin_bit_cast(bf16x8, ap[(size_t)(q * MB + mr) * 64]);
        }
    };
    load_w_all(static_cast<const unsigned char*>(a.wp));

    // ---------------- X tile (channel chunk xc): LeakyReLU + zero padding in registers, swizzled ds_write_b128 ----------------
    // XB = loads in flight per thread: all of them for the first chunk (nothing else is live yet), a few at a time for
    // a later chunk (the accumulators are live: 128 VGPRs)
    auto stage_x = [&](int xc, auto xb_tag) {
        constexpr int XB = decltype(xb_tag)::value;
        constexpr int RPI = THREADS / SPR1;  // tile rows between a thread's consecutive 16-byte units
        static_assert(THREADS % SPR1 == 0 && RPI % 16 == 0, "a thread's units share their column and their swizzle");
        const int tx0 = t0 - H2 - h1;
        // a wave loads 64 / SPR1 whole rows (1 KiB contiguous).  Row-major tiles (SPR1 >= 16): lane -> (row, slot) in memory order.
        // Blocked tiles (SPR1 = 4, 8): 8 consecutive lanes take 8 consecutive rows of one slot, which is what keeps the
        // ds_write_b128 conflict-free there (tile_off); the wave touches the same cache lines either way.
        constexpr int RW = 64 / SPR1;  // rows per wave and unit
        const int row0 = (SPR1 >= 16 || !VTTS_TILE_BLOCKED) ? tid / SPR1 : wave * RW + lane % RW, c = (SPR1 >= 16 || !VTTS_TILE_BLOCKED) ? tid % SPR1 : lane / RW;
        auto act2 = [](unsigned u) { return lrelu01_pack(bf16_lo(u), bf16_hi(u)); };  // LRELU_SLOPE, model.py:5,46
        unsigned char* const lds0 = xt + tile_off<SPR1>(row0, c);  // unit i: + i * RPI * P1 (RPI is a multiple of 16: same swizzle / same place in its block)
        // T::RAWRES: X-tile row r holds time t0 - H2 - h1 + r; rows of times t0 .. t0 + N1 - 1 also go, un-activated, to the residual region
        [[maybe_unused]] auto keep_raw = [&](int r, int cc, const uint4& raw, bool live) {
            const int rr = r - H2 - h1;
            if (live && rr >= 0 && rr < N1) *reinterpret_cast<uint4*>(xraw + tile_off<SPR1>(rr, cc)) = raw;
        };
        if (tx0 >= 0 && tx0 + XPT * RPI <= L) {
            // interior tile (all but the first / last of an utterance): no clamping, no masking, constant strides
            const unsigned short* __restrict__ g0 = xg + (size_t)(tx0 + row0) * C + xc * XC + c * 8;
#pragma unroll
            for (int i0 = 0; i0 < XPT; i0 += XB) {
                uint4 v[XB];
#pragma unroll
                for (int i = 0; i < XB; ++i)
                    if (i0 + i < XPT) v[i] = *reinterpret_cast<const uint4*>(g0 + (size_t)(i0 + i) * RPI * C);
#if VTTS_TIMELINE
                if (i0 == 0) {
                    VTTS_TL(a, wg_lin, 7);
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    VTTS_TL(a, wg_lin, 8);
                }
#endif
                if constexpr (T::RAWRES) {
#pragma unroll
                    for (int i = 0; i < XB; ++i) keep_raw(row0 + (i0 + i) * RPI, c, v[i], i0 + i < XPT);
                }
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    v[i].x = act2(v[i].x);
                    v[i].y = act2(v[i].y);
                    v[i].z = act2(v[i].z);
                    v[i].w = act2(v[i].w);
                }
#pragma unroll
                for (int i = 0; i < XB; ++i)
                    if (i0 + i < XPT && row0 + (i0 + i) * RPI < rowsx) *reinterpret_cast<uint4*>(lds0 + (i0 + i) * RPI * P1) = v[i];
            }
            return;
        }
#pragma unroll
        for (int i0 = 0; i0 < XPT; i0 += XB) {
            uint4 v[XB];
            bool okx[XB];
            // unconditional loads from clamped addresses, masked afterwards: a load under a per-element branch makes
            // hipcc wait for each one before issuing the next
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                const int t = tx0 + row0 + (i0 + i) * RPI;
                okx[i] = i0 + i < XPT && row0 + (i0 + i) * RPI < rowsx && t >= 0 && t < L;
                const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
                v[i] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + xc * XC + c * 8);
            }
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                if (!okx[i]) v[i] = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (T::RAWRES) keep_raw(row0 + (i0 + i) * RPI, c, v[i], i0 + i < XPT);
                v[i].x = act2(v[i].x);
                v[i].y = act2(v[i].y);
                v[i].z = act2(v[i].z);
                v[i].w = act2(v[i].w);
            }
#pragma unroll
            for (int i = 0; i < XB; ++i)
                if (i0 + i < XPT && row0 + (i0 + i) * RPI < rowsx) *reinterpret_cast<uint4*>(lds0 + (i0 + i) * RPI * P1) = v[i];
        }
    };
    stage_x(0, std::integral_constant<int, XPT>{});
    VTTS_TL(a, wg_lin, 9);
    __syncthreads();  // B1: X tile staged
    VTTS_TL(a, wg_lin, 1);

    // ---- one convolution pass over the LDS tile: acc += W[:, chunk] (*) tile -----------------------------------------
    // The tile holds NKS k-steps (16 channels each) per row, channels 16*ks0 .. of the convolution's input.
    // A fragment (tap, ks, mr): 16 bytes per lane at  wconv + (((tap*KSTEPS + ks0 + ks)*MB + wm*MR + mr)*64 + lane)*16
    // B fragment (tap, ks, nr): tile row  n + tap*dl  (n = this lane's output column), 16-byte slot 2*ks + lh of that row
    const int rowbase0 = wn * (N1 / WN) + l31;
    auto conv_phase = [&](const unsigned char* __restrict__ wconv, int dl, auto sprb_tag, auto nks_tag, int ks0, auto fresh_tag) {
        constexpr int SPRB = decltype(sprb_tag)::value, PB = SPRB * 16, NKS = decltype(nks_tag)::value;
        constexpr bool FRESH = decltype(fresh_tag)::value;  // the pass's first step starts the accumulators from the bias block
        constexpr int NSTEPS = KS * NKS;
        const uint4* __restrict__ aptr = reinterpret_cast<const uint4*>(wconv) + ((size_t)ks0 * MB + wm * MR) * 64 + lane;
        bf16x8 af[RA][MR], bf[2][NR];
        auto load_a = [&](int tap, int ks, int slot) {  // (tap, ks) may run past the end: re-read the last step (stays inside the blob)
            const int tc = tap < KS ? tap : KS - 1;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
                af[slot][mr] = __builtin_bit_cast(bf16x8, aptr[(size_t)((tc * KSTEPS + ks) * MB + mr) * 64]);
        };
        auto load_b = [&](int tap, int ks, int par) {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int row = rowbase0 + tap * dl + nr * 32;
                bf[par][nr] = *reinterpret_cast<const bf16x8*>(xt + tile_off<SPRB>(row, ks * 2 + lh));
            }
        };
        auto mfma_step = [&](int slot, int par, bool first) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[slot][mr], bf[par][nr], first ? bblk[mr] : acc[mr][nr], 0, 0, 0);

                }
        };
        // keep hipcc's scheduler from sinking the look-ahead loads to their uses (it does, to save registers), and spread
        // them between the MFMAs: grouped issue (all loads, then all MFMAs) left the matrix pipe idle while a lone wave
        // issued its 6 memory instructions (workgroup alone on a CU: 27.2k -> 26.0k cycles per 704-MFMA loop; whole
        // forward +2.1 %, profiles/r01_j_kbench_findings.md)
        auto pin_step = [&](bool has_b) {
            constexpr int NM = MR * NR;
            const int mem = MR + (has_b ? NR : 0);
            int done = 0;
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA of this step
                const int upto = (i + 1) * mem / NM;
                for (; done < upto; ++done) {
                    if (done < MR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read: an A fragment PA steps ahead
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // DS read: a B fragment of the next step
                }
            }
        };
        static_assert(T::UNROLL_ALL || PA <= NKS, "look-ahead within two taps");
#pragma unroll
        for (int s = 0; s < PA; ++s) load_a(s / NKS, s % NKS, s % RA);
        load_b(0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, MR * PA, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
        if constexpr (T::UNROLL_ALL) {
#pragma unroll
            for (int s = 0; s < NSTEPS; ++s) {
                load_a((s + PA) / NKS, (s + PA) % NKS, (s + PA) % RA);
                if (s + 1 < NSTEPS) load_b((s + 1) / NKS, (s + 1) % NKS, (s + 1) & 1);
                mfma_step(s % RA, s & 1, FRESH && s == 0);
                pin_step(s + 1 < NSTEPS);
            }
        } else {
            // rolled loop over blocks of UB k-steps (flat step index s = tap*NKS + ks): the ring slot and the B parity
            // of a step depend only on its position in the block
            constexpr int UB = NKS < 8 ? NKS : 8;
            static_assert(T::UNROLL_ALL || (UB % RA == 0 && UB % 2 == 0 && NKS % UB == 0), "ring slot / B parity must be compile-time in the block loop");
#if VTTS_LEAN
            // Lean addressing of the rolled loop (round 2): the loop's own address arithmetic shares the SIMD's issue port with
            // the co-resident workgroup's staging / epilogue VALU work.  A fragments: buffer loads, lane offset in a VGPR, the
            // k-step's offset in an SGPR, the m-block an immediate.  B fragments: one row address + one swizzle term per tap
            // (a block never straddles taps), (xs ^ ks << 5) + row address per k-step (v_xad_u32), the column block an immediate.
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wconv), 0, (int)T::CONV_BYTES, 0x00020000);
            const unsigned a_voff = (unsigned)((wm * MR) * 64 + lane) * 16;
            auto load_a2 = [&](int sa, int slot) {  // flat step sa of this pass (may run past the end: re-read the last step)
                const int sc = sa < NSTEPS ? sa : NSTEPS - 1;
                const int tap = sc / NKS, ks = sc - tap * NKS;
                const int soff = ((tap * KSTEPS + ks0 + ks) * MB) * 1024;
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + mr * 1024, soff, 0);
                    af[slot][mr] = __builtin_bit_cast(bf16x8, v);
                }
            };
            auto tap_terms = [&](int tap, unsigned& tapaddr, unsigned& xs) {
                const int row = rowbase0 + tap * dl;
                if constexpr (SPRB >= 16 || !VTTS_TILE_BLOCKED) {
                    tapaddr = (unsigned)row * PB;
                    xs = (unsigned)(swz_of<SPRB>(row) ^ lh) << 4;
                } else {  // blocked tile: slot 2*ks + lh is 512*ks + 256*lh bytes into the row's block
                    tapaddr = (unsigned)tile_off<SPRB>(row, lh);
                    xs = 0;
                }
            };
            auto load_b2 = [&](unsigned tapaddr, unsigned xs, int ks, int par) {
                const unsigned addr = (SPRB >= 16 || !VTTS_TILE_BLOCKED) ? tapaddr + (xs ^ (unsigned)(ks << 5)) : tapaddr + (unsigned)(ks << 9);
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bf[par][nr] = *reinterpret_cast<const bf16x8*>(xt + addr + nr * 32 * PB);
            };
            auto block = [&](int s0, auto first_tag) {  // the first block of a fresh pass is peeled: its first step reads the bias block
                const int tap = s0 / NKS, ksb = s0 - tap * NKS;  // this block's tap and first k-step in it
                unsigned ta, xs, tn, xn;
                tap_terms(tap, ta, xs);
                const bool wrap = ksb + UB >= NKS;                // the block's last look-ahead B fragment is the next tap's first
                const int tapn = tap + 1 < KS ? tap + 1 : KS - 1;
                tap_terms(wrap ? tapn : tap, tn, xn);
                const int ksn = wrap ? 0 : ksb + UB;
#pragma unroll
                for (int i = 0; i < UB; ++i) {
                    load_a2(s0 + i + PA, (i + PA) % RA);
                    if (i + 1 < UB) load_b2(ta, xs, ksb + i + 1, (i + 1) & 1);
                    else load_b2(tn, xn, ksn, (i + 1) & 1);
                    mfma_step(i % RA, i & 1, decltype(first_tag)::value && i == 0);
                    pin_step(true);
                }
            };
#else
            auto block = [&](int s0, auto first_tag) {  // the first block of a fresh pass is peeled: its first step reads the bias block
#pragma unroll
                for (int i = 0; i < UB; ++i) {
                    const int sa = s0 + i + PA, sb = s0 + i + 1;
                    load_a(sa / NKS, sa % NKS, (i + PA) % RA);
                    load_b(sb < NSTEPS ? sb / NKS : KS - 1, sb < NSTEPS ? sb % NKS : 0, (i + 1) & 1);
                    mfma_step(i % RA, i & 1, decltype(first_tag)::value && i == 0);
                    pin_step(true);
                }
            };
#endif
            block(0, std::integral_constant<bool, FRESH>{});
#pragma nounroll
            for (int s0 = UB; s0 < NSTEPS; s0 += UB) block(s0, std::false_type{});
        }
    };

    // the same pass with the A operand in registers (T::WREG): one LDS fragment read per MFMA, nothing else in the loop
    auto conv_phase_wreg = [&](int dl, auto sprb_tag) {
        constexpr int SPRB = decltype(sprb_tag)::value;
        bf16x8 bf[2][NR];
        auto load_b = [&](int q, int par) {
            const int tap = q / KSTEPS, ks = q % KSTEPS;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) bf[par][nr] = *reinterpret_cast<const bf16x8*>(xt + tile_off<SPRB>(rowbase0 + tap * dl + nr * 32, ks * 2 + lh));
        };
        load_b(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
        for (int q = 0; q < NQT; ++q) {
            if (q + 1 < NQT) load_b(q + 1, (q + 1) & 1);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q < NQW ? q : 0][mr], bf[q & 1][nr], q == 0 ? bblk[mr] : acc[mr][nr], 0, 0, 0);
            for (int i = 0; i < MR * NR; ++i) {  // one fragment read behind each MFMA
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q + 1 < NQT && i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };

    // ---------------- phase 1: xt = c1(lrelu(x)); column n <-> xt time t0 - H2 + n; tap j reads X row n + j*dil ----------------
#pragma unroll
    for (int xc = 0; xc < NXC; ++xc) {
        if (xc > 0) {
            __syncthreads();  // every wave is done reading the previous channel chunk
            stage_x(xc, std::integral_constant<int, 4>{});
            __syncthreads();
        }
        if constexpr (T::WREG)
            conv_phase_wreg(dil, std::integral_constant<int, SPR1>{});
        else if (xc == 0)
            conv_phase(static_cast<const unsigned char*>(a.wp), dil, std::integral_constant<int, SPR1>{}, std::integral_constant<int, KSX>{}, 0, std::true_type{});
        else
            conv_phase(static_cast<const unsigned char*>(a.wp), dil, std::integral_constant<int, SPR1>{}, std::integral_constant<int, KSX>{}, xc * KSX, std::false_type{});
    }
    VTTS_TL(a, wg_lin, 2);
    load_w_all(static_cast<const unsigned char*>(a.wp) + T::CONV_BYTES);  // T::WREG: c2's fragments, in flight under epilogue 1
    __syncthreads();  // B2: every wave is done reading the X tile

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
                    unsigned p0 = lrelu01_pack(acc[mr][nr][r0 + 0], acc[mr][nr][r0 + 1]);
                    unsigned p1 = lrelu01_pack(acc[mr][nr][r0 + 2], acc[mr][nr][r0 + 3]);
                    unsigned q0 = lrelu01_pack(acc[mr][nr][r0 + 4], acc[mr][nr][r0 + 5]);
                    unsigned q1 = lrelu01_pack(acc[mr][nr][r0 + 6], acc[mr][nr][r0 + 7]);
                    if (!ok) p0 = p1 = q0 = q1 = 0u;  // c2's own zero padding applies to xt
                    swap_pair(p0, q0);
                    swap_pair(p1, q1);
                    const int slot = (cb >> 3) + lh;
                    *reinterpret_cast<uint4*>(xt + tile_off<SPR2>(row, slot)) = make_uint4(p0, p1, q0, q1);
                }
            }
        }
        // rows N1 .. N1 + 2*H2 - 1 are only read by the discarded output columns: keep them finite
        for (int u = tid; u < 2 * H2 * SPR2; u += THREADS) {
            const int row = N1 + u % (2 * H2), c = u / (2 * H2);  // consecutive lanes: consecutive rows of one slot (conflict-free in the blocked tiles)
            *reinterpret_cast<uint4*>(xt + tile_off<SPR2>(row, c)) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();  // B3: xt tile written
    VTTS_TL(a, wg_lin, 3);

    // ---------------- phase 2: c2 over the xt tile (rate 1): column n <-> time t0 + n, tap j reads xt row n + j ----------------
    if constexpr (T::WREG)
        conv_phase_wreg(1, std::integral_constant<int, SPR2>{});
    else
        conv_phase(static_cast<const unsigned char*>(a.wp) + T::CONV_BYTES, 1, std::integral_constant<int, SPR2>{}, std::integral_constant<int, KSTEPS>{}, 0, std::true_type{});
    VTTS_TL(a, wg_lin, 4);

    // ---------------- epilogue 2: + x [MRF accumulate / mean] [consumer's LeakyReLU] -> bf16, 16-byte stores ----------------
    {
        const float s_out = a.slope_out;
        const float dv = a.div;
        const float rdv = mrf_recip(dv);
        unsigned short* __restrict__ yg = static_cast<unsigned short*>(a.y) + (size_t)b * Lp * C;
        // rows of a [B][L][C] tensor in the swapped accumulator layout: all requests first, one wait
        auto add_rows = [&](const unsigned short* __restrict__ src) {
            uint4 rv[MR][2][NR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int t = t0 + wn * (N1 / WN) + nr * 32 + l31;
                        const int tc = t < 0 ? 0 : (t < L ? t : L - 1);  // rows outside the utterance are never stored: any in-bounds address will do
                        rv[mr][p][nr] = *reinterpret_cast<const uint4*>(src + (size_t)tc * C + wm * (C / T::WM) + mr * 32 + 16 * p + 8 * lh);
                    }
#if VTTS_TIMELINE
            VTTS_TL(a, wg_lin, 10);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            VTTS_TL(a, wg_lin, 11);
#endif
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        uint4 r = rv[mr][p][nr];
                        swap_pair(r.x, r.z);  // un-swap the chunk into the accumulator layout
                        swap_pair(r.y, r.w);
                        const int r0 = 8 * p;
                        acc[mr][nr][r0 + 0] = vadd_raw(bf16_lo(r.x), acc[mr][nr][r0 + 0]); acc[mr][nr][r0 + 1] = vadd_raw(bf16_hi(r.x), acc[mr][nr][r0 + 1]);
                        acc[mr][nr][r0 + 2] = vadd_raw(bf16_lo(r.y), acc[mr][nr][r0 + 2]); acc[mr][nr][r0 + 3] = vadd_raw(bf16_hi(r.y), acc[mr][nr][r0 + 3]);
                        acc[mr][nr][r0 + 4] = vadd_raw(bf16_lo(r.z), acc[mr][nr][r0 + 4]); acc[mr][nr][r0 + 5] = vadd_raw(bf16_hi(r.z), acc[mr][nr][r0 + 5]);
                        acc[mr][nr][r0 + 6] = vadd_raw(bf16_lo(r.w), acc[mr][nr][r0 + 6]); acc[mr][nr][r0 + 7] = vadd_raw(bf16_hi(r.w), acc[mr][nr][r0 + 7]);
                    }
        };
        if constexpr (T::RAWRES) {                                          // x = xt + x        (model.py:50), rows from the LDS copy
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int col = wn * (N1 / WN) + nr * 32 + l31;
                        uint4 r = *reinterpret_cast<const uint4*>(xraw + tile_off<SPR1>(col, (wm * (C / T::WM) + mr * 32 + 16 * p) / 8 + lh));
                        swap_pair(r.x, r.z);
                        swap_pair(r.y, r.w);
                        const int r0 = 8 * p;
                        acc[mr][nr][r0 + 0] = vadd_raw(bf16_lo(r.x), acc[mr][nr][r0 + 0]); acc[mr][nr][r0 + 1] = vadd_raw(bf16_hi(r.x), acc[mr][nr][r0 + 1]);
                        acc[mr][nr][r0 + 2] = vadd_raw(bf16_lo(r.y), acc[mr][nr][r0 + 2]); acc[mr][nr][r0 + 3] = vadd_raw(bf16_hi(r.y), acc[mr][nr][r0 + 3]);
                        acc[mr][nr][r0 + 4] = vadd_raw(bf16_lo(r.z), acc[mr][nr][r0 + 4]); acc[mr][nr][r0 + 5] = vadd_raw(bf16_hi(r.z), acc[mr][nr][r0 + 5]);
                        acc[mr][nr][r0 + 6] = vadd_raw(bf16_lo(r.w), acc[mr][nr][r0 + 6]); acc[mr][nr][r0 + 7] = vadd_raw(bf16_hi(r.w), acc[mr][nr][r0 + 7]);
                    }
        } else {
            add_rows(xg);                                                   // x = xt + x        (model.py:50)
        }
        if (a.acc_add != 0) add_rows(yg);                                   // xs += rb(x)       (model.py:118-120)
        VTTS_TL(a, wg_lin, 12);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int cb = wm * (C / T::WM) + mr * 32 + 16 * p;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int row = wn * (N1 / WN) + nr * 32 + l31;
                    const int t = t0 + row;
                    const bool ok = row < NT2 && t >= 0 && t < L;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[mr][nr][8 * p + e];
                    if (dv != 1.0f) {  // x = xs / num_kernels  (model.py:121)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = VTTS_MRF_DIV ? v[e] / dv : v[e] * rdv;
                    }
                    if (s_out != 1.0f) {  // the (only) consumer's LeakyReLU, applied once by the producer
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = lrelu_f(v[e], s_out);
                    }
                    unsigned p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
                    unsigned q0 = pack_bf16x2(v[4], v[5]), q1 = pack_bf16x2(v[6], v[7]);
                    swap_pair(p0, q0);
                    swap_pair(p1, q1);
                    if constexpr (T::TAIL) {
                        // the row stays in LDS, where this lane has just read its residual values (same row, same slot: no other lane touches it);
                        // rows outside the utterance are conv_post's zero padding
                        const uint4 o = ok ? make_uint4(p0, p1, q0, q1) : make_uint4(0u, 0u, 0u, 0u);
                        if (row < NT2) *reinterpret_cast<uint4*>(xraw + tile_off<SPR1>(row, (cb >> 3) + lh)) = o;
                    } else {
                        if (ok) *reinterpret_cast<uint4*>(yg + (size_t)t * C + cb + 8 * lh) = make_uint4(p0, p1, q0, q1);
                    }
                }
            }
        }
    }
    if constexpr (T::TAIL) {
        // ---------------- tail: tanh(conv_post(.)) over the rows just written (model.py:123-124); conv_post_bf16_k's arithmetic, in its order ----------------
        __syncthreads();
        const float* tw = reinterpret_cast<const float*>(lds + T::tile_bytes(dil) + T::RAW_BYTES);
        const float tb = a.tail_bias[0];
        for (int n = tid; n < TS; n += THREADS) {
            const int t = t0 + T::TAIL_H + n;  // output sample: reads rows n .. n + 6 = times t - 3 .. t + 3
            if (t >= L) break;
            float accp = 0.f;
#pragma unroll
            for (int j = 0; j < T::TAIL_KS; ++j) {
#pragma unroll
                for (int c = 0; c < SPR1; ++c) {
                    const uint4 v = *reinterpret_cast<const uint4*>(xraw + tile_off<SPR1>(n + j, c));
                    const float* w = tw + j * C + c * 8;
                    accp = fmaf(w[0], bf16_lo(v.x), accp); accp = fmaf(w[1], bf16_hi(v.x), accp);
                    accp = fmaf(w[2], bf16_lo(v.y), accp); accp = fmaf(w[3], bf16_hi(v.y), accp);
                    accp = fmaf(w[4], bf16_lo(v.z), accp); accp = fmaf(w[5], bf16_hi(v.z), accp);
                    accp = fmaf(w[6], bf16_lo(v.w), accp); accp = fmaf(w[7], bf16_hi(v.w), accp);
                }
            }
            a.tail_wav[(size_t)b * Lp + t] = tanhf(accp + tb);
        }
    }
    VTTS_TL(a, wg_lin, 6);
}

// ---- tile table -------------------------------------------------------------------------------------
//                                       C   KS   N1  WM WN PA MINWG
#ifndef VTTS_EXP_LDS_PAD  // kernel-development switch (tools/kbench): extra LDS per workgroup, e.g. 20000 = one workgroup per CU
#define VTTS_EXP_LDS_PAD 0
#endif
#ifndef VTTS_G64_N1  // tile-geometry experiments (tools/kbench): time steps per tile and workgroups per CU of the narrow stages
#define VTTS_G64_N1 512
#define VTTS_G64_WG 2
#endif
#ifndef VTTS_G32_N1
#define VTTS_G32_N1 512
#define VTTS_G32_WG 2
#endif
#ifndef VTTS_G128K3_N1
#define VTTS_G128K3_N1 256
#define VTTS_G128K3_WG 2
#endif
template <int KS> using G128 = GTile<128, KS, KS == 3 ? VTTS_G128K3_N1 : 256, 2, 2, 3, KS == 3 ? VTTS_G128K3_WG : 2>;
template <int KS> using G64 = GTile<64, KS, VTTS_G64_N1, 1, 4, 3, VTTS_G64_WG>;
#ifndef VTTS_G32_PA
#define VTTS_G32_PA 3
#endif
template <int KS> using G32 = GTile<32, KS, VTTS_G32_N1, 1, 4, VTTS_G32_PA, VTTS_G32_WG>;
template <int KS> using G256 = GTile<256, KS, 128, 4, 1, 3, 2, 128>;
// narrow tiles for SMALL launches (batch-1 latency: a 512-frame utterance is 35 wide tiles at C = 256, 134 at C = 128 — a
// fraction of the 512 workgroup slots): half the time steps per workgroup, twice the workgroups.  Same per-element
// accumulation order (the k loop), so the samples are bit-identical to the wide tiles'.
template <int KS> using G256S = GTile<256, KS, 64, 4, 1, 3, 2, 128>;
template <int KS> using G128S = GTile<128, KS, 128, 2, 2, 3, 2>;
template <int KS> using G64S = GTile<64, KS, 256, 1, 4, 3, 2>;
template <int KS> using G32S = GTile<32, KS, 256, 1, 4, 3, 2>;
constexpr long G_MIN_WGS = 384;  // below this many wide-tile workgroups (1.5 per CU slot pair) the narrow tile is launched
template <class T>
static hipError_t launch_g(const BConvArgs& a, hipStream_t s) {
    static DynLdsOnce once;  // per device (vtts_internal.h)
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&resblock_pair_g_bf16_k<T>), T::lds_bytes(T::MAXDIL) + VTTS_EXP_LDS_PAD, once); e != hipSuccess) return e;
    if (a.dil < 1 || a.dil > T::MAXDIL) return hipErrorInvalidValue;
    constexpr int TS = T::TAIL ? T::NT2 - 2 * T::TAIL_H : T::NT2;
    dim3 grid((a.L + TS - 1) / TS, 1, a.B);
#if VTTS_XCD_MAP
    if ((int)grid.x >= XCD_MAP_MIN_TILES) grid.x = (grid.x + 7) / 8 * 8;  // whole rounds of the 8 XCDs; a tile index past the utterance exits at once
#endif
    hipLaunchKernelGGL(resblock_pair_g_bf16_k<T>, grid, dim3(T::THREADS), T::lds_bytes(a.dil) + VTTS_EXP_LDS_PAD, s, a);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_g_ks(const BConvArgs& a, int K, hipStream_t s) {
    switch (K) {
        case 3: return launch_g<TT<3>>(a, s);
        case 7: return launch_g<TT<7>>(a, s);
        case 11: return launch_g<TT<11>>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pair_g_bf16(int C, int K, const BConvArgs& a, hipStream_t s) {
    // narrow tiles for small launches (a.tile_pref: 1 forces the wide tile, 2 the narrow one)
    auto narrow = [&](int nt2) { return a.tile_pref == 2 || (a.tile_pref != 1 && (long)((a.L + nt2 - 1) / nt2) * a.B < G_MIN_WGS); };
    switch (C) {
        case 256: return narrow(G256<11>::NT2) ? launch_g_ks<G256S>(a, K, s) : launch_g_ks<G256>(a, K, s);
        case 128: return narrow(G128<11>::NT2) ? launch_g_ks<G128S>(a, K, s) : launch_g_ks<G128>(a, K, s);
        case 64: return narrow(G64<11>::NT2) ? launch_g_ks<G64S>(a, K, s) : launch_g_ks<G64>(a, K, s);
        case 32:
            if (a.tail_wav) {  // the stage-4 tail rides on this launch (engine.hip: only where pair_tail_bf16_supported)
                if (K != 11) return hipErrorInvalidValue;
                return narrow(G32<11>::NT2) ? launch_g<GTail<G32S<11>>>(a, s) : launch_g<GTail<G32<11>>>(a, s);
            }
            return narrow(G32<11>::NT2) ? launch_g_ks<G32S>(a, K, s) : launch_g_ks<G32>(a, K, s);
    }
    return hipErrorInvalidValue;
}

// where the last pair of the last stage can carry conv_post + tanh: C = 32 pairs at k = 11 (the V1 generator's last ResBlock), conv_post 32 -> 1, k = 7
bool pair_tail_bf16_supported(int C, int K, int post_cin, int post_cout, int post_k) { return C == 32 && K == 11 && post_cin == 32 && post_cout == 1 && post_k == 7; }

// one convolution = [q = tap*KSTEPS + ks][mblk][lane][8] bf16: bf16_pack with (ckc = C, tg = 1, mt = C)
BPackGeom pair_g_pack_geom(int C, int K) { return BPackGeom{C, C, C, K, C, 1}; }

const char* pair_g_kernel_name(int C, int K) {
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "resblock_pair_g_bf16_k<GTile<%d, %d,", C, K);
    return buf;
}

// ---- the fused-pair entry points the engine uses ---------------------------------------------------------------------
// (the first generation — weight slabs double-buffered through LDS, one s_barrier per slab, kernels_bf16_pair.hip in the
// history — lost to this one at every channel count once the tile staging here got its interior fast path)
bool pair_bf16_supported(int C, int K, int dil) {
    return (C == 256 || C == 128 || C == 64 || C == 32) && (K == 3 || K == 7 || K == 11) && dil >= 1 && dil <= 5;
}
BPackGeom pair_pack_geom(int C, int K) { return pair_g_pack_geom(C, K); }
hipError_t launch_pair_bf16(int C, int K, const BConvArgs& a, hipStream_t s) { return launch_pair_g_bf16(C, K, a, s); }
const char* pair_kernel_name(int C, int K) { return pair_g_kernel_name(C, K); }

}  // namespace vtts
