// Whole ResBlock1 in one kernel, bf16:
//     for rate in (d0, d1, d2):  x = c2(lrelu(c1(lrelu(x)))) + x            (vietTTS/hifigan/model.py:44-51)
// followed by the MRF bookkeeping of model.py:115-121 (store / accumulate / accumulate-and-divide, consumer's LeakyReLU).
//
// Why: a fused pair pays a tile staging, two epilogues and an HBM round trip of x per pair.  For the C = 32 stage that is
// HBM-bound outright (3.3 GB per pair for 0.2-0.8 TFLOP; profiles/r01_e_pmc_bf16.md), and for the k = 3 ResBlocks of the
// wider stages staging + epilogues outweigh the six taps (MfmaUtil 0.25-0.46).  Fused per ResBlock the running x never
// leaves the CU between the three pairs: one read of the stage input (+ halo), one read-modify-write of the MRF accumulator.
//
// One workgroup = 4 waves = a window of W time steps x C channels; wave tile 32*MR x 32*NR.  Every intermediate tensor lives
// at the SAME window coordinates (row r <-> time t0 - M + r), convolutions are centred (output row r reads input rows
// r + (j - H)*rate), so
//   * the residual of output row r is always held by the same lane: the running x stays in registers (packed bf16, as the
//     pair-by-pair path rounds it when it stores x' to HBM);
//   * two LDS tiles suffice: A = lrelu(x) (c1's B operand) and T = lrelu(c1(.)) (c2's), each W + 2*GUARD rows with zeroed
//     guard rows, <= 71 KiB together -> two workgroups per CU;
//   * rows whose dependency cone left the window are garbage after each convolution; the margin grows by H*rate + H per
//     pair to M = H*(d0+d1+d2) + 3H per side (12 / 36 / 60 for k = 3 / 7 / 11); only the W - 2M centre rows are stored.
// Zero padding: every convolution of the reference pads ITS input with zeros outside [0, L), so every tile write masks rows
// whose time lies outside the utterance.  Weights: the six convolutions' A fragments are ONE continuous L2 -> register-ring
// stream (as kernels_bf16_rbg.hip: no staging, no synchronisation inside the MFMA loops); two s_barriers per pair.
//
//   C = 32 : W = 512, waves 1 x 4, wave tile 32 x 128      used for k = 3 (k = 7, 11 exist; the pair kernel wins there)
//   C = 64 : W = 256, waves 1 x 4, wave tile 64 x 64       used for k = 3
//   C = 128: W = 128, waves 2 x 2, wave tile 64 x 64       exists for k = 3; the pair kernel wins there
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

#ifndef VTTS_WREG  // register-resident weights (A/B switch)
#define VTTS_WREG 1
#endif
#ifndef VTTS_WREG_RB_MAX  // ... for convolutions of at most this many VGPRs of fragments per lane: 56 = C = 32 at k = 3, 7 (k = 11: 88, spills beside the
#define VTTS_WREG_RB_MAX 64  // running x; C = 64, k = 3: 96)
#endif

namespace vtts {

template <int C_, int KS_, int W_, int WM_, int WN_, int PA_, int MINWG_>
struct RBTile {
    static constexpr int C = C_, KS = KS_, W = W_, WM = WM_, WN = WN_, PA = PA_, MINWG = MINWG_;
    static constexpr int THREADS = 64 * WM * WN, MR = C / WM / 32, NR = W / WN / 32;
    static constexpr int H = (KS - 1) / 2;
    static constexpr int MAXDIL = 5;
    static constexpr int GUARD = H * MAXDIL;            // rows a centred tap can reach beyond the window
    static constexpr int ROWS = W + 2 * GUARD;
    static constexpr int SPR = C / 8, P = C * 2;        // 16-byte slots / bytes per tile row
    static constexpr int KSTEPS = C / 16;               // k-steps per tap
    static constexpr int NQT = KS * KSTEPS;             // k-steps per convolution
    static constexpr int MB = C / 32;
    static constexpr int RA = PA + 1;
    static constexpr int TILE_BYTES = tile_rows16(ROWS) * P;  // rows in multiples of 16 (tile_off's blocks at C = 32 / 64)
    static constexpr int BIAS_BYTES = 6 * C * 4;        // the six convolutions' biases, staged once
    static constexpr int LDS_BYTES = 2 * TILE_BYTES + BIAS_BYTES;
    static constexpr size_t CONV_BYTES = (size_t)KS * C * C * 2;
    static constexpr int XPT = (W * SPR + THREADS - 1) / THREADS;
    static_assert(C % (WM * 32) == 0 && W % (WN * 32) == 0 && LDS_BYTES <= 160 * 1024, "window / LDS");
    static_assert(RA == 4 && (NQT % 4 == 0 || NQT % 4 == 2), "ring of 4; a phase starts at slot 0 or 2");
    // C = 32: a convolution's A operand (NQT * MR fragments, <= 88 VGPRs) is loaded in one burst a phase ahead and the MFMA loops carry
    // no vector-memory instruction (kernels_bf16_rbg.hip: GTile::WREG, same reason)
    static constexpr bool WREG = VTTS_WREG && NQT * MR * 4 <= VTTS_WREG_RB_MAX;
};

template <class T>
__global__ __launch_bounds__(T::THREADS, (T::MINWG * T::THREADS + 255) / 256) void resblock_bf16_k(BConvArgs a) {
    constexpr int C = T::C, KS = T::KS, W = T::W, PA = T::PA, RA = T::RA, MR = T::MR, NR = T::NR, H = T::H, GUARD = T::GUARD;
    constexpr int THREADS = T::THREADS, SPR = T::SPR, P = T::P, KSTEPS = T::KSTEPS, NQT = T::NQT, XPT = T::XPT, MB = T::MB;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* tA = lds;                  // lrelu(x): c1's input
    unsigned char* tT = lds + T::TILE_BYTES;  // lrelu(c1(.)): c2's input
    const float* sbias = reinterpret_cast<const float*>(lds + 2 * T::TILE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int cb0 = wm * (C / T::WM);  // this wave's first output channel
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int Lp = a.L;                                      // rows allocated per utterance
    const int L = a.lens ? min(max(a.lens[b], 0) * a.len_mul, a.L) : a.L;  // valid rows of this utterance, clamped to its slot (ragged batch: the rest reads as zero padding)
    const int d0 = a.dils[0], d1 = a.dils[1], d2 = a.dils[2];
    const int M = H * (d0 + d1 + d2) + 3 * H;  // invalid margin per side after the three pairs
    const int NT = W - 2 * M;                  // outputs per workgroup
    // XCD-aware tile order (as resblock_pair_g_bf16_k, kernels_bf16_rbg.hip): gridDim.x is a multiple of 8 and an XCD (= workgroup index % 8)
    // takes a contiguous, balanced eighth of the utterance's VALID windows (rotated per utterance): the 2 * M margin rows shared with the
    // previous window are L2 hits
    const int ntv = (L + NT - 1) / NT, rx = (int)((blockIdx.x + b) & 7), lox = (rx * ntv) >> 3, hix = ((rx + 1) * ntv) >> 3;
    const bool xmap = (a.L + NT - 1) / NT >= 192;  // launches of short utterance slots keep the launch order (kernels_bf16_rbg.hip: XCD_MAP_MIN_TILES)
    const int tile = xmap ? lox + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (xmap && tile >= hix) return;
    const int t0 = tile * NT;                  // first output time step
    if (t0 >= L) return;                       // a tile past this utterance's end
    const int tw = t0 - M;                     // time of window row 0
    const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + (size_t)b * Lp * C;
    unsigned short* __restrict__ yg = static_cast<unsigned short*>(a.y) + (size_t)b * Lp * C;

    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };
    auto act2 = [](unsigned u) { return lrelu01_pack(bf16_lo(u), bf16_hi(u)); };  // LRELU_SLOPE, model.py:5

    for (int u = tid; u < 6 * C; u += THREADS) reinterpret_cast<float*>(lds + 2 * T::TILE_BYTES)[u] = a.bias[u];
    // ---- guard rows of both tiles = 0 (never written again) ----
    for (int u = tid; u < 2 * 2 * GUARD * SPR; u += THREADS) {
        const int tile = u / (2 * GUARD * SPR), v = u % (2 * GUARD * SPR);
        const int gr = v % (2 * GUARD), c = v / (2 * GUARD);  // consecutive lanes: consecutive rows of one slot (conflict-free in the blocked tiles, tile_off)
        const int row = gr < GUARD ? gr : W + gr;  // [0, GUARD) and [GUARD + W, ROWS)
        *reinterpret_cast<uint4*>((tile ? tT : tA) + tile_off<SPR>(row, c)) = make_uint4(0u, 0u, 0u, 0u);
    }
    // ---- stage A = lrelu(x) over the window (zero outside the utterance) ----
    {
        constexpr int RPI = THREADS / SPR;  // window rows between a thread's consecutive 16-byte units
        static_assert(THREADS % SPR == 0 && RPI % 16 == 0 && (W * SPR) % THREADS == 0, "a thread's units share their column and their swizzle");
        // lane -> (row, slot) as in resblock_pair_g_bf16_k: blocked tiles (SPR = 4, 8) give 8 consecutive lanes 8 consecutive rows of one slot
        constexpr int RW = 64 / SPR;
        const int r0 = (SPR >= 16 || !VTTS_TILE_BLOCKED) ? tid / SPR : wave * RW + lane % RW, c = (SPR >= 16 || !VTTS_TILE_BLOCKED) ? tid % SPR : lane / RW;
        unsigned char* const lds0 = tA + tile_off<SPR>(GUARD + r0, c);  // unit i: + i * RPI * P (RPI is a multiple of 16)
        uint4 v[XPT];
        if (tw >= 0 && tw + W <= L) {  // interior window: no clamping, no masking, constant strides
            const unsigned short* __restrict__ g0 = xg + (size_t)(tw + r0) * C + c * 8;
#pragma unroll
            for (int i = 0; i < XPT; ++i) v[i] = *reinterpret_cast<const uint4*>(g0 + (size_t)i * RPI * C);
        } else {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int t = tw + r0 + i * RPI;
                const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
                v[i] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + c * 8);
                if (t < 0 || t >= L) v[i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            v[i].x = act2(v[i].x);
            v[i].y = act2(v[i].y);
            v[i].z = act2(v[i].z);
            v[i].w = act2(v[i].w);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) *reinterpret_cast<uint4*>(lds0 + i * RPI * P) = v[i];
    }
    // ---- the running x of this lane's outputs, in the accumulator layout, packed bf16: xr[mr][nr][p] = 8 values r0 = 8p ----
    // (column n = wn*(W/WN) + nr*32 + l31, channels cb0 + 32*mr + 8*rq + 4*lh + i for r = 4*rq + i)
    const int col0 = wn * (W / T::WN) + l31;
    uint4 xr[MR][NR][2];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int t = tw + col0 + nr * 32;
                const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
                uint4 r = *reinterpret_cast<const uint4*>(xg + (size_t)tc * C + cb0 + 32 * mr + 16 * p + 8 * lh);
                if (t < 0 || t >= L) r = make_uint4(0u, 0u, 0u, 0u);
                swap_pair(r.x, r.z);  // 8 consecutive channels per lane -> the accumulator layout's 4 + 4
                swap_pair(r.y, r.w);
                xr[mr][nr][p] = r;
            }

    f32x16 acc[MR][NR];
    auto init_acc = [&](int conv) {  // accumulators start from the bias (LDS copy)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 bv = *reinterpret_cast<const float4*>(sbias + conv * C + cb0 + 32 * mr + 8 * rq + 4 * lh);
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    acc[mr][nr][4 * rq + 0] = bv.x;
                    acc[mr][nr][4 * rq + 1] = bv.y;
                    acc[mr][nr][4 * rq + 2] = bv.z;
                    acc[mr][nr][4 * rq + 3] = bv.w;
                }
            }
    };
    // acc += W (*) tile, centred taps: output column n reads tile row GUARD + n + (tap - H)*dl.
    // The six convolutions' weights are contiguous, so the A-fragment stream is ONE sequence of 6*NQT k-steps: the
    // register ring runs on across the epilogues (its look-ahead loads at the end of a phase are the next phase's first
    // fragments, in flight while the epilogue runs).  Ring slot of global step g = g mod 4: a c1 phase starts at slot 0, a
    // c2 phase at slot NQT mod 4 (0 or 2).
    const uint4* __restrict__ aptr = reinterpret_cast<const uint4*>(a.wp) + (size_t)(wm * MR) * 64 + lane;
    bf16x8 af[RA][MR];
    auto load_a = [&](int g, int slot) {
        const int gc = g < 6 * NQT ? g : 6 * NQT - 1;  // the very last look-aheads re-read the last step (stays inside the blob)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) af[slot][mr] = __builtin_bit_cast(bf16x8, aptr[(size_t)(gc * MB + mr) * 64]);
    };
    auto conv_phase = [&](int conv, auto s0_tag, const unsigned char* __restrict__ tile, int dl) {
        constexpr int S0 = decltype(s0_tag)::value;  // ring slot of this phase's step 0
        const int g0 = conv * NQT;
        bf16x8 bf[2][NR];
        auto load_b = [&](int q, int par) {
            const int tap = q / KSTEPS, ks = q % KSTEPS;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int row = GUARD + col0 + nr * 32 + (tap - H) * dl;
                bf[par][nr] = *reinterpret_cast<const bf16x8*>(tile + tile_off<SPR>(row, ks * 2 + lh));
            }
        };
        auto step = [&](int q, int i, bool has_b) {  // i = q mod 4: ring slot and B parity are compile-time
            load_a(g0 + q + PA, (S0 + i + PA) % RA);
            if (has_b) load_b(q + 1, (i + 1) & 1);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(S0 + i) % RA][mr], bf[i & 1][nr], acc[mr][nr], 0, 0, 0);
            // keep hipcc's scheduler from sinking the look-ahead loads to their uses, spread between the MFMAs
            // (as in kernels_bf16_rbg.hip)
            constexpr int NM = MR * NR;
            const int mem = MR + (has_b ? NR : 0);
            int done = 0;
            for (int k = 0; k < NM; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                const int upto = (k + 1) * mem / NM;
                for (; done < upto; ++done) {
                    if (done < MR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        };
        constexpr int TAILQ = NQT % 4;  // 0 or 2 steps peeled after the rolled blocks of 4
        load_b(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma nounroll
        for (int q0 = 0; q0 < NQT - 4 - TAILQ; q0 += 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) step(q0 + i, i, true);
        }
        if constexpr (TAILQ == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) step(NQT - 4 + i, i, i < 3);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) step(NQT - 6 + i, i, true);
            step(NQT - 2, 0, true);
            step(NQT - 1, 1, false);
        }
    };
    // T::WREG: convolution `conv`'s fragments in registers; requested with load_w_all(conv) a phase ahead
    constexpr int NQW = T::WREG ? NQT : 1;
    bf16x8 aw[NQW][MR];
    auto load_w_all = [&](int conv) {
        if constexpr (T::WREG) {
            if (conv < 6) {
#pragma unroll
                for (int q = 0; q < NQT; ++q)
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) aw[q][mr] = __builtin_bit_cast(bf16x8, aptr[(size_t)((conv * NQT + q) * MB + mr) * 64]);
            }
        }
    };
    // (the accumulators of a WREG pass start from the bias block as the first MFMA's C operand: 16 registers read from the LDS copy per
    //  m-block instead of 16 * NR v_mov per convolution and wave)
    auto conv_phase_wreg = [&](const unsigned char* __restrict__ tile, int dl, int conv) {
        f32x16 bblk[MR];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 bv = *reinterpret_cast<const float4*>(sbias + conv * C + cb0 + 32 * mr + 8 * rq + 4 * lh);
                bblk[mr][4 * rq + 0] = bv.x;
                bblk[mr][4 * rq + 1] = bv.y;
                bblk[mr][4 * rq + 2] = bv.z;
                bblk[mr][4 * rq + 3] = bv.w;
            }
        bf16x8 bf[2][NR];
        auto load_b = [&](int q, int par) {
            const int tap = q / KSTEPS, ks = q % KSTEPS;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) bf[par][nr] = *reinterpret_cast<const bf16x8*>(tile + tile_off<SPR>(GUARD + col0 + nr * 32 + (tap - H) * dl, ks * 2 + lh));
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
            for (int i = 0; i < MR * NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q + 1 < NQT && i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    if constexpr (T::WREG) {
        load_w_all(0);
    } else {
#pragma unroll
        for (int q = 0; q < PA; ++q) load_a(q, q % RA);  // the stream's first fragments, under the tile staging
    }
    // this lane's 16 values of block nr -> tile rows (8 consecutive channels per lane after the swap), masked outside [0, L)
    const bool interior = tw >= 0 && tw + W <= L;  // every row of the window lies inside the utterance: no zero-padding masks (wave-uniform)
    auto write_tile = [&](unsigned char* tile, int mr, int nr, unsigned p0, unsigned p1, unsigned q0, unsigned q1, int p) {
        const int row = GUARD + col0 + nr * 32;
        const int t = tw + col0 + nr * 32;
        if (!interior && (t < 0 || t >= L)) p0 = p1 = q0 = q1 = 0u;
        swap_pair(p0, q0);
        swap_pair(p1, q1);
        const int slot = ((cb0 + 32 * mr) >> 3) + 2 * p + lh;
        *reinterpret_cast<uint4*>(tile + tile_off<SPR>(row, slot)) = make_uint4(p0, p1, q0, q1);
    };

    __syncthreads();  // tile A staged, biases in LDS
    if constexpr (!T::WREG) init_acc(0);

    constexpr int S2 = NQT % 4;  // ring slot at which a c2 phase starts
#pragma nounroll
    for (int pr = 0; pr < 3; ++pr) {
        const int dl = pr == 0 ? d0 : (pr == 1 ? d1 : d2);
        // ---- c1 over A ----
        if constexpr (T::WREG) {
            conv_phase_wreg(tA, dl, 2 * pr);
            load_w_all(2 * pr + 1);  // c2's fragments, in flight under the epilogue
        } else {
            conv_phase(2 * pr, std::integral_constant<int, 0>{}, tA, dl);
        }
        // ---- xt = lrelu(.) -> T ----
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int r0 = 8 * p;
                    const f32x16& c = acc[mr][nr];
                    write_tile(tT, mr, nr, lrelu01_pack(c[r0 + 0], c[r0 + 1]), lrelu01_pack(c[r0 + 2], c[r0 + 3]),
                               lrelu01_pack(c[r0 + 4], c[r0 + 5]), lrelu01_pack(c[r0 + 6], c[r0 + 7]), p);
                }
        if constexpr (!T::WREG) init_acc(2 * pr + 1);
        __syncthreads();  // T written; every wave is done reading A
        // ---- c2 over T (rate 1) ----
        if constexpr (T::WREG) {
            conv_phase_wreg(tT, 1, 2 * pr + 1);
            load_w_all(2 * pr + 2);  // the next pair's c1 (nothing after the last pair)
        } else {
            conv_phase(2 * pr + 1, std::integral_constant<int, S2>{}, tT, 1);
        }
        // ---- x = c2 + x  (model.py:50) ----
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int r0 = 8 * p;
                    const uint4 r = xr[mr][nr][p];
                    f32x16& c = acc[mr][nr];
                    c[r0 + 0] = vadd_raw(bf16_lo(r.x), c[r0 + 0]); c[r0 + 1] = vadd_raw(bf16_hi(r.x), c[r0 + 1]);
                    c[r0 + 2] = vadd_raw(bf16_lo(r.y), c[r0 + 2]); c[r0 + 3] = vadd_raw(bf16_hi(r.y), c[r0 + 3]);
                    c[r0 + 4] = vadd_raw(bf16_lo(r.z), c[r0 + 4]); c[r0 + 5] = vadd_raw(bf16_hi(r.z), c[r0 + 5]);
                    c[r0 + 6] = vadd_raw(bf16_lo(r.w), c[r0 + 6]); c[r0 + 7] = vadd_raw(bf16_hi(r.w), c[r0 + 7]);
                }
        if (pr < 2) {
            // the pair-by-pair path stores x' as bf16 and the next pair reloads it: round here the same way, keep it as
            // the next residual, and write lrelu(x') as the next c1's input
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int r0 = 8 * p;
                        const f32x16& c = acc[mr][nr];
                        uint4 r;
                        r.x = pack_bf16x2(c[r0 + 0], c[r0 + 1]);
                        r.y = pack_bf16x2(c[r0 + 2], c[r0 + 3]);
                        r.z = pack_bf16x2(c[r0 + 4], c[r0 + 5]);
                        r.w = pack_bf16x2(c[r0 + 6], c[r0 + 7]);
                        xr[mr][nr][p] = r;
                        write_tile(tA, mr, nr, act2(r.x), act2(r.y), act2(r.z), act2(r.w), p);  // lrelu of the ROUNDED x', as the pair path's staging does
                    }
            if constexpr (!T::WREG) init_acc(2 * pr + 2);
            __syncthreads();  // A written; every wave is done reading T
        }
    }

    // ---------------- MRF bookkeeping + store of the W - 2M centre rows ----------------
    {
        const float s_out = a.slope_out;
        const float dv = a.div;
        const float rdv = mrf_recip(dv);
        const bool acc_add = a.acc_add != 0;
        // the residual registers are dead now: the MRF accumulator rows take their place (all requests first, one wait)
        if (acc_add) {  // xs += rb(x)  (model.py:118-120)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int t = tw + col0 + nr * 32;
                        const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
                        xr[mr][nr][p] = *reinterpret_cast<const uint4*>(yg + (size_t)tc * C + cb0 + 32 * mr + 16 * p + 8 * lh);
                    }
        }
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int n = col0 + nr * 32;
                    const int t = tw + n;
                    const bool ok = n >= M && n < W - M && t >= 0 && t < L;
                    const int r0 = 8 * p;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[mr][nr][r0 + e];
                    if (acc_add) {
                        uint4 o = xr[mr][nr][p];
                        swap_pair(o.x, o.z);
                        swap_pair(o.y, o.w);
                        v[0] = vadd_raw(bf16_lo(o.x), v[0]); v[1] = vadd_raw(bf16_hi(o.x), v[1]); v[2] = vadd_raw(bf16_lo(o.y), v[2]); v[3] = vadd_raw(bf16_hi(o.y), v[3]);
                        v[4] = vadd_raw(bf16_lo(o.z), v[4]); v[5] = vadd_raw(bf16_hi(o.z), v[5]); v[6] = vadd_raw(bf16_lo(o.w), v[6]); v[7] = vadd_raw(bf16_hi(o.w), v[7]);
                    }
                    if (dv != 1.0f) {  // x = xs / num_kernels  (model.py:121)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = VTTS_MRF_DIV ? v[e] / dv : v[e] * rdv;
                    }
                    if (s_out != 1.0f) {  // the consumer's LeakyReLU (model.py:112 / :122), applied once by the producer
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = lrelu_f(v[e], s_out);
                    }
                    unsigned p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
                    unsigned q0 = pack_bf16x2(v[4], v[5]), q1 = pack_bf16x2(v[6], v[7]);
                    swap_pair(p0, q0);
                    swap_pair(p1, q1);
                    if (ok) *reinterpret_cast<uint4*>(yg + (size_t)t * C + cb0 + 32 * mr + 16 * p + 8 * lh) = make_uint4(p0, p1, q0, q1);
                }
    }
}

// window geometry (kernel-development switches, tools/ab_bench.sh): time steps per window, waves along time, workgroups per CU
#ifndef VTTS_RB32_W
#define VTTS_RB32_W 512
#define VTTS_RB32_WN 4
#define VTTS_RB32_WG 2
#endif
#ifndef VTTS_RB64_W
#define VTTS_RB64_W 256
#define VTTS_RB64_WN 4
#define VTTS_RB64_WG 2
#endif
//                                     C   KS   W  WM WN PA MINWG
#ifndef VTTS_RB32K3_W  // k = 3 at C = 32: 256-step windows, three workgroups per CU (1.17 -> 1.03 ms per launch: its phases are 768 cycles of MFMA each, more
#define VTTS_RB32K3_W 256   // waves in flight hide their start-up latencies; k = 7 loses 15 % that way, gpurun_out/r03_exp24)
#define VTTS_RB32K3_WG 3
#endif
template <int KS> using RB32 = RBTile<32, KS, KS == 3 ? VTTS_RB32K3_W : VTTS_RB32_W, 1, VTTS_RB32_WN, 3, KS == 3 ? VTTS_RB32K3_WG : VTTS_RB32_WG>;
template <int KS> using RB64 = RBTile<64, KS, VTTS_RB64_W, 1, VTTS_RB64_WN, 3, VTTS_RB64_WG>;
#ifndef VTTS_RB128_W  // A/B (round 4, fuse = 3): a 256-step window on eight waves (one workgroup per CU, 9 % margin) 3.08 ms, the 128-step one (two per CU, 19 %) 3.12, three pair launches 2.88
#define VTTS_RB128_W 128
#define VTTS_RB128_WN 2
#define VTTS_RB128_WG 2
#endif
template <int KS> using RB128 = RBTile<128, KS, VTTS_RB128_W, 2, VTTS_RB128_WN, 3, VTTS_RB128_WG>;
constexpr bool RB64_ALL_K = RB64<11>::LDS_BYTES * VTTS_RB64_WG <= 160 * 1024 && VTTS_RB64_W - 2 * 60 >= 128;  // k = 7, 11 too once the window is wide enough

template <class T>
static hipError_t launch_rb(const BConvArgs& a, hipStream_t s) {
    static DynLdsOnce once;  // per device (vtts_internal.h)
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&resblock_bf16_k<T>), T::LDS_BYTES, once); e != hipSuccess) return e;
    int dsum = 0;
    for (int i = 0; i < 3; ++i) {
        if (a.dils[i] < 1 || a.dils[i] > T::MAXDIL) return hipErrorInvalidValue;
        dsum += a.dils[i];
    }
    const int NT = T::W - 2 * (T::H * dsum + 3 * T::H);
    if (NT < 32) return hipErrorInvalidValue;
    dim3 grid((a.L + NT - 1) / NT, 1, a.B);
    if ((int)grid.x >= 192) grid.x = (grid.x + 7) / 8 * 8;  // whole rounds of the 8 XCDs (XCD-aware order); a window past the utterance exits at once
    hipLaunchKernelGGL(resblock_bf16_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

// where a whole-ResBlock kernel exists
bool resblock_bf16_supported(int C, int K, const int* dils) {
    for (int i = 0; i < 3; ++i)
        if (dils[i] < 1 || dils[i] > 5) return false;
    if (C == 32) return K == 3 || K == 7 || K == 11;
    if (C == 64 && RB64_ALL_K) return K == 3 || K == 7 || K == 11;
    return (C == 64 || C == 128) && K == 3;
}
// ... and where it is the faster choice (per ResBlock at B = 64 x T = 1024, rocprofv3; round 1: profiles/r01_h_fuse_policy.md, round 3 after
// the blocked LDS tiles and the un-packed VALU: gpurun_out/r03_exp6/fuse3_stats.md vs profiles/r03_b_kernel_stats.md):
//   C = 32 : k = 3  1.17 ms vs 1.62 ms as three pair launches;  k = 7  2.07 vs 2.21 (round 1: 2.27 vs 2.22);  k = 11  3.20 vs 2.90 (23 % of the window is margin)
//   C = 64 : k = 3  1.77 vs 2.29
//   C = 128: k = 3  3.27 vs 3.12 (64 x 64 wave tiles double the weight-fragment traffic per MFMA; 19 % margin)
bool resblock_bf16_preferred(int C, int K) { return (C == 32 && (K == 3 || K == 7)) || (C == 64 && K == 3); }

// a.wp = [pair0 c1][pair0 c2][pair1 c1]...[pair2 c2], each one convolution in pair_g_pack_geom(C, K) order;
// a.bias = 6 x [C]; a.dils = the three rates; a.x = stage input (raw), a.y = MRF accumulator / stage output
hipError_t launch_resblock_bf16(int C, int K, const BConvArgs& a, hipStream_t s) {
    if (C == 32) switch (K) {
            case 3: return launch_rb<RB32<3>>(a, s);
            case 7: return launch_rb<RB32<7>>(a, s);
            case 11: return launch_rb<RB32<11>>(a, s);
        }
    if (C == 64 && K == 3) return launch_rb<RB64<3>>(a, s);
    if constexpr (RB64_ALL_K) {
        if (C == 64 && K == 7) return launch_rb<RB64<7>>(a, s);
        if (C == 64 && K == 11) return launch_rb<RB64<11>>(a, s);
    }
    if (C == 128 && K == 3) return launch_rb<RB128<3>>(a, s);
    return hipErrorInvalidValue;
}

const char* resblock_kernel_name(int C, int K) {
    static thread_local char buf[64];
    snprintf(buf, sizeof(buf), "resblock_bf16_k<RBTile<%d, %d,", C, K);
    return buf;
}

}  // namespace vtts
