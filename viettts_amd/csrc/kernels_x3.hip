// Fused ResBlock1 pair on the bf16 matrix pipe with SPLIT operands ("bf16x3"):   x' = c2(lrelu(c1(lrelu(x)))) + x
// (vietTTS/hifigan/model.py:45-50) at fp32-grade accuracy.
//
// Why (round 4, profiles/r04_b_split_findings.md): gfx950's bf16 MFMA rate is 16x its fp32 MFMA rate, and BASELINE.json's 1e-4 is only met by
// the fp32 engine.  Write every operand as the sum of two bf16 terms, v = v0 + v1 with v0 = bf16(v), v1 = bf16(v - v0) (16 significand bits),
// and form a product from three bf16 x bf16 terms accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//     x w  ~=  x1 w0 + x0 w1 + x0 w0                  (the dropped x1 w1 is 2^-18 of the product, as is the two-term split's own residual)
// Whole generator, CPU emulation against the fp64 oracle: max-abs 1.6-1.9e-5 on the waveform (bf16 operands: 1.0e-2; fp32: 9e-7) — 5x inside
// the 1e-4 bar — for 3/16 of the fp32 MFMA time.
//
// This kernel is a drop-in for the fp32 engine's pair launches (kernels_f32_pair.hip): activations stay fp32, CHANNEL-MAJOR [B][C][L] in HBM (so
// conv_pre, the transposed convolutions, conv_post and everything element-wise remain the fp32 engine's exact kernels), only the two
// convolutions' products take the split route.  One workgroup = 8 waves = C output channels x N1 columns:
//   * staging: lane <-> time step (a wave reads 64 consecutive floats of one channel: coalesced), 8 channels per thread and unit;
//     LeakyReLU in fp32, split, two 16-byte ds_write_b128 into a HI and a LO tile — channels-last bf16 rows with the bf16 engine's
//     conflict-free layouts (bf16_common.h: tile_off), i.e. the transposition costs no extra pass;
//   * MFMA loops: wave tile 64 x 64 (32 x 64 at C = 32); per k-step 2 x 2 A fragments (weights hi / lo, host-packed in A-fragment order,
//     straight from L2 through a register ring) and 2 x 2 B fragments (tile hi / lo, one k-step ahead) feed 12 MFMAs — 2/3 of the bf16
//     engine's operand traffic per MFMA; small terms first;
//   * epilogue 1: + b1, LeakyReLU, the reference's zero padding of xt, split, into the xt tiles (hi / lo) over the dead X tiles;
//   * epilogue 2: + b2, + x (fp32, from HBM / L2), MRF accumulate / mean as device_common.h: epilogue_store, fp32 stores — in the MFMA
//     accumulator layout a half-wave holds 32 consecutive time steps of one channel: coalesced 128-byte rows, no lane exchange.
// LDS: two tiles of up to 78 KB -> one workgroup of 8 waves per CU (two waves per SIMD cover each other's fragment latencies).
#include <string.h>

#include <type_traits>

#include "bf16_common.h"
#include "device_common.h"

namespace vtts {

constexpr int X3_XCD_MIN_TILES = 64;

struct PairArgsX3 {
    ConvArgs a;          // x, bias (b1), dil, slope_in, B, L, zrev; output side: y, res (= x), acc_mode, div
    const void* w1;      // c1: [hi fragments][lo fragments], each pair_g_pack_geom(C, K) order
    const void* w2;      // c2 likewise
    const float* bias2;
};

template <int C_, int KS_, int N1_, int WM_, int WN_, int XC_ = C_, int WPS_ = 2>
struct XTile {
    static constexpr int C = C_, KS = KS_, N1 = N1_, WM = WM_, WN = WN_, XC = XC_, NXC = C / XC;
    static constexpr int WPS = WPS_;                    // waves per SIMD the registers are budgeted for (workgroups per CU x waves / 4)
    static constexpr int THREADS = 64 * WM * WN, NWAVES = WM * WN;
    static constexpr int MR = C / WM / 32, NR = N1 / WN / 32;
    static constexpr int H2 = (KS - 1) / 2, MAXDIL = 5, NT2 = N1 - 2 * H2;
    static constexpr int SPR1 = XC / 8, P1 = XC * 2;    // X tile (one channel chunk): 16-byte slots / bytes per row
    static constexpr int SPR2 = C / 8, P2 = C * 2;      // xt tile (all channels)
    static constexpr int KSTEPS = C / 16, KSX = XC / 16, MB = C / 32;
    static constexpr int PA = KSX >= 4 ? 3 : 1, RA = PA + 1;  // A-fragment ring: k-steps ahead / slots
    static constexpr int ROWST = N1 + 2 * H2;
    static constexpr size_t CONV_BYTES = (size_t)KS * C * C * 2;  // one plane (hi or lo) of one convolution
    static __host__ __device__ constexpr int plane_bytes(int dil) {
        const int bx = tile_rows16(N1 + 2 * H2 * dil) * P1, bt = tile_rows16(ROWST) * P2;
        return bx > bt ? bx : bt;
    }
    static __host__ __device__ constexpr int lds_bytes(int dil) { return 2 * plane_bytes(dil); }
    static_assert(C % (WM * 32) == 0 && N1 % (WN * 32) == 0 && C % XC == 0, "tiling");
    static_assert(KSX % RA == 0 && KSTEPS % RA == 0, "ring slots are compile-time positions in a tap's k-steps");
    static_assert(2 * plane_bytes(MAXDIL) <= 160 * 1024, "LDS budget");
};

template <class T>
__global__ __launch_bounds__(T::THREADS, T::WPS) void resblock_pair_x3_k(PairArgsX3 p) {
    constexpr int C = T::C, KS = T::KS, N1 = T::N1, WN = T::WN, H2 = T::H2, NT2 = T::NT2, MR = T::MR, NR = T::NR;
    constexpr int SPR1 = T::SPR1, SPR2 = T::SPR2, XC = T::XC, NXC = T::NXC, KSX = T::KSX, KSTEPS = T::KSTEPS, MB = T::MB;
    constexpr int PA = T::PA, RA = T::RA, THREADS = T::THREADS, NWAVES = T::NWAVES;
    const ConvArgs& a = p.a;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int dil = a.dil;
    unsigned char* const thi = lds;
    unsigned char* const tlo = lds + T::plane_bytes(dil);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, lh = lane >> 5;
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int LP = a.L;              // row pitch of x / y
    const int L = valid_len(a, b);   // this utterance's columns (ragged batches; == LP otherwise): zero padding and store masks follow it
    int tile = blockIdx.x;
    if (gridDim.x >= X3_XCD_MIN_TILES) {  // XCD-aware tile order (as the other pair kernels): XCD blockIdx.x % 8 takes a contiguous eighth of the tiles
        const int nt = (L + NT2 - 1) / NT2, r = (int)((blockIdx.x + blockIdx.z) & 7), lo = (r * nt) >> 3, hi = ((r + 1) * nt) >> 3;
        tile = lo + (int)(blockIdx.x >> 3);
        if (tile >= hi) return;
    }
    const int t0 = tile * NT2;
    if (t0 >= L) return;
    const int h1 = H2 * dil;
    const int rowsx = N1 + 2 * h1;           // X rows: times t0 - H2 - h1 ...
    const float* __restrict__ xb = a.x + (long)b * C * LP;
    const int m0 = wm * (C / T::WM);         // first output channel of this wave
    const float slope = a.slope_in;

    // two bf16 terms of two fp32 values: (hi pair, lo pair), round-to-nearest-even; v - float(hi) is exact in fp32
    auto split2 = [](float v0, float v1, unsigned& hi, unsigned& lo) {
        hi = pack_bf16x2(v0, v1);
        lo = pack_bf16x2(v0 - bf16_lo(hi), v1 - bf16_hi(hi));
    };

    // ---------------- X tile (channel chunk xc): lane <-> time step, 8 channels per unit; LeakyReLU, split, swizzled ds_write_b128 x 2 ----------------
    auto stage_x = [&](int xc) {
        const int tx0 = t0 - H2 - h1;
        const int nblk = (rowsx + 63) >> 6;
        const int units = nblk * SPR1;  // (64-row block, 16-byte slot)
        const float* __restrict__ xc0 = xb + (long)(xc * XC) * LP;
        constexpr int UB = 4;           // units in flight per thread: 32 dword loads
        for (int u0 = wave * UB; u0 < units; u0 += NWAVES * UB) {
            float v[UB][8];
            int row[UB], slot[UB];
            bool live[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int u = u0 + q < units ? u0 + q : units - 1;
                live[q] = u0 + q < units;
                slot[q] = u % SPR1;
                row[q] = (u / SPR1) * 64 + lane;
                const int t = tx0 + row[q];
                const bool ok = live[q] && row[q] < rowsx && t >= 0 && t < L;
                const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
                const float* __restrict__ g = xc0 + (long)(slot[q] * 8) * LP + tc;  // unconditional loads from clamped addresses, masked afterwards
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q][e] = g[(long)e * LP];
                if (!ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q][e] = 0.0f;
                }
            }
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                if (!live[q] || row[q] >= rowsx) continue;
                uint4 h4, l4;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q][e] = lrelu(v[q][e], slope);
                split2(v[q][0], v[q][1], h4.x, l4.x);
                split2(v[q][2], v[q][3], h4.y, l4.y);
                split2(v[q][4], v[q][5], h4.z, l4.z);
                split2(v[q][6], v[q][7], h4.w, l4.w);
                const int off = tile_off<SPR1>(row[q], slot[q]);
                *reinterpret_cast<uint4*>(thi + off) = h4;
                *reinterpret_cast<uint4*>(tlo + off) = l4;
            }
        }
    };

    f32x16 acc[MR][NR];
    // the accumulators start from the bias (round 5: the same values resblock_x3_k's first MFMA takes as its C operand — kernels_x3_rb.hip — so the two
    // kernels stay bit-identical; the epilogues add no bias)
    auto init_acc = [&](const float* __restrict__ bias) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = bias[m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) acc[mr][nr][r] = bv;
            }
    };

    // ---- one convolution pass over the LDS tiles: acc += W[:, chunk] (*) tile, three bf16 products per operand pair ----
    // A fragment (plane, tap, ks, mr): 16 bytes per lane at  w + plane*CONV_BYTES + (((tap*KSTEPS + ks0 + ks)*MB + wm*MR + mr)*64 + lane)*16
    // B fragment (plane, tap, ks, nr): tile row  n + tap*dl  (n = this lane's output column), 16-byte slot 2*ks + lh of that row
    const int rowbase0 = wn * (N1 / WN) + l31;
    auto conv_phase = [&](const unsigned char* __restrict__ w, int dl, auto sprb_tag, auto nks_tag, int ks0) {
        constexpr int SPRB = decltype(sprb_tag)::value, NKS = decltype(nks_tag)::value;
        constexpr int NSTEPS = KS * NKS;
        // weight fragments by buffer loads (the bf16 pair kernel's lean addressing): the lane's offset in a VGPR, the k-step's in an SGPR, the
        // m-block an immediate — no per-load 64-bit address arithmetic on the issue port the MFMAs share
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(w), 0, (int)(2 * T::CONV_BYTES), 0x00020000);
        const unsigned a_voff = (unsigned)((wm * MR) * 64 + lane) * 16;
        bf16x8 af[RA][MR][2], bf[2][NR][2];
        auto load_a = [&](int s, int slot) {  // flat step s = tap*NKS + ks of this pass (past the end: re-read the last step, never used)
            const int sc = s < NSTEPS ? s : NSTEPS - 1;
            const int tap = sc / NKS, ks = sc - tap * NKS;
            const int soff = ((tap * KSTEPS + ks0 + ks) * MB) * 1024;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                af[slot][mr][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + mr * 1024, soff, 0));
                af[slot][mr][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + mr * 1024, soff + (int)T::CONV_BYTES, 0));
            }
        };
        auto load_b = [&](int s, int par) {
            const int sc = s < NSTEPS ? s : NSTEPS - 1;
            const int tap = sc / NKS, ks = sc - tap * NKS;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int off = tile_off<SPRB>(rowbase0 + tap * dl + nr * 32, ks * 2 + lh);
                bf[par][nr][0] = *reinterpret_cast<const bf16x8*>(thi + off);
                bf[par][nr][1] = *reinterpret_cast<const bf16x8*>(tlo + off);
            }
        };
#pragma unroll
        for (int s = 0; s < PA; ++s) load_a(s, s % RA);
        load_b(0, 0);
        static_assert(NKS % RA == 0 && NKS % 2 == 0, "ring slot / B parity are compile-time positions in a tap");
#pragma unroll 1
        for (int s0 = 0; s0 < NSTEPS; s0 += NKS) {  // one tap per iteration
#pragma unroll
            for (int i = 0; i < NKS; ++i) {
                load_a(s0 + i + PA, (i + PA) % RA);
                load_b(s0 + i + 1, (i + 1) & 1);
                const int sl = i % RA, par = i & 1;
                // small terms first; MR * NR independent accumulators between two MFMAs on the same one
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sl][mr][1], bf[par][nr][0], acc[mr][nr], 0, 0, 0);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sl][mr][0], bf[par][nr][1], acc[mr][nr], 0, 0, 0);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sl][mr][0], bf[par][nr][0], acc[mr][nr], 0, 0, 0);
                // keep hipcc from sinking the look-ahead loads to their uses (it does, to save registers: the first build waited for every
                // fragment right in front of the MFMA that consumes it) and spread them between this step's MFMAs (the bf16 pair kernel's
                // pin_step): one MFMA, then the next pending A load (VMEM) or B read (DS)
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
        }
    };

    // ---------------- phase 1: xt = c1(lrelu(x)); column n <-> time t0 - H2 + n; tap j reads X row n + j*dil ----------------
    init_acc(a.bias);
#pragma unroll
    for (int xc = 0; xc < NXC; ++xc) {
        if (xc > 0) __syncthreads();  // every wave is done reading the previous channel chunk
        stage_x(xc);
        __syncthreads();
        conv_phase(static_cast<const unsigned char*>(p.w1), dil, std::integral_constant<int, SPR1>{}, std::integral_constant<int, KSX>{}, xc * KSX);
    }
    __syncthreads();  // every wave is done reading the X tiles

    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };

    // ---------------- epilogue 1: + b1, LeakyReLU, zero outside [0, L), split -> xt tiles ----------------
    // A lane's accumulators of one 32 x 32 block: column (time) l31, rows (channels) 8*rq + 4*lh + i, r = 4*rq + i.  After the exchange across
    // the wave halves lh = 0 owns channels 16p .. 16p+7 and lh = 1 owns 16p+8 .. 16p+15 of the 16-channel half p (as the bf16 pair kernel).
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int cb = m0 + mr * 32 + 16 * pp;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int row = wn * (N1 / WN) + nr * 32 + l31;
                const int tt = t0 - H2 + row;
                const bool ok = tt >= 0 && tt < L;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = lrelu(acc[mr][nr][8 * pp + e], slope);  // (b1 is in the sum already)
                    if (!ok) v[e] = 0.0f;  // c2's own zero padding applies to xt
                }
                unsigned hp0, hp1, hq0, hq1, lp0, lp1, lq0, lq1;
                split2(v[0], v[1], hp0, lp0);
                split2(v[2], v[3], hp1, lp1);
                split2(v[4], v[5], hq0, lq0);
                split2(v[6], v[7], hq1, lq1);
                swap_pair(hp0, hq0);
                swap_pair(hp1, hq1);
                swap_pair(lp0, lq0);
                swap_pair(lp1, lq1);
                const int off = tile_off<SPR2>(row, (cb >> 3) + lh);
                *reinterpret_cast<uint4*>(thi + off) = make_uint4(hp0, hp1, hq0, hq1);
                *reinterpret_cast<uint4*>(tlo + off) = make_uint4(lp0, lp1, lq0, lq1);
            }
        }
    for (int u = tid; u < 2 * H2 * SPR2; u += THREADS) {  // rows N1 .. N1 + 2*H2 - 1 are only read by the discarded output columns: keep them finite
        const int off = tile_off<SPR2>(N1 + u % (2 * H2), u / (2 * H2));
        *reinterpret_cast<uint4*>(thi + off) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(tlo + off) = make_uint4(0u, 0u, 0u, 0u);
    }
    init_acc(p.bias2);
    __syncthreads();  // xt tiles written

    // ---------------- phase 2: c2 over the xt tiles (rate 1): column n <-> time t0 + n, tap j reads xt row n + j ----------------
    conv_phase(static_cast<const unsigned char*>(p.w2), 1, std::integral_constant<int, SPR2>{}, std::integral_constant<int, KSTEPS>{}, 0);

    // ---------------- epilogue 2: + b2, + x, MRF accumulate / mean (the operations of device_common.h: epilogue_store, in its order) ----------------
    // Every residual value of an m-block is requested before the first one is used (the ring and fragment registers are dead here): MR
    // exposed round trips per tile instead of eight — it matters in a one-workgroup-per-CU kernel, nobody else's MFMAs cover it.  (Requested ahead
    // of the c2 loop they would arrive for free, but 64 more live registers spill: 94 VGPRs, measured by the compiler.)
    const int mode = a.acc_mode;
    const float dv = a.div;
    const float* __restrict__ resb = a.res + (long)b * C * LP;
    float* yb = a.y + (long)b * C * LP;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        float rres[NR][16];  // one m-block's residual values per round trip (all MR * NR * 16 at once spill)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int n = wn * (N1 / WN) + nr * 32 + l31;
            const int t = t0 + n;
            const int tc = (n < NT2 && t < L) ? t : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) rres[nr][r] = resb[(m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + tc];
        }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int n = wn * (N1 / WN) + nr * 32 + l31;
            const int t = t0 + n;
            const bool ok = n < NT2 && t < L;
            const int tc = ok ? t : 0;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float yv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = r0 + q;
                    yv[q] = mode != ACC_STORE ? yb[(m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + tc] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = r0 + q;
                    const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[mr][nr][r];  // (b2 is in the sum already)
                    v = v + rres[nr][r];
                    if (mode == ACC_ADD) v = yv[q] + v;
                    else if (mode == ACC_MEAN) v = (yv[q] + v) / dv;
                    if (ok) yb[co * LP + tc] = v;
                }
            }
        }
    }
}

// ---- tile table ------------------------------------------------------------------------------------
// Two geometries were measured per class (64 x 1024 frames, rocprofv3 per launch, gpurun_out/r04_run11): ONE 8-wave workgroup per CU on full-width
// tiles (two waves per SIMD cover each other's fragment latencies; every phase is workgroup-wide) against TWO 4-wave workgroups per CU on
// half-width tiles with the X tile staged 64 channels at a time (one's staging / epilogues under the other's MFMAs; more halo per output):
//   C = 128: k = 11  7.45 vs 7.52 ms, k = 7  5.20 vs 5.17, k = 3  3.07 vs 2.90      C = 64: 4.46 vs 4.06, 3.32 vs 3.09, 2.36 vs 2.18
//   C = 32 : k = 11  3.55 vs 3.20, k = 7  2.80 vs 2.74, k = 3  1.92 vs 2.29           C = 256 (two workgroups do not fit): 3.56 / 2.37 / 1.21
// The table takes the faster one per class (VTTS_X3_GEOM: 0 = this table, 1 = one workgroup everywhere, 2 = two workgroups wherever they fit).
//                                        C   KS   N1  WM WN  XC  WPS
#ifndef VTTS_X3_GEOM
#define VTTS_X3_GEOM 0
#endif
template <int KS> using X256 = XTile<256, KS, 128, 4, 2, 128, 2>;
template <int KS> using X128one = XTile<128, KS, 256, 2, 4, 128, 2>;
template <int KS> using X128two = XTile<128, KS, 128, 2, 2, 64, 2>;
template <int KS> using X64one = XTile<64, KS, 512, 1, 8, 64, 2>;
template <int KS> using X64two = XTile<64, KS, 256, 1, 4, 64, 2>;
template <int KS> using X32one = XTile<32, KS, 512, 1, 8, 32, 2>;
template <int KS> using X32two = XTile<32, KS, 512, 1, 4, 32, 2>;
#if VTTS_X3_GEOM == 1
template <int KS> using X128 = X128one<KS>;
template <int KS> using X64 = X64one<KS>;
template <int KS> using X32 = X32one<KS>;
#elif VTTS_X3_GEOM == 2
template <int KS> using X128 = X128two<KS>;
template <int KS> using X64 = X64two<KS>;
template <int KS> using X32 = X32two<KS>;
#else
template <int KS> using X128 = std::conditional_t<KS == 3, X128two<KS>, X128one<KS>>;
template <int KS> using X64 = X64two<KS>;
template <int KS> using X32 = std::conditional_t<KS == 3, X32one<KS>, X32two<KS>>;
#endif

template <class T>
static hipError_t launch_x3(const PairArgsX3& p, hipStream_t s) {
    static DynLdsOnce once;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&resblock_pair_x3_k<T>), T::lds_bytes(T::MAXDIL), once); e != hipSuccess) return e;
    dim3 grid((p.a.L + T::NT2 - 1) / T::NT2, 1, p.a.B);
    if ((int)grid.x >= X3_XCD_MIN_TILES) grid.x = (grid.x + 7) / 8 * 8;
    hipLaunchKernelGGL(resblock_pair_x3_k<T>, grid, dim3(T::THREADS), T::lds_bytes(p.a.dil), s, p);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_x3_ks(const PairArgsX3& p, int K, hipStream_t s) {
    switch (K) {
        case 3: return launch_x3<TT<3>>(p, s);
        case 7: return launch_x3<TT<7>>(p, s);
        case 11: return launch_x3<TT<11>>(p, s);
    }
    return hipErrorInvalidValue;
}

bool pair_x3_supported(int C, int K, int dil, int L) {
    return (C == 256 || C == 128 || C == 64 || C == 32) && (K == 3 || K == 7 || K == 11) && dil >= 1 && dil <= 5 && L >= 1 && (long)C * L < (1l << 31);
}

// bytes of one convolution's packed weights: hi plane + lo plane, each in pair_g_pack_geom(C, K) order
size_t pair_x3_conv_bytes(int C, int K) { return 2 * (size_t)K * C * C * 2; }

// Haiku [K][Cin][Cout] fp32 -> [hi fragments][lo fragments] (bf16, A-fragment order of the bf16 pair kernel)
void pair_x3_pack(const float* w_hk, int C, int K, unsigned short* out) {
    const size_t n = (size_t)K * C * C;
    float* hi = new float[2 * n];
    float* lo = hi + n;
    for (size_t i = 0; i < n; ++i) {
        unsigned u;
        memcpy(&u, &w_hk[i], 4);
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;  // round to nearest even
        memcpy(&hi[i], &u, 4);
        lo[i] = w_hk[i] - hi[i];  // exact; bf16_pack rounds it to bf16
    }
    const BPackGeom g = pair_g_pack_geom(C, K);
    bf16_pack(hi, C, g, out);
    bf16_pack(lo, C, g, out + n);
    delete[] hi;
}

hipError_t launch_pair_x3(const ConvArgs& a, const void* w1, const void* w2, const float* bias2, hipStream_t s) {
    PairArgsX3 p{a, w1, w2, bias2};
    switch (a.Cin) {
        case 256: return launch_x3_ks<X256>(p, a.K, s);
        case 128: return launch_x3_ks<X128>(p, a.K, s);
        case 64: return launch_x3_ks<X64>(p, a.K, s);
        case 32: return launch_x3_ks<X32>(p, a.K, s);
    }
    return hipErrorInvalidValue;
}


// =====================================================================================================
// The four transposed convolutions (model.py:112-114: x = ups_i(leaky_relu(x, 0.1))) with split operands, polyphase form (k = 2 * stride;
// kernels_f32_mfma.hip: output p = s q + r of group g = r / (s/2) reads input frames q + g - 1 + m, m = 0, 1, through tap
// j = s (g - 1 + m) + pad_a - r): per group a GEMM  Y_g[m' = co * SH + ph, q] = sum_{m, ci} W_g[m'][(m, ci)] lrelu(x)[ci][q + g - 1 + m].
// Same staging as the pair kernel (fp32 channel-major in, hi / lo channels-last tiles of N1 + 2 frames); a wave owns one 32-row block of m' in BOTH
// groups (m-block index = group), so the three distinct B fragments (frames q - 1, q, q + 1) are read once per k-step and, in the accumulator layout,
// a lane ends up with consecutive output samples of one channel: SH = 4 -> a float4 per group (16 contiguous bytes of y[co][8 q + 4 g ..]),
// SH = 1 -> a float2 of both groups (y[co][2 q], y[co][2 q + 1]).  fp32 in HBM on both sides, as everything outside the matrix products.
// =====================================================================================================
#ifndef VTTS_UX_STAGED
#define VTTS_UX_STAGED 1
#endif
template <int CIN_, int COUT_, int SH_, int N1_, int WM_, int WN_>
struct UXTile {
    static constexpr int CIN = CIN_, COUT = COUT_, SH = SH_, S = 2 * SH_, N1 = N1_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 64 * WM * WN, NWAVES = WM * WN;
    static constexpr int MPG = COUT * SH;              // GEMM rows per group
    static constexpr int MT = WM * 32;                 // m' rows per workgroup (per group)
    static constexpr int NR = N1 / WN / 32;
    static constexpr int SPR = CIN / 8, P = CIN * 2, KSTEPS = CIN / 16, MB = MPG / 32;
    static constexpr int ROWS = N1 + 2;                // frames t0 - 1 .. t0 + N1
    static constexpr int PLANE = tile_rows16(ROWS) * P;
    static constexpr int LDS_BYTES = 2 * PLANE;
    static constexpr size_t PLANE_W = (size_t)4 * KSTEPS * MB * 1024;  // bytes of one weight plane: [g][m][ks][mblk][lane][8]
    // SH = 1: the output rows leave through an fp32 transposition area in the planes' LDS (kernel-development switch VTTS_UX_STAGED, default on)
    static constexpr int FS = 2 * N1 + 4;
    static constexpr bool STAGED = VTTS_UX_STAGED && SH == 1 && (size_t)MT * FS * 4 <= (size_t)LDS_BYTES;
    static_assert(MPG % MT == 0 && N1 % (WN * 32) == 0 && (SH == 4 || SH == 1) && KSTEPS % 2 == 0, "tiling");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <class T>
__global__ __launch_bounds__(T::THREADS, 2) void convt_x3_k(ConvArgs a) {
    constexpr int CIN = T::CIN, COUT = T::COUT, SH = T::SH, S = T::S, N1 = T::N1, WN = T::WN, NR = T::NR;
    constexpr int SPR = T::SPR, KSTEPS = T::KSTEPS, MB = T::MB, NWAVES = T::NWAVES, ROWS = T::ROWS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const thi = lds;
    unsigned char* const tlo = lds + T::PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, lh = lane >> 5;
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int LP = a.L, Lout = a.Lout;  // row pitches of x and y
    const int L = valid_len(a, b);      // this utterance's input frames (ragged batches; == LP otherwise)
    const int t0 = blockIdx.x * N1;  // first input frame of this workgroup
    if (t0 >= L) return;
    const int mblk = blockIdx.y * T::WM + wm;  // this wave's 32-row block of m' (in both groups)
    const float* __restrict__ xb = a.x + (long)b * CIN * LP;
    const float slope = a.slope_in;
    auto split2 = [](float v0, float v1, unsigned& hi, unsigned& lo) {
        hi = pack_bf16x2(v0, v1);
        lo = pack_bf16x2(v0 - bf16_lo(hi), v1 - bf16_hi(hi));
    };
    // ---- input tile: frames t0 - 1 .. t0 + N1 (zero outside the utterance: lax "SAME"), lane <-> frame, 8 channels per unit ----
    {
        constexpr int NBLK = (ROWS + 63) / 64, UNITS = NBLK * SPR, UB = 4;
        for (int u0 = wave * UB; u0 < UNITS; u0 += NWAVES * UB) {
            float v[UB][8];
            int row[UB], slot[UB];
            bool live[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int u = u0 + q < UNITS ? u0 + q : UNITS - 1;
                live[q] = u0 + q < UNITS;
                slot[q] = u % SPR;
                row[q] = (u / SPR) * 64 + lane;
                const int t = t0 - 1 + row[q];
                const bool ok = live[q] && row[q] < ROWS && t >= 0 && t < L;
                const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
                const float* __restrict__ g = xb + (long)(slot[q] * 8) * LP + tc;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q][e] = g[(long)e * LP];
                if (!ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[q][e] = 0.0f;
                }
            }
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                if (!live[q] || row[q] >= ROWS) continue;
                uint4 h4, l4;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[q][e] = lrelu(v[q][e], slope);
                split2(v[q][0], v[q][1], h4.x, l4.x);
                split2(v[q][2], v[q][3], h4.y, l4.y);
                split2(v[q][4], v[q][5], h4.z, l4.z);
                split2(v[q][6], v[q][7], h4.w, l4.w);
                const int off = tile_off<SPR>(row[q], slot[q]);
                *reinterpret_cast<uint4*>(thi + off) = h4;
                *reinterpret_cast<uint4*>(tlo + off) = l4;
            }
        }
    }
    __syncthreads();

    // ---- MFMA loop over the k-steps: A fragment (plane, g, m, ks): a.wp + plane*PLANE_W + ((((g*2 + m)*KSTEPS + ks)*MB + mblk)*64 + lane)*16 ----
    f32x16 acc[2][NR];  // [group][column block]
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][nr][r] = 0.0f;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, (int)(2 * T::PLANE_W), 0x00020000);
    const unsigned a_voff = (unsigned)(mblk * 64 + lane) * 16;
    bf16x8 af[2][4][2], bf[2][3][NR][2];  // A: [ring slot][g*2 + m][plane]; B: [parity][frame offset f][column block][plane]
    auto load_a = [&](int ks, int slot) {
        const int kc = ks < KSTEPS ? ks : KSTEPS - 1;
#pragma unroll
        for (int gm = 0; gm < 4; ++gm) {
            const int soff = ((gm * KSTEPS + kc) * MB) * 1024;
            af[slot][gm][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff, soff, 0));
            af[slot][gm][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff, soff + (int)T::PLANE_W, 0));
        }
    };
    const int rowbase0 = wn * (N1 / WN) + l31;  // tile row of frame q - 1 for this lane's column q of block 0
    auto load_b = [&](int ks, int par) {
        const int kc = ks < KSTEPS ? ks : KSTEPS - 1;
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int off = tile_off<SPR>(rowbase0 + f + nr * 32, kc * 2 + lh);
                bf[par][f][nr][0] = *reinterpret_cast<const bf16x8*>(thi + off);
                bf[par][f][nr][1] = *reinterpret_cast<const bf16x8*>(tlo + off);
            }
    };
    load_a(0, 0);
    load_b(0, 0);
#pragma unroll 1
    for (int ks0 = 0; ks0 < KSTEPS; ks0 += 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            load_a(ks0 + i + 1, (i + 1) & 1);
            load_b(ks0 + i + 1, (i + 1) & 1);
            const int sl = i & 1;
            // group g, tap m reads frame offset f = g + m; small terms first
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int gm = 0; gm < 4; ++gm)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int g = gm >> 1, f = g + (gm & 1);
                        const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
                        acc[g][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[sl][gm][pa], bf[sl][f][nr][pb], acc[g][nr], 0, 0, 0);
                    }
            constexpr int NMF = 12 * NR, NA = 8, NB = 6 * NR;
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
    }

    // ---- epilogue: + bias, fp32 stores of consecutive output samples of one channel ----
    float* yb = a.y + (long)b * COUT * Lout;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int q = t0 + wn * (N1 / WN) + nr * 32 + l31;
        if (q >= L) continue;
        if constexpr (SH == 4) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int co = mblk * 8 + 2 * rq + lh;  // rows 8 rq + 4 lh + i of the block = (co, ph = i)
                    const float bv = a.bias[co];
                    const float4 o = make_float4(acc[g][nr][4 * rq + 0] + bv, acc[g][nr][4 * rq + 1] + bv, acc[g][nr][4 * rq + 2] + bv, acc[g][nr][4 * rq + 3] + bv);
                    *reinterpret_cast<float4*>(yb + (long)co * Lout + (long)S * q + 4 * g) = o;
                }
        } else if constexpr (!T::STAGED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float bv = a.bias[co];
                *reinterpret_cast<float2*>(yb + (long)co * Lout + 2l * q) = make_float2(acc[0][nr][r] + bv, acc[1][nr][r] + bv);
            }
        }
    }
    if constexpr (SH == 1 && T::STAGED) {
        // Round 6: the stride-2 upsamplers are HBM-bound (3.6-3.9 TB/s) and stored 8 bytes per lane, 32 lanes = 256 bytes per channel row and instruction.
        // The output rows go through an fp32 area [MT][2 N1 + 4] in the (dead) planes' LDS and leave as 16-byte units, a wave = 1 KiB of one row — the bf16
        // engine's UTile epilogue (kernels_bf16_up.hip).  Same sums, same bias addition: the same bits.
        constexpr int FS = T::FS;
        float* const fs = reinterpret_cast<float*>(lds);
        __syncthreads();  // every wave is done reading the tiles
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int ql = wn * (N1 / WN) + nr * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float bv = a.bias[blockIdx.y * T::MT + col];
                *reinterpret_cast<float2*>(fs + col * FS + 2 * ql) = make_float2(acc[0][nr][r] + bv, acc[1][nr][r] + bv);
            }
        }
        __syncthreads();
        constexpr int UPR = N1 / 2;  // 16-byte units per output row of the tile (2 N1 samples)
        const long s0 = 2l * t0, send = 2l * L;
        for (int u = tid; u < T::MT * UPR; u += T::THREADS) {
            const int col = u / UPR, x4 = u - col * UPR;
            const long sidx = s0 + 4l * x4;
            float* dst = yb + (long)(blockIdx.y * T::MT + col) * Lout + sidx;
            const float4 v = *reinterpret_cast<const float4*>(fs + col * FS + 4 * x4);
            if (sidx + 3 < send) *reinterpret_cast<float4*>(dst) = v;
            else if (sidx + 1 < send) *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
        }
    }
}

//                                      CIN COUT SH  N1  WM WN
using UX0 = UXTile<512, 256, 4, 64, 8, 1>;
using UX1 = UXTile<256, 128, 4, 128, 4, 2>;
using UX2 = UXTile<128, 64, 1, 256, 2, 4>;
using UX3 = UXTile<64, 32, 1, 512, 1, 8>;

static int convt_x3_class(int Cin, int Cout, int K, int stride) {
    if (Cin == 512 && Cout == 256 && K == 16 && stride == 8) return 0;
    if (Cin == 256 && Cout == 128 && K == 16 && stride == 8) return 1;
    if (Cin == 128 && Cout == 64 && K == 4 && stride == 2) return 2;
    if (Cin == 64 && Cout == 32 && K == 4 && stride == 2) return 3;
    return -1;
}
bool convt_x3_supported(int Cin, int Cout, int K, int stride, int pad_a, int L) {
    if (convt_x3_class(Cin, Cout, K, stride) < 0 || L < 1) return false;
    for (int r = 0; r < stride; ++r) {  // the (q-1, q) / (q, q+1) polyphase split: every phase's two taps must exist
        const int g = r / (stride / 2);
        for (int m = 0; m < 2; ++m) {
            const int j = stride * (g - 1 + m) + pad_a - r;
            if (j < 0 || j >= K) return false;
        }
    }
    return true;
}
size_t convt_x3_bytes(int Cin, int Cout, int stride) { return 2 * (size_t)4 * (Cin / 16) * (Cout * (stride / 2) / 32) * 1024; }

// Haiku [K][Cout][Cin] fp32 -> [plane][g][m][ks][mblk][lane][8] bf16: row m' = mblk*32 + (lane & 31) = co*SH + ph, k = ks*16 + 8*(lane >> 5) + e = ci
void convt_x3_pack(const float* w_hk, int Cin, int Cout, int K, int stride, int pad_a, unsigned short* out) {
    const int SH = stride / 2, MB = Cout * SH / 32, KSTEPS = Cin / 16;
    const size_t plane = (size_t)4 * KSTEPS * MB * 512;  // elements
    for (int gm = 0; gm < 4; ++gm)
        for (int ks = 0; ks < KSTEPS; ++ks)
            for (int mb = 0; mb < MB; ++mb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int mp = mb * 32 + (lane & 31), co = mp / SH, ph = mp % SH, g = gm >> 1, m = gm & 1;
                        const int ci = ks * 16 + 8 * (lane >> 5) + e;
                        const int j = stride * (g - 1 + m) + pad_a - (g * SH + ph);
                        const float w = w_hk[((size_t)j * Cout + co) * Cin + ci];
                        unsigned u;
                        memcpy(&u, &w, 4);
                        const unsigned uh = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
                        float hi;
                        memcpy(&hi, &uh, 4);
                        const float lo = w - hi;
                        unsigned ul;
                        memcpy(&ul, &lo, 4);
                        ul = (ul + 0x7fffu + ((ul >> 16) & 1u)) >> 16;
                        const size_t o = ((((size_t)gm * KSTEPS + ks) * MB + mb) * 64 + lane) * 8 + e;
                        out[o] = (unsigned short)(uh >> 16);
                        out[plane + o] = (unsigned short)ul;
                    }
}

template <class T>
static hipError_t launch_ux(const ConvArgs& a, hipStream_t s) {
    static DynLdsOnce once;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&convt_x3_k<T>), T::LDS_BYTES, once); e != hipSuccess) return e;
    dim3 grid((a.L + T::N1 - 1) / T::N1, T::MPG / T::MT, a.B);
    hipLaunchKernelGGL(convt_x3_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

// a.x [B][Cin][L] fp32, a.y [B][Cout][stride*L], a.wp = convt_x3_pack, a.bias [Cout], a.slope_in = the LeakyReLU on the input
hipError_t launch_convt_x3(const ConvArgs& a, hipStream_t s) {
    switch (convt_x3_class(a.Cin, a.Cout, a.K, a.stride)) {
        case 0: return launch_ux<UX0>(a, s);
        case 1: return launch_ux<UX1>(a, s);
        case 2: return launch_ux<UX2>(a, s);
        case 3: return launch_ux<UX3>(a, s);
    }
    return hipErrorInvalidValue;
}


// =====================================================================================================
// conv_pre (model.py:83,110: hk.Conv1D(512, 7, padding 3) on the mel, no activation in front) with split operands — the split engine's last fp32
// MFMA launch until round 5.  mel [B][T][80] fp32 NWC (the boundary layout: a tile's rows are one contiguous run of 320-byte rows) ->
// y [B][512][T] fp32 channel-major.  GEMM: M = 512 output channels, N = time, K = 7 taps x 80 mel bins = 35 k-steps of 16.
// One workgroup = 8 waves = all 512 channels x 64 time steps (wave tile 64 x 64); the mel tile (70 rows x 80 bins) is split while staging into
// a HI and a LO channels-last bf16 tile (10 slots of 16 bytes per row, stored in blocks of 16 rows as bf16_common.h: tile_off does for narrow rows);
// the taps are shifted row views of it; weights [plane][tap][ks][mblk][lane][8] straight from L2 (1.1 MB, every workgroup reads all of it:
// L2 / Infinity-Cache resident); epilogue: + bias, fp32 stores in the accumulator layout (a half-wave = 32 consecutive time steps of one channel).
// =====================================================================================================
struct PreX3 {
    static constexpr int CIN = 80, COUT = 512, KS = 7, H = 3, N1 = 64, WM = 8, WN = 1;
    static constexpr int THREADS = 64 * WM * WN, MR = COUT / WM / 32, NR = N1 / WN / 32;
    static constexpr int SPR = CIN / 8, KSTEPS = CIN / 16, MB = COUT / 32;
    static constexpr int ROWS = N1 + 2 * H;
    static constexpr int PLANE = tile_rows16(ROWS) * CIN * 2;
    static constexpr int LDS_BYTES = 2 * PLANE;
    static constexpr size_t PLANE_W = (size_t)KS * KSTEPS * MB * 1024;  // bytes of one weight plane
    static constexpr int NSTEPS = KS * KSTEPS;
};

__global__ __launch_bounds__(PreX3::THREADS, 2) void conv_pre_x3_k(ConvArgs a) {
    using T = PreX3;
    constexpr int CIN = T::CIN, COUT = T::COUT, N1 = T::N1, MR = T::MR, NR = T::NR, SPR = T::SPR, KSTEPS = T::KSTEPS, MB = T::MB, ROWS = T::ROWS, NSTEPS = T::NSTEPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const thi = lds;
    unsigned char* const tlo = lds + T::PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int LP = a.L;             // frames allocated per utterance (row pitch of y)
    const int L = valid_len(a, b);  // this utterance's frames
    const int t0 = blockIdx.x * N1;
    if (t0 >= L) return;
    const float* __restrict__ xb = a.x + (long)b * a.x_sb;  // [T][80]
    const float slope = a.slope_in;                           // 1 for conv_pre (identity); kept general
    auto split2 = [](float v0, float v1, unsigned& hi, unsigned& lo) {
        hi = pack_bf16x2(v0, v1);
        lo = pack_bf16x2(v0 - bf16_lo(hi), v1 - bf16_hi(hi));
    };
    // ---- mel tile: frames t0 - 3 .. t0 + N1 + 2 (zero outside the utterance), unit = (row, 8 bins): two float4 loads of a contiguous row ----
    for (int u = tid; u < ROWS * SPR; u += T::THREADS) {
        const int row = u / SPR, slot = u - row * SPR;
        const int t = t0 - T::H + row;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (t >= 0 && t < L) {
            const float4* g = reinterpret_cast<const float4*>(xb + (long)t * CIN + slot * 8);  // 320-byte rows: 16-byte aligned
            v0 = g[0];
            v1 = g[1];
        }
        uint4 h4, l4;
        split2(lrelu(v0.x, slope), lrelu(v0.y, slope), h4.x, l4.x);
        split2(lrelu(v0.z, slope), lrelu(v0.w, slope), h4.y, l4.y);
        split2(lrelu(v1.x, slope), lrelu(v1.y, slope), h4.z, l4.z);
        split2(lrelu(v1.z, slope), lrelu(v1.w, slope), h4.w, l4.w);
        const int off = tile_off<SPR>(row, slot);
        *reinterpret_cast<uint4*>(thi + off) = h4;
        *reinterpret_cast<uint4*>(tlo + off) = l4;
    }
    __syncthreads();

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, (int)(2 * T::PLANE_W), 0x00020000);
    const unsigned a_voff = (unsigned)((wave * MR) * 64 + lane) * 16;
    bf16x8 af[2][MR][2], bf[2][NR][2];
    auto load_a = [&](int s, int slot) {
        const int sc = s < NSTEPS ? s : NSTEPS - 1;
        const int soff = (sc * MB) * 1024;  // step = tap * KSTEPS + ks
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            af[slot][mr][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + mr * 1024, soff, 0));
            af[slot][mr][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + mr * 1024, soff + (int)T::PLANE_W, 0));
        }
    };
    auto load_b = [&](int s, int par) {
        const int sc = s < NSTEPS ? s : NSTEPS - 1;
        const int tap = sc / KSTEPS, ks = sc - tap * KSTEPS;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int off = tile_off<SPR>(l31 + tap + nr * 32, ks * 2 + lh);  // output column n, tap j reads tile row n + j
            bf[par][nr][0] = *reinterpret_cast<const bf16x8*>(thi + off);
            bf[par][nr][1] = *reinterpret_cast<const bf16x8*>(tlo + off);
        }
    };
    load_a(0, 0);
    load_b(0, 0);
#pragma unroll 1
    for (int s = 0; s < NSTEPS; ++s) {
        const int cur = s & 1;
        // (the ring slot is a runtime parity here: 35 steps; both branches are the same code on swapped registers)
        if (cur == 0) {
            load_a(s + 1, 1);
            load_b(s + 1, 1);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mr][term == 0 ? 1 : 0], bf[0][nr][term == 1 ? 1 : 0], acc[mr][nr], 0, 0, 0);
        } else {
            load_a(s + 1, 0);
            load_b(s + 1, 0);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][mr][term == 0 ? 1 : 0], bf[1][nr][term == 1 ? 1 : 0], acc[mr][nr], 0, 0, 0);
        }
    }
    // ---- epilogue: + bias, fp32 channel-major stores ----
    float* yb = a.y + (long)b * COUT * LP;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int t = t0 + nr * 32 + l31;
            if (t >= L) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (wave * MR + mr) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                yb[(long)co * LP + t] = acc[mr][nr][r] + a.bias[co];
            }
        }
}

bool conv_pre_x3_supported(int Cin, int Cout, int K, int dil) { return Cin == PreX3::CIN && Cout == PreX3::COUT && K == PreX3::KS && dil == 1; }
size_t conv_pre_x3_bytes() { return 2 * PreX3::PLANE_W; }

// Haiku [K][Cin][Cout] fp32 -> [plane][tap][ks][mblk][lane][8] bf16: row m = mblk*32 + (lane & 31) = co, k = ks*16 + 8*(lane >> 5) + e = ci
void conv_pre_x3_pack(const float* w_hk, unsigned short* out) {
    using T = PreX3;
    const size_t plane = T::PLANE_W / 2;  // elements
    for (int tap = 0; tap < T::KS; ++tap)
        for (int ks = 0; ks < T::KSTEPS; ++ks)
            for (int mb = 0; mb < T::MB; ++mb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = mb * 32 + (lane & 31), ci = ks * 16 + 8 * (lane >> 5) + e;
                        const float w = w_hk[((size_t)tap * T::CIN + ci) * T::COUT + co];
                        unsigned u;
                        memcpy(&u, &w, 4);
                        const unsigned uh = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
                        float hi;
                        memcpy(&hi, &uh, 4);
                        const float lo = w - hi;
                        unsigned ul;
                        memcpy(&ul, &lo, 4);
                        ul = (ul + 0x7fffu + ((ul >> 16) & 1u)) >> 16;
                        const size_t o = ((((size_t)tap * T::KSTEPS + ks) * T::MB + mb) * 64 + lane) * 8 + e;
                        out[o] = (unsigned short)(uh >> 16);
                        out[plane + o] = (unsigned short)ul;
                    }
}

// a.x = mel [B][L][80] (a.x_sb = batch stride), a.y [B][512][L], a.wp = conv_pre_x3_pack's output, a.bias [512]
hipError_t launch_conv_pre_x3(const ConvArgs& a, hipStream_t s) {
    static DynLdsOnce once;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_pre_x3_k), PreX3::LDS_BYTES, once); e != hipSuccess) return e;
    dim3 grid((a.L + PreX3::N1 - 1) / PreX3::N1, 1, a.B);
    hipLaunchKernelGGL(conv_pre_x3_k, grid, dim3(PreX3::THREADS), PreX3::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace vtts
