// fp32 implicit-GEMM convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//  * conv1d_f32_mfma_k   — dilated Conv1d.  The 72 ResBlock1 convolutions (96.8 % of the FLOPs;
//                          vietTTS/hifigan/model.py:21-28 convs1 with rate d, :33-40 convs2) and
//                          conv_pre (model.py:83, mel NWC in).
//  * convT1d_f32_mfma_k  — the four ConvTranspose1d upsamplers in polyphase form (model.py:88-94,
//                          SURVEY.md Appendix A.2): no zero-stuffing, every output phase uses exactly
//                          its two live taps.
//
// GEMM view:  Y[co, t] = sum_{j,ci} W[j][ci][co] * f(X[ci, t + j*d - p])
//   M = co (output channels)   N = t (time, the only long axis)   K = (tap j, input channel ci)
// fp32 MFMA is bit-identical to a k-ordered fmaf chain (cdna_hip_programming.md §3), so the only
// difference from the reference is summation order.
//
// Data movement per workgroup (256 threads = 4 waves, one MT x NT output tile of one utterance):
//   * activations are channel-major [B][C][L] in HBM, so a tile row is a contiguous run along
//     time: float4 global loads, LeakyReLU applied once while staging ("im2col into LDS" without
//     materialising it: the K taps are K shifted *views* of the same LDS rows);
//   * LDS tile xs[CK][NT + 2*PA] (PA = halo rounded up to 4) holds CK input channels; the MFMA B
//     fragment B[k = lane>>5][n = lane&31] is one conflict-free ds_read_b32 per lane
//     (32 consecutive dwords per half-wave);
//   * weights are pre-packed on the host in A-fragment order, four k-steps per lane contiguous,
//     so each wave pulls its A operands with one 1-KiB global_load_dwordx4 per 4 k-steps straight
//     from L2 (the whole model is 55.7 MB: L2/Infinity-Cache resident), software-prefetched one
//     iteration ahead;
//   * epilogue fuses bias, the ResBlock residual and the MRF accumulate/mean (device_common.h).
#include "device_common.h"

#ifndef VTTS_F32_PF  // A-fragment ring depth of conv1d_f32_mfma_k in iterations of 4 k-steps (A/B switch; 1 = round 3's one-ahead prefetch)
#define VTTS_F32_PF 4
#endif

namespace vtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int round_up4(int v) { return (v + 3) & ~3; }
constexpr int MAX_DIL = 5;  // resblock_dilation_sizes max in V1; larger rates fall back to the generic kernel
constexpr int F32_XCD_MIN_TILES = 64;  // XCD-aware tile order of conv1d_f32_mfma_k from this many time tiles per grid row on

// =================================================================================================
// dilated Conv1d
// =================================================================================================
template <int CIN_, int COUT_, int KS_, int MT_, int NT_, int WM_, int WN_, int CK_, bool NWC_>
struct ConvTile {
    static constexpr int CIN = CIN_, COUT = COUT_, KS = KS_, MT = MT_, NT = NT_, WM = WM_, WN = WN_, CK = CK_;
    static constexpr bool NWC = NWC_;                 // input is [B][L][CIN] (conv_pre reads the mel as given)
    static constexpr int MR = MT / WM / 32;           // 32x32 accumulator blocks per wave along M
    static constexpr int NR = NT / WN / 32;           // ... along N
    static constexpr int PA = round_up4((KS - 1) / 2 * (NWC ? 1 : MAX_DIL));  // staged halo per side
    static constexpr int W = NT + 2 * PA;             // staged columns (multiple of 4)
    static constexpr int RS = NWC ? W + 1 : W;        // LDS row stride in floats (odd for the transposing stage)
    static constexpr int NCH = CIN / CK;              // input-channel chunks
    static constexpr int CQ = CK / 8;                 // float4 A loads per tap per chunk
    static constexpr int NIT = KS * CQ;               // main-loop iterations per chunk
    static constexpr int LDS_BYTES = CK * RS * 4;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MT % (WM * 32) == 0 && NT % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(COUT % MT == 0 && CIN % CK == 0 && CK % 8 == 0, "channel tiling");
};

// Packed weight layout (floats): [mblk = COUT/32][chunk][j][cq][lane = 64][e = 4]
//   element = W_hk[j][ci = chunk*CK + 2*(4*cq + e) + (lane >> 5)][co = mblk*32 + (lane & 31)]
// i.e. for k-step (j, cp = 4*cq + e) lane l holds A[i = l&31][k = l>>5] of the 32x32x2 MFMA.
template <class T>
__global__ __launch_bounds__(256, 3) void conv1d_f32_mfma_k(ConvArgs a) {  // register budget of three workgroups per CU (what the 47 KB tiles of C >= 128 allow)
    constexpr int COUT = T::COUT, MT = T::MT, NT = T::NT, WN = T::WN, CK = T::CK;
    constexpr int MR = T::MR, NR = T::NR, PA = T::PA, W = T::W, RS = T::RS, NCH = T::NCH, CQ = T::CQ, NIT = T::NIT;

    extern __shared__ __attribute__((aligned(16))) float xs[];  // [CK][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int LP = a.L;              // row pitch of x / res / y
    const int L = valid_len(a, b);   // this utterance's columns (ragged batches; == LP otherwise)
    // XCD-aware tile order (as the bf16 pair kernel, kernels_bf16_rbg.hip): workgroups go to the 8 XCDs round-robin in launch order, so on
    // launches of at least F32_XCD_MIN_TILES tiles per row of the grid (gridDim.x then padded to a multiple of 8 by the launcher) XCD
    // blockIdx.x % 8 takes a contiguous, balanced eighth of the time tiles: a tile's halo columns were staged by the same L2's previous tile
    int tile = blockIdx.x;
    if (gridDim.x >= F32_XCD_MIN_TILES) {
        const int nt = (L + NT - 1) / NT, r = (int)((blockIdx.x + blockIdx.z) & 7), lo = (r * nt) >> 3, hi = ((r + 1) * nt) >> 3;
        tile = lo + (int)(blockIdx.x >> 3);
        if (tile >= hi) return;
    }
    const int t0 = tile * NT;
    if (t0 >= L) return;  // a tile past the utterance's end (ragged batches)
    const int m0 = blockIdx.y * MT + wm * (MT / T::WM);  // first output channel of this wave
    const float slope = a.slope_in;
    const float* __restrict__ xb = a.x + (long)b * a.x_sb;

    // per-(mr) base of this wave's packed A stream; consecutive main-loop iterations are contiguous
    const float4* __restrict__ wbase[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int mblk = m0 / 32 + mr;
        wbase[mr] = reinterpret_cast<const float4*>(a.wp) + (long)mblk * (NCH * NIT) * 64 + lane;
    }

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;

    // A fragments: a ring of PF iterations (4 k-steps each) in flight per m-block, running across the channel chunks (round 4: one iteration
    // ahead left an L2 round trip exposed whenever fewer than ~4 waves per SIMD were there to cover it); B fragments are read one k-step ahead.
    constexpr int PF = NIT % VTTS_F32_PF == 0 ? VTTS_F32_PF : (NIT % 2 == 0 ? 2 : 1);  // conv_pre: 70 iterations per chunk
    static_assert(NIT % PF == 0, "ring slots are compile-time positions in the block loop");
    constexpr long TOTAL = (long)NCH * NIT;
    float4 a_ring[PF][MR];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) a_ring[u][mr] = wbase[mr][(u < TOTAL ? u : TOTAL - 1) * 64];

    const int dil = a.dil;
    const int colbase = wn * (NT / WN) + l31 - a.pad + PA;  // + j*dil + nr*32

    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk) __syncthreads();  // all waves done reading the previous chunk
        // ---- stage CK input channels x W columns, LeakyReLU fused, zero outside [0, L) ----
        if constexpr (!T::NWC) {
            constexpr int W4 = W / 4;
            const float* __restrict__ xc = xb + (long)(chunk * CK) * a.x_sc;
            for (int idx = tid; idx < CK * W4; idx += 256) {
                const int row = idx / W4;
                const int c4 = idx - row * W4;
                const int t = t0 - PA + 4 * c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t >= 0 && t < L) v = mask_tail4(*reinterpret_cast<const float4*>(xc + (long)row * a.x_sc + t), t, L);
                v.x = lrelu(v.x, slope);
                v.y = lrelu(v.y, slope);
                v.z = lrelu(v.z, slope);
                v.w = lrelu(v.w, slope);
                *reinterpret_cast<float4*>(&xs[row * RS + 4 * c4]) = v;
            }
        } else {
            // time-major input [L][CIN]: coalesced reads along channels, transposing LDS writes
            // (row stride RS is odd, so the stride-RS ds_write_b32 is conflict-free)
            const float* __restrict__ xc = xb + chunk * CK;
            for (int idx = tid; idx < CK * W; idx += 256) {
                const int col = idx / CK;
                const int row = idx - col * CK;
                const int t = t0 - PA + col;
                float v = 0.f;
                if (t >= 0 && t < L) v = xc[(long)t * a.x_st + row];
                xs[row * RS + col] = lrelu(v, slope);
            }
        }
        __syncthreads();

        // ---- MFMA over (tap j, channel pairs) of this chunk ----
        auto bload = [&](int it, int e, float (&dst)[NR]) {
            const int j = it / CQ;
            const int cq = it - j * CQ;
            const float* xr = &xs[(cq * 8 + 2 * e + lh) * RS + colbase + j * dil];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) dst[nr] = xr[nr * 32];
        };
        float bf[2][NR];
        bload(0, 0, bf[0]);
#pragma unroll 1
        for (int it0 = 0; it0 < NIT; it0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int it = it0 + u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int par = (u * 4 + e) & 1;
                    if (e < 3) bload(it, e + 1, bf[par ^ 1]);
                    else if (it + 1 < NIT) bload(it + 1, 0, bf[par ^ 1]);
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) {
                        const float av = e == 0 ? a_ring[u][mr].x : e == 1 ? a_ring[u][mr].y : e == 2 ? a_ring[u][mr].z : a_ring[u][mr].w;
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[par][nr], acc[mr][nr], 0, 0, 0);
                    }
                }
                const long q = (long)chunk * NIT + it + PF;  // refill this slot PF iterations ahead (the tail re-reads the last fragment: in bounds, never used)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) a_ring[u][mr] = wbase[mr][(q < TOTAL ? q : TOTAL - 1) * 64];
            }
        }
    }

    // ---- epilogue: C/D layout col = lane&31 (time), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel) ----
    // The operations of device_common.h: epilogue_store in the same order (v = acc + bias; v = v + res; ACC_ADD: v = y + v; ACC_MEAN:
    // v = (y + v) / div), but with every residual / accumulator value of a 32 x 32 block REQUESTED before the block's first store (round 4):
    // a.res and a.y may alias as far as the compiler knows, so the per-element form was 16 dependent load -> add -> store round trips per block.
    const int mode = a.acc_mode;
    const float dv = a.div;
    const bool has_res = a.res != nullptr;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int t = t0 + wn * (NT / WN) + nr * 32 + l31;
            const bool ok = t < L;
            const int tc = ok ? t : 0;  // a masked lane reads an in-bounds address of its own row and stores nothing
            if (!has_res && mode == ACC_STORE) {
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        a.y[((long)b * COUT + co) * LP + t] = acc[mr][nr][r] + a.bias[co];
                    }
                }
                continue;
            }
            // 8 values per round trip, 32-bit element offsets inside the utterance (COUT * L < 2^31): the register budget of three workgroups per CU
            const float* resb = has_res ? a.res + (long)b * COUT * LP : a.y + (long)b * COUT * LP;  // no __restrict__: the un-fused ResBlock path runs in place (res == y)
            float* yb = a.y + (long)b * COUT * LP;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float rv[8], yv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = r0 + q;
                    const int off = (m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LP + tc;
                    rv[q] = has_res ? resb[off] : 0.0f;
                    yv[q] = mode != ACC_STORE ? yb[off] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = r0 + q;
                    const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int off = co * LP + tc;
                    float v = acc[mr][nr][r] + a.bias[co];
                    if (has_res) v = v + rv[q];
                    if (mode == ACC_ADD) v = yv[q] + v;
                    else if (mode == ACC_MEAN) v = (yv[q] + v) / dv;
                    if (ok) yb[off] = v;
                }
            }
        }
    }
}

// ---- tile selection --------------------------------------------------------------------------
//                                          CIN  COUT KS  MT   NT  WM WN CK  NWC
template <int KS> using Tile256 = ConvTile<256, 256, KS, 128, 128, 2, 2, 64, false>;
template <int KS> using Tile128 = ConvTile<128, 128, KS, 128, 128, 2, 2, 64, false>;
// C <= 64: the "narrow" geometry is also the one for large launches since round 3 — 30.7 / 23.5 KB of LDS per workgroup instead of 47 / 39, i.e. five or
// six workgroups per CU instead of three: these convolutions are bound by the latency of their staging / epilogue phases, not by halo bytes
// (64 x 1024 frames: 378.9 -> 367.2 ms per pass; twice as WIDE tiles 429 ms, at C >= 128 too 535 ms; narrow tiles at C >= 128 383 ms; the
// transposed convolutions' tiles half as wide: no change; gpurun_out/r03_exp45)
template <int KS> using Tile64 = ConvTile<64, 64, KS, 64, 64, 2, 2, 64, false>;
template <int KS> using Tile32 = ConvTile<32, 32, KS, 32, 128, 1, 4, 32, false>;
using TilePre = ConvTile<80, 512, 7, 128, 64, 4, 1, 80, true>;  // conv_pre: mel [T][80] -> [512][T]
// narrow time tiles for short inputs (batch-1 latency): same math, 4x / 2x more workgroups
template <int KS> using Tile256S = ConvTile<256, 256, KS, 128, 32, 4, 1, 64, false>;
template <int KS> using Tile128S = ConvTile<128, 128, KS, 128, 64, 2, 2, 64, false>;
template <int KS> using Tile64S = ConvTile<64, 64, KS, 64, 64, 2, 2, 64, false>;
template <int KS> using Tile32S = ConvTile<32, 32, KS, 32, 128, 1, 4, 32, false>;

template <class T>
static hipError_t launch_tile(const ConvArgs& a, hipStream_t s) {
    static DynLdsOnce once;  // per device (vtts_internal.h)
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv1d_f32_mfma_k<T>), T::LDS_BYTES, once); e != hipSuccess) return e;
    dim3 grid((a.L + T::NT - 1) / T::NT, T::COUT / T::MT, a.B);
    if ((int)grid.x >= F32_XCD_MIN_TILES) grid.x = (grid.x + 7) / 8 * 8;  // whole rounds of the 8 XCDs; with gridDim.x a multiple of 8, workgroup (x, y, z) runs on XCD x % 8
    hipLaunchKernelGGL(conv1d_f32_mfma_k<T>, grid, dim3(256), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_ks(const ConvArgs& a, hipStream_t s) {
    switch (a.K) {
        case 3: return launch_tile<TT<3>>(a, s);
        case 7: return launch_tile<TT<7>>(a, s);
        case 11: return launch_tile<TT<11>>(a, s);
    }
    return hipErrorInvalidValue;
}

static bool is_pre_shape(int Cin, int Cout, int K, int dil) { return Cin == 80 && Cout == 512 && K == 7 && dil == 1; }

bool conv1d_f32_mfma_supported(int Cin, int Cout, int K, int dil, int L, bool nwc) {
    if (nwc) return is_pre_shape(Cin, Cout, K, dil);
    if (Cin != Cout) return false;
    if (!(Cin == 256 || Cin == 128 || Cin == 64 || Cin == 32)) return false;
    if (!(K == 3 || K == 7 || K == 11)) return false;
    if (dil < 1 || dil > MAX_DIL) return false;
    if (L % 4 != 0) return false;  // float4 staging
    return true;
}

static int chunk_of(int Cin) { return Cin == 80 ? 80 : (Cin >= 64 ? 64 : 32); }

size_t conv1d_f32_mfma_packed_floats(int Cin, int Cout, int K) { return (size_t)Cin * Cout * K; }

void conv1d_f32_mfma_pack(const float* w_hk, int Cin, int Cout, int K, float* out) {
    const int CK = chunk_of(Cin), NCH = Cin / CK, CQ = CK / 8;
    size_t o = 0;
    for (int mblk = 0; mblk < Cout / 32; ++mblk)
        for (int chunk = 0; chunk < NCH; ++chunk)
            for (int j = 0; j < K; ++j)
                for (int cq = 0; cq < CQ; ++cq)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int ci = chunk * CK + 2 * (4 * cq + e) + (lane >> 5);
                            const int co = mblk * 32 + (lane & 31);
                            out[o++] = w_hk[((size_t)j * Cin + ci) * Cout + co];
                        }
}

// workgroups the wide tile would launch; below ~1.5 per CU the narrow tile wins (measured at B=1)
static long wide_wgs(const ConvArgs& a, int NT, int mtiles) { return ((long)(a.L + NT - 1) / NT) * mtiles * a.B; }

hipError_t launch_conv1d_f32_mfma(const ConvArgs& a, hipStream_t s) {
    if (a.x_st != 1) return launch_tile<TilePre>(a, s);
    constexpr long MIN_WGS = 384;
    auto narrow = [&](int NT, int mtiles) { return a.tile_pref == 2 || (a.tile_pref == 0 && wide_wgs(a, NT, mtiles) < MIN_WGS); };
    switch (a.Cin) {
        case 256: return narrow(128, 2) ? launch_ks<Tile256S>(a, s) : launch_ks<Tile256>(a, s);
        case 128: return narrow(128, 1) ? launch_ks<Tile128S>(a, s) : launch_ks<Tile128>(a, s);
        case 64: return narrow(128, 1) ? launch_ks<Tile64S>(a, s) : launch_ks<Tile64>(a, s);
        case 32: return narrow(256, 1) ? launch_ks<Tile32S>(a, s) : launch_ks<Tile32>(a, s);
    }
    return hipErrorInvalidValue;
}

const char* conv1d_f32_mfma_kernel_name(int C, int K) {
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "conv1d_f32_mfma_k<ConvTile<%d, %d, %d", C, C, K);
    return buf;
}

// =================================================================================================
// ConvTranspose1d, polyphase, k == 2*stride  (all four V1 upsamplers: (16,8),(16,8),(4,2),(4,2))
// =================================================================================================
// With pad_a from lax's "SAME" rule, output p = s*q + r reads input frames
//   r <  s/2 : (q-1, q)        ("group" g = 0)
//   r >= s/2 : (q,   q+1)      (g = 1)
// through taps j = s*(g - 1 + m) + pad_a - r, m = 0,1.  Each group is a GEMM
//   Yg[m' = co*SH + ph, q] = sum_{m, ci} Wg[m', (m, ci)] * f(X[ci, q + g - 1 + m]),  SH = s/2, r = g*SH + ph
// A wave keeps both groups' accumulators, so the three distinct B fragments (frames q-1, q, q+1) are
// read from LDS once and each lane ends up with s consecutive output samples of one channel:
// SH = 4 -> two float4 stores (32 contiguous bytes), SH = 1 -> one float2 store; fully coalesced
// although the GEMM's N axis (q) is strided by s in the output.
template <int CIN_, int COUT_, int SH_, int MT_, int NT_, int WM_, int WN_, int CK_>
struct ConvTTile {
    static constexpr int CIN = CIN_, COUT = COUT_, SH = SH_, MT = MT_, NT = NT_, WM = WM_, WN = WN_, CK = CK_;
    static constexpr int MP = COUT * SH;              // GEMM rows per group
    static constexpr int MR = MT / WM / 32;
    static constexpr int NR = NT / WN / 32;
    static constexpr int PA = 4;                      // 1-frame halo, rounded up for float4 staging
    static constexpr int W = NT + 2 * PA;
    static constexpr int RS = W;
    static constexpr int NCH = CIN / CK;
    static constexpr int CQ = CK / 8;
    static constexpr int LDS_BYTES = CK * RS * 4;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MP % MT == 0 && CIN % CK == 0 && CK % 8 == 0, "tiling");
    static_assert(SH == 4 || SH == 1, "stride 8 or 2");
};

// Packed layout (floats): [mblk = MP/32][chunk][cq][gm = 4][lane][e = 4],  gm = 2*g + m
//   element = W_hk[j(g, m, ph)][co][ci]  with  m' = mblk*32 + (lane&31) = co*SH + ph,
//             ci = chunk*CK + 2*(4*cq + e) + (lane>>5)
template <class T>
__global__ __launch_bounds__(256) void convT1d_f32_mfma_k(ConvArgs a) {
    constexpr int COUT = T::COUT, SH = T::SH, MT = T::MT, NT = T::NT, WN = T::WN, CK = T::CK;
    constexpr int MR = T::MR, NR = T::NR, PA = T::PA, W = T::W, RS = T::RS, NCH = T::NCH, CQ = T::CQ;
    constexpr int S = 2 * SH;

    extern __shared__ __attribute__((aligned(16))) float xs[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int q0 = blockIdx.x * NT;
    const int m0 = blockIdx.y * MT + wm * (MT / T::WM);
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int LP = a.L;              // row pitch of x (frames); y's is LP * S
    const int L = valid_len(a, b);   // this utterance's input frames (ragged batches; == LP otherwise)
    if (q0 >= L) return;
    const float slope = a.slope_in;
    const float* __restrict__ xb = a.x + (long)b * a.x_sb;

    const float4* __restrict__ wbase[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) wbase[mr] = reinterpret_cast<const float4*>(a.wp) + (long)(m0 / 32 + mr) * (NCH * CQ * 4) * 64 + lane;

    f32x16 acc[2][MR][NR];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][mr][nr][r] = 0.0f;

    const int colbase = wn * (NT / WN) + l31 + PA;  // frame q;  -1 / +1 for the neighbours

    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk) __syncthreads();
        {
            constexpr int W4 = W / 4;
            const float* __restrict__ xc = xb + (long)(chunk * CK) * a.x_sc;
            for (int idx = tid; idx < CK * W4; idx += 256) {
                const int row = idx / W4;
                const int c4 = idx - row * W4;
                const int t = q0 - PA + 4 * c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t >= 0 && t < L) v = mask_tail4(*reinterpret_cast<const float4*>(xc + (long)row * a.x_sc + t), t, L);  // a ragged utterance's frame count need not be a multiple of 4
                v.x = lrelu(v.x, slope);
                v.y = lrelu(v.y, slope);
                v.z = lrelu(v.z, slope);
                v.w = lrelu(v.w, slope);
                *reinterpret_cast<float4*>(&xs[row * RS + 4 * c4]) = v;
            }
        }
        __syncthreads();

        for (int cq = 0; cq < CQ; ++cq) {
            float4 aw[MR][4];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int gm = 0; gm < 4; ++gm) aw[mr][gm] = wbase[mr][((long)(chunk * CQ + cq) * 4 + gm) * 64];
            const float* xrow = &xs[(cq * 8 + lh) * RS + colbase];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bm[NR], bz[NR], bp[NR];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const float* p = xrow + e * 2 * RS + nr * 32;
                    bm[nr] = p[-1];
                    bz[nr] = p[0];
                    bp[nr] = p[1];
                }
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    float av[4];
#pragma unroll
                    for (int gm = 0; gm < 4; ++gm)
                        av[gm] = e == 0 ? aw[mr][gm].x : e == 1 ? aw[mr][gm].y : e == 2 ? aw[mr][gm].z : aw[mr][gm].w;
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        acc[0][mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bm[nr], acc[0][mr][nr], 0, 0, 0);
                        acc[1][mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bz[nr], acc[1][mr][nr], 0, 0, 0);
                        acc[0][mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bz[nr], acc[0][mr][nr], 0, 0, 0);
                        acc[1][mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bp[nr], acc[1][mr][nr], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: lane owns S consecutive samples y[co][S*q .. S*q + S-1] per (co) it holds ----
    const long Lout = (long)LP * S;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int q = q0 + wn * (NT / WN) + nr * 32 + l31;
            if (q < L) {
                if constexpr (SH == 4) {
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int co = (m0 + mr * 32 + 8 * rq + 4 * lh) / 4;
                        const float bv = a.bias[co];
                        float* yp = a.y + ((long)b * COUT + co) * Lout + (long)q * S;
                        float4 v0, v1;
                        v0.x = acc[0][mr][nr][4 * rq + 0] + bv;
                        v0.y = acc[0][mr][nr][4 * rq + 1] + bv;
                        v0.z = acc[0][mr][nr][4 * rq + 2] + bv;
                        v0.w = acc[0][mr][nr][4 * rq + 3] + bv;
                        v1.x = acc[1][mr][nr][4 * rq + 0] + bv;
                        v1.y = acc[1][mr][nr][4 * rq + 1] + bv;
                        v1.z = acc[1][mr][nr][4 * rq + 2] + bv;
                        v1.w = acc[1][mr][nr][4 * rq + 3] + bv;
                        *reinterpret_cast<float4*>(yp) = v0;
                        *reinterpret_cast<float4*>(yp + 4) = v1;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float bv = a.bias[co];
                        float2 v;
                        v.x = acc[0][mr][nr][r] + bv;
                        v.y = acc[1][mr][nr][r] + bv;
                        *reinterpret_cast<float2*>(a.y + ((long)b * COUT + co) * Lout + (long)q * 2) = v;
                    }
                }
            }
        }
    }
}

//                        CIN  COUT SH  MT   NT  WM WN CK
using TileUp0 = ConvTTile<512, 256, 4, 128, 64, 2, 2, 64>;
using TileUp1 = ConvTTile<256, 128, 4, 128, 64, 2, 2, 64>;
using TileUp2 = ConvTTile<128, 64, 1, 64, 128, 2, 2, 64>;
using TileUp3 = ConvTTile<64, 32, 1, 32, 256, 1, 4, 64>;

static int convT_tile_id(int Cin, int Cout, int K, int stride) {
    if (Cin == 512 && Cout == 256 && K == 16 && stride == 8) return 0;
    if (Cin == 256 && Cout == 128 && K == 16 && stride == 8) return 1;
    if (Cin == 128 && Cout == 64 && K == 4 && stride == 2) return 2;
    if (Cin == 64 && Cout == 32 && K == 4 && stride == 2) return 3;
    return -1;
}

// tap index for (group g, tap m, phase-in-group ph), or -1 if the polyphase split is not the
// "(q-1,q) / (q,q+1)" one this kernel implements
static int convT_tap(int K, int s, int pad_a, int g, int m, int ph) {
    const int r = g * (s / 2) + ph;
    const int j = s * (g - 1 + m) + pad_a - r;
    return (j >= 0 && j < K) ? j : -1;
}

bool convT1d_f32_mfma_supported(int Cin, int Cout, int K, int stride, int pad_a, int L) {
    if (convT_tile_id(Cin, Cout, K, stride) < 0) return false;
    if (K != 2 * stride || L % 4 != 0) return false;
    // every phase must own exactly taps {j(g,0), j(g,1)} — verify against the definition
    for (int r = 0; r < stride; ++r) {
        int j0 = ((pad_a - r) % stride + stride) % stride;
        const int g = r / (stride / 2), ph = r % (stride / 2);
        int cnt = 0;
        for (int j = j0; j < K; j += stride, ++cnt) {
            const int off = (r + j - pad_a) / stride;
            if (cnt > 1 || off != g - 1 + cnt || convT_tap(K, stride, pad_a, g, cnt, ph) != j) return false;
        }
        if (cnt != 2) return false;
    }
    return true;
}

size_t convT1d_f32_mfma_packed_floats(int Cin, int Cout, int K) { return (size_t)Cin * Cout * K; }

void convT1d_f32_mfma_pack(const float* w_hk, int Cin, int Cout, int K, int stride, int pad_a, float* out) {
    const int SH = stride / 2, MP = Cout * SH, CK = 64, NCH = Cin / CK, CQ = CK / 8;
    size_t o = 0;
    for (int mblk = 0; mblk < MP / 32; ++mblk)
        for (int chunk = 0; chunk < NCH; ++chunk)
            for (int cq = 0; cq < CQ; ++cq)
                for (int gm = 0; gm < 4; ++gm)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int mp = mblk * 32 + (lane & 31);
                            const int co = mp / SH, ph = mp % SH;
                            const int ci = chunk * CK + 2 * (4 * cq + e) + (lane >> 5);
                            const int j = convT_tap(K, stride, pad_a, gm >> 1, gm & 1, ph);
                            out[o++] = w_hk[((size_t)j * Cout + co) * Cin + ci];  // Haiku [K][Cout][Cin]
                        }
}

template <class T>
static hipError_t launch_ttile(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.L + T::NT - 1) / T::NT, T::MP / T::MT, a.B);
    hipLaunchKernelGGL(convT1d_f32_mfma_k<T>, grid, dim3(256), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_convT1d_f32_mfma(const ConvArgs& a, hipStream_t s) {
    switch (convT_tile_id(a.Cin, a.Cout, a.K, a.stride)) {
        case 0: return launch_ttile<TileUp0>(a, s);
        case 1: return launch_ttile<TileUp1>(a, s);
        case 2: return launch_ttile<TileUp2>(a, s);
        case 3: return launch_ttile<TileUp3>(a, s);
    }
    return hipErrorInvalidValue;
}

// =================================================================================================
// conv_post: LeakyReLU(0.01) -> Conv1d 32 -> 1, k = 7 -> tanh   (model.py:122-124)
// =================================================================================================
// Not GEMM-shaped (one output channel): a streaming reduction over the largest activation of the
// network.  Each thread produces 4 consecutive samples from three aligned float4 loads per input
// channel; weights (C*7 floats) sit in LDS.  HBM-bound: 4*C bytes read + 4 written per sample.
template <int C, int KS>
__global__ __launch_bounds__(256) void conv_post_k(ConvArgs a) {
    static_assert(KS == 7, "halo of 3 fits the [t-4, t+8) window");
    __shared__ float ws[KS * C];
    for (int i = threadIdx.x; i < KS * C; i += 256) ws[i] = a.w[i];  // Haiku [K][Cin][1]
    __syncthreads();
    const int b = blockIdx.y;
    const long LP = a.L;             // row pitch
    const long L = valid_len(a, b);  // this utterance's samples (ragged batches; == LP otherwise): the rest of its wav slot stays as the caller left it
    const float* __restrict__ xb = a.x + (long)b * a.x_sb;
    const float slope = a.slope_in;
    const float bias = a.bias[0];
    for (long t = ((long)blockIdx.x * 256 + threadIdx.x) * 4; t < L; t += (long)gridDim.x * 1024) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int ci = 0; ci < C; ++ci) {
            const float* xr = xb + (long)ci * a.x_sc + t;
            float v[12];
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 lo = (t >= 4) ? *reinterpret_cast<const float4*>(xr - 4) : z;
            const float4 mid = mask_tail4(*reinterpret_cast<const float4*>(xr), (int)t, (int)L);
            const float4 hi = (t + 4 < L) ? mask_tail4(*reinterpret_cast<const float4*>(xr + 4), (int)t + 4, (int)L) : z;
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w;
            v[4] = mid.x; v[5] = mid.y; v[6] = mid.z; v[7] = mid.w;
            v[8] = hi.x; v[9] = hi.y; v[10] = hi.z; v[11] = hi.w;
#pragma unroll
            for (int i = 1; i < 11; ++i) v[i] = lrelu(v[i], slope);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const float wv = ws[j * C + ci];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = fmaf(wv, v[o + j + 1], acc[o]);  // x[t + o + j - 3]
            }
        }
        float4 pre, out;
        pre.x = acc[0] + bias; pre.y = acc[1] + bias; pre.z = acc[2] + bias; pre.w = acc[3] + bias;
        out.x = tanhf(pre.x); out.y = tanhf(pre.y); out.z = tanhf(pre.z); out.w = tanhf(pre.w);
        const long idx = (long)b * LP + t;
        if (t + 4 <= L) {
            if (a.pre_act) *reinterpret_cast<float4*>(a.pre_act + idx) = pre;
            *reinterpret_cast<float4*>(a.y + idx) = out;
        } else {  // (a valid length that is not a multiple of 4: no V1 layer has one — hop = 256 — but a bad count must not leave the slot's rules)
            const float pv[4] = {pre.x, pre.y, pre.z, pre.w}, ov[4] = {out.x, out.y, out.z, out.w};
            for (int e = 0; e < 4 && t + e < L; ++e) {
                if (a.pre_act) a.pre_act[idx + e] = pv[e];
                a.y[idx + e] = ov[e];
            }
        }
    }
}

bool conv_post_fast_supported(int Cin, int Cout, int K, int L) { return Cin == 32 && Cout == 1 && K == 7 && L % 4 == 0; }

hipError_t launch_conv_post_fast(const ConvArgs& a, hipStream_t s) {
    long nthr = ((long)a.L / 4 + 255) / 256;
    int gx = (int)(nthr < 4096 ? nthr : 4096);
    hipLaunchKernelGGL((conv_post_k<32, 7>), dim3(gx, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace vtts
