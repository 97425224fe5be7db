// Fused ResBlock1 pair in fp32 on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32):
//     x' = c2(lrelu(c1(lrelu(x)))) + x          (vietTTS/hifigan/model.py:45-50)
// one launch instead of two, `xt = c1(.)` never leaves the CU.
//
// Why (round 4; profiles/r03_g_f32_pmc.md): the fp32 engine — the path that meets BASELINE.json's 1e-4 — is NOT power-limited
// (2.38 GHz at 1.15 kW), so every point of matrix-pipe utilisation is throughput, and its C <= 64 convolutions sit at MfmaUtil
// 0.36-0.71 while moving 2-3 TB/s: per convolution one full read of the input and one full write of the output (plus the
// residual's read) for 2*C*k FLOP per element.  Fused, a pair reads x once (+ halo), writes x' once and keeps xt in LDS:
// 2 tensor passes + the residual instead of 5.
//
// Arithmetic: the SAME fmaf chains in the same order as the two separate launches (conv1d_f32_mfma_k: chunk, tap, channel pair;
// bias added to the finished sum; LeakyReLU of the stored fp32 value; the reference's zero padding of xt outside [0, L)), so a
// fused pair is BIT-IDENTICAL to the unfused path (tests/test_gpu_parity.py::test_fp32_fused_pairs_are_bit_identical).
//
// Geometry (one workgroup = 4 waves = all C output channels x N1 columns of c1, of which c2 keeps NT2 = N1 - (KS - 1)):
//   X tile   [CK][rsx]  fp32, channel-major as in HBM (rows contiguous along time: float4 loads, LeakyReLU once while staging),
//            columns = times t0 - H2 - H2*dil ... rounded down to a multiple of 4 for the float4 loads (`off` = the remainder);
//            C = 128 stages 64 input channels at a time;
//   xt tile  [C][RST]   written by epilogue 1 over the dead X tile (two workgroup barriers), RST = N1 + 24: the 2*H2 columns past N1
//            that only discarded output columns read are zero-filled, and RST = 8 (mod 16) keeps the accumulator-layout writes
//            (a half-wave = 32 consecutive columns of one channel, the halves 4 channels apart) on distinct banks;
//   B fragment of v_mfma_f32_32x32x2_f32 = one ds_read_b32 per lane (32 consecutive columns of channel k, the other half-wave k + 1);
//   A fragments (weights) straight from L2 in the packed order of conv1d_f32_mfma_k (one 16-byte load = 4 k-steps), one iteration ahead.
#include <type_traits>

#include "device_common.h"

namespace vtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FP_MAX_DIL = 5;
constexpr int FP_XCD_MIN_TILES = 64;
constexpr int fp_round_up4(int v) { return (v + 3) & ~3; }

struct PairArgsF32 {
    ConvArgs a;          // c1's view: x, wp (c1 packed), bias (b1), dil; and the OUTPUT side: y, res (= x), acc_mode, div, B, L, zrev
    const void* wp2;     // c2's packed weights (conv1d_f32_mfma_pack order)
    const float* bias2;
};

template <int C_, int KS_, int N1_, int WM_, int WN_, int CK_, int WPS_>
struct F32PairTile {
    static constexpr int C = C_, KS = KS_, N1 = N1_, WM = WM_, WN = WN_, CK = CK_;
    static constexpr int WPS = WPS_;                   // waves per SIMD the register budget is set for (= workgroups per CU the LDS tile allows)
    static constexpr int H2 = (KS - 1) / 2;
    static constexpr int NT2 = N1 - 2 * H2;            // outputs per workgroup
    static constexpr int MR = C / WM / 32, NR = N1 / WN / 32;
    static constexpr int NCH = C / CK, CQ = CK / 8, NIT = KS * CQ;
    static constexpr int RST = N1 + 24;                // xt row stride (floats)
    static constexpr int W4MAX = (N1 + 2 * H2 * FP_MAX_DIL + 3 + 3) / 4;
    static constexpr int ITER = (CK * W4MAX + 255) / 256;  // float4 units per thread and chunk at the largest rate
    static __host__ __device__ constexpr int w4_of(int dil) { return (N1 + 2 * H2 * dil + 3 + 3) / 4; }
    static __host__ __device__ constexpr int lds_floats(int dil) {
        const int x = CK * 4 * w4_of(dil), t = C * RST;
        return x > t ? x : t;
    }
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(C % (WM * 32) == 0 && N1 % (WN * 32) == 0 && C % CK == 0 && CK % 8 == 0, "tiling");
    static_assert(2 * H2 <= 24 && RST % 16 == 8, "xt tail columns / bank spread");
};

template <class T>
__global__ __launch_bounds__(256, T::WPS) void resblock_pair_f32_k(PairArgsF32 p) {
    constexpr int C = T::C, KS = T::KS, N1 = T::N1, WN = T::WN, CK = T::CK, H2 = T::H2, NT2 = T::NT2;
    constexpr int MR = T::MR, NR = T::NR, NCH = T::NCH, CQ = T::CQ, NIT = T::NIT, RST = T::RST, ITER = T::ITER;
    const ConvArgs& a = p.a;

    extern __shared__ __attribute__((aligned(16))) float xs[];  // X tile [CK][rsx], later the xt tile [C][RST]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int LP = a.L;              // row pitch of x / y
    const int L = valid_len(a, b);   // this utterance's columns (ragged batches; == LP otherwise): zero padding and store masks follow it
    // XCD-aware tile order (as conv1d_f32_mfma_k): XCD blockIdx.x % 8 takes a contiguous, balanced eighth of the utterance's time tiles
    int tile = blockIdx.x;
    if (gridDim.x >= FP_XCD_MIN_TILES) {
        const int nt = (L + NT2 - 1) / NT2, r = (int)((blockIdx.x + blockIdx.z) & 7), lo = (r * nt) >> 3, hi = ((r + 1) * nt) >> 3;
        tile = lo + (int)(blockIdx.x >> 3);
        if (tile >= hi) return;
    }
    const int t0 = tile * NT2;  // first output time of this workgroup
    if (t0 >= L) return;
    const int dil = a.dil;
    const int h1 = H2 * dil;                     // c1's symmetric pad (model.py:8-10)
    const int tstart = t0 - H2 - h1;             // time of X column `off`: c1's output column n, tap j reads time tstart + n + j*dil
    const int off = ((tstart % 4) + 4) % 4;
    const int tx0 = tstart - off;                // multiple of 4 (t0 = tile*NT2 and the pads are arbitrary; L % 4 == 0)
    const int w4 = T::w4_of(dil);                // float4 units per staged row
    const int rsx = 4 * w4;                      // X row stride (floats)
    const unsigned w4_magic = 0xFFFFFFFFu / (unsigned)w4 + 1u;  // idx / w4 == umulhi(idx, magic) for idx < 2^16
    const float* __restrict__ xb = a.x + (long)b * a.x_sb;
    const int m0 = wm * (C / T::WM);             // first output channel of this wave

    const float4* wbase1[MR];
    const float4* wbase2[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const long o = (long)(m0 / 32 + mr) * (NCH * NIT) * 64 + lane;
        wbase1[mr] = reinterpret_cast<const float4*>(a.wp) + o;
        wbase2[mr] = reinterpret_cast<const float4*>(p.wp2) + o;
    }

    f32x16 acc[MR][NR];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;
    };
    zero_acc();

    // A fragments: a ring of PF iterations (4 k-steps each) in flight per m-block, so a fragment is requested PF * 4 * MR * NR MFMAs (>= 2048
    // matrix-pipe cycles) before its use — the fused kernel runs 2-4 workgroups per CU, too few waves to hide an L2 round trip behind other
    // waves' MFMAs the way the 5-6 workgroups of the single-convolution kernel do.  The ring runs across the channel chunks and from c1
    // straight into c2 (whose first fragments are in flight under epilogue 1).
    constexpr int PF = 4;
    static_assert(NIT % PF == 0, "ring slots are compile-time positions in the block loop");
    constexpr long TOTAL = (long)NCH * NIT;
    float4 a_ring[PF][MR];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) a_ring[u][mr] = wbase1[mr][u * 64];

    // one convolution pass over the LDS tile (PHASE 1: c1 over the X tile, PHASE 2: c2 over the xt tile): `rs` = the tile's row stride,
    // `col0` = the column tap 0 of this lane's output column 0 reads, `dl` = the rate, `chunk` = which CK input channels of the packed A
    // stream, `rowoff` = first tile row of that chunk.  B fragments (one ds_read_b32 per lane and 32-column block) are read one k-step ahead.
    auto conv_chunk = [&](auto phase_tag, int chunk, int rowoff, int rs, int col0, int dl) {
        constexpr int PHASE = decltype(phase_tag)::value;
        auto bload = [&](int it, int e, float (&dst)[NR]) {
            const int j = it / CQ;
            const int cq = it - j * CQ;
            const float* xr = &xs[(rowoff + cq * 8 + 2 * e + lh) * rs + col0 + j * dl];
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
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int par = (u * 4 + e) & 1;
                    // the next k-step's B fragment (the pass's last step reads nothing)
                    if (e < 3) bload(it, e + 1, bf[par ^ 1]);
                    else if (it + 1 < NIT) bload(it + 1, 0, bf[par ^ 1]);
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) {
                        const float av = e == 0 ? a_ring[u][mr].x : e == 1 ? a_ring[u][mr].y : e == 2 ? a_ring[u][mr].z : a_ring[u][mr].w;
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[par][nr], acc[mr][nr], 0, 0, 0);
                    }
                }
                // refill this slot with the fragment PF iterations ahead: the same pass, or (from c1's tail) c2's first ones
                const long q = (long)chunk * NIT + it + PF;
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    if constexpr (PHASE == 1) a_ring[u][mr] = q < TOTAL ? wbase1[mr][q * 64] : wbase2[mr][(q - TOTAL) * 64];
                    else a_ring[u][mr] = wbase2[mr][(q < TOTAL ? q : TOTAL - 1) * 64];  // the tail re-reads the last fragment (in bounds, never used)
                }
            }
        }
    };

    // ---------------- phase 1: xt = c1(lrelu(x)); column n <-> time t0 - H2 + n ----------------
    const int colbase1 = off + wn * (N1 / WN) + l31;
    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk) __syncthreads();  // all waves done reading the previous chunk
        {
            // stage CK channels x rsx columns: all of a thread's loads first (clamped addresses, masked afterwards), LeakyReLU, float4 LDS writes
            const float* __restrict__ xc = xb + (long)(chunk * CK) * a.x_sc;
            const int total = CK * w4;
            float4 v[ITER];
            int dst[ITER];
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                const int idx = tid + i * 256;
                const int idc = idx < total ? idx : total - 1;
                const int row = (int)__umulhi((unsigned)idc, w4_magic);
                const int c4 = idc - row * w4;
                const int t = tx0 + 4 * c4;
                const bool ok = idx < total && t >= 0 && t < L;
                const int tc = t < 0 ? 0 : (t >= LP ? LP - 4 : t);
                v[i] = mask_tail4(*reinterpret_cast<const float4*>(xc + (long)row * a.x_sc + tc), t, L);
                if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                dst[i] = idx < total ? row * rsx + 4 * c4 : -1;
            }
#pragma unroll
            for (int i = 0; i < ITER; ++i) {
                if (dst[i] < 0) continue;
                float4 q = v[i];
                q.x = lrelu(q.x, a.slope_in);
                q.y = lrelu(q.y, a.slope_in);
                q.z = lrelu(q.z, a.slope_in);
                q.w = lrelu(q.w, a.slope_in);
                *reinterpret_cast<float4*>(&xs[dst[i]]) = q;
            }
        }
        __syncthreads();
        conv_chunk(std::integral_constant<int, 1>{}, chunk, 0, rsx, colbase1, dil);
    }
    __syncthreads();  // every wave is done reading the X tile

    // ---------------- epilogue 1: + b1, LeakyReLU(0.1), zero outside [0, L) -> xt tile ----------------
    // C/D layout: column = lane & 31 (time), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float bv = a.bias[co];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int n = wn * (N1 / WN) + nr * 32 + l31;
                const int tt = t0 - H2 + n;
                float v = lrelu(acc[mr][nr][r] + bv, a.slope_in);  // LRELU_SLOPE for both activations of a pair (model.py:46,48)
                if (tt < 0 || tt >= L) v = 0.0f;                   // c2's own zero padding applies to xt
                xs[co * RST + n] = v;
            }
        }
    for (int u = tid; u < C * 2 * H2; u += 256) {  // columns N1 .. N1 + 2*H2 - 1 are only read by the discarded output columns: keep them finite
        const int row = u / (2 * H2), c = u - row * (2 * H2);
        xs[row * RST + N1 + c] = 0.0f;
    }
    zero_acc();
    __syncthreads();  // xt tile written

    // ---------------- phase 2: c2 over the xt tile (rate 1): column n <-> time t0 + n, tap j reads xt column n + j ----------------
    const int colbase2 = wn * (N1 / WN) + l31;
#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) conv_chunk(std::integral_constant<int, 2>{}, chunk, chunk * CK, RST, colbase2, 1);

    // ---------------- epilogue 2: + b2, + x (residual), MRF accumulate / mean ----------------
    // The same operations in the same order as device_common.h: epilogue_store — v = acc + b2; v = v + x; ACC_ADD: v = y + v; ACC_MEAN:
    // v = (y + v) / div — but every residual (and accumulator) value of a block is REQUESTED before the first store: a.res and a.y may alias
    // as far as the compiler knows, so the generic per-element form is a chain of 16 dependent load -> add -> store round trips per block.
    const int mode = a.acc_mode;
    const float dv = a.div;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int n = wn * (N1 / WN) + nr * 32 + l31;
            const int t = t0 + n;
            const bool ok = n < NT2 && t < L;
            const int tc = ok ? t : 0;  // a masked lane reads an in-bounds address of its own row and stores nothing
            constexpr int EB = T::WPS >= 4 ? 8 : 16;  // values requested per round trip (the 128-register budget of 4 waves per SIMD holds 8)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += EB) {
                float rv[EB], yv[EB];
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const int r = r0 + q;
                    const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const long idx = ((long)b * C + co) * LP + tc;
                    rv[q] = a.res[idx];
                    yv[q] = mode != ACC_STORE ? a.y[idx] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const int r = r0 + q;
                    const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const long idx = ((long)b * C + co) * LP + tc;
                    float v = acc[mr][nr][r] + p.bias2[co];
                    v = v + rv[q];
                    if (mode == ACC_ADD) v = yv[q] + v;
                    else if (mode == ACC_MEAN) v = (yv[q] + v) / dv;
                    if (ok) a.y[idx] = v;
                }
            }
        }
}

// ---- tile table ------------------------------------------------------------------------------------
//                                                 C   KS   N1  WM WN CK WPS
#ifndef VTTS_FP32_N1  // tile-geometry experiments (A/B builds): columns per workgroup and the waves per SIMD the registers are budgeted for
#define VTTS_FP32_N1 256
#define VTTS_FP32_WPS 4
#endif
#ifndef VTTS_FP64_N1  // 256 columns (80 KB of LDS, two workgroups per CU): 64 x 1024 frames 348.3 -> 346.7 ms against 128 columns x 3 workgroups;
#define VTTS_FP64_N1 256  // 64 columns x 5 workgroups 359.3 ms; C = 32: 512 columns 350.4, 128 columns 356.7 (gpurun_out/r04_run3/f32_ab.log)
#define VTTS_FP64_WPS 2
#endif
template <int KS> using FP32 = F32PairTile<32, KS, VTTS_FP32_N1, 1, 4, 32, VTTS_FP32_WPS>;
template <int KS> using FP64 = F32PairTile<64, KS, VTTS_FP64_N1, 2, 2, 64, VTTS_FP64_WPS>;
template <int KS> using FP128 = F32PairTile<128, KS, 128, 2, 2, 64, 2>;

template <class T>
static hipError_t launch_fp(const PairArgsF32& p, hipStream_t s) {
    static DynLdsOnce once;
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&resblock_pair_f32_k<T>), T::lds_floats(FP_MAX_DIL) * 4, once); e != hipSuccess) return e;
    dim3 grid((p.a.L + T::NT2 - 1) / T::NT2, 1, p.a.B);
    if ((int)grid.x >= FP_XCD_MIN_TILES) grid.x = (grid.x + 7) / 8 * 8;
    hipLaunchKernelGGL(resblock_pair_f32_k<T>, grid, dim3(256), T::lds_floats(p.a.dil) * 4, s, p);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_fp_ks(const PairArgsF32& p, int K, hipStream_t s) {
    switch (K) {
        case 3: return launch_fp<TT<3>>(p, s);
        case 7: return launch_fp<TT<7>>(p, s);
        case 11: return launch_fp<TT<11>>(p, s);
    }
    return hipErrorInvalidValue;
}

bool pair_f32_supported(int C, int K, int dil, int L) {
    return (C == 128 || C == 64 || C == 32) && (K == 3 || K == 7 || K == 11) && dil >= 1 && dil <= FP_MAX_DIL && L % 4 == 0 && L >= 4;
}

// x [B][C][L] -> y [B][C][L]; a = c1's ConvArgs with the output side (y, res = x, acc_mode, div) filled in
hipError_t launch_pair_f32(const ConvArgs& a, const void* wp2, const float* bias2, hipStream_t s) {
    PairArgsF32 p{a, wp2, bias2};
    switch (a.Cin) {
        case 128: return launch_fp_ks<FP128>(p, a.K, s);
        case 64: return launch_fp_ks<FP64>(p, a.K, s);
        case 32: return launch_fp_ks<FP32>(p, a.K, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace vtts
