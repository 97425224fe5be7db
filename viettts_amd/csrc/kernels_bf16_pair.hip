// Fused ResBlock1 pair in bf16:   x' = c2(lrelu(c1(lrelu(x)))) + x      (vietTTS/hifigan/model.py:45-50)
//
// The un-fused bf16 path moves five tensors through HBM per pair (c1: read x, write xt; c2: read xt,
// read x as residual, write x') and is bound by those HBM phases, not by the MFMAs (ablation in
// DESIGN.md).  Here xt never leaves the CU: one workgroup computes, for NT2 output time steps,
//   phase 1   xt[r][:] for the NT2 + 2*H2 rows c2 needs (H2 = (K-1)/2; c2 has rate 1), from an X tile of
//             N1 + 2*H1 rows (H1 = (K-1)/2 * rate) staged once with LeakyReLU and zero padding applied;
//             epilogue 1 = bias, LeakyReLU, bf16 rounding, ZERO outside [0, L) (c2's own zero padding
//             applies to xt), written straight into LDS in the swizzled channels-last tile layout;
//   phase 2   c2 over that LDS tile; epilogue 2 = bias + raw x rows (re-read through L2) [+ MRF
//             accumulate / mean / consumer's LeakyReLU], via the fp32 LDS transpose of kernels_bf16.hip.
// HBM traffic per pair: x once (+ halo), x' once.  Both convolutions share one stream of LDS-DMA'd
// weight slabs (c1's slabs, then c2's), double-buffered across the phase boundary.
//
// Tiling: N1 = 32*NR*WN MFMA columns per phase; phase 2 computes N1 columns too and discards the last
// 2*H2 (so NT2 = N1 - 2*H2 outputs per workgroup: 4 % waste at N1 = 256, K = 11).
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

// timing ablations (never defined in the product build): results are wrong when any is set
#ifndef VTTS_ABL_NOB
#define VTTS_ABL_NOB 0
#endif
#ifndef VTTS_ABL_NOA
#define VTTS_ABL_NOA 0
#endif
#ifndef VTTS_ABL_NODMA
#define VTTS_ABL_NODMA 0
#endif
#ifndef VTTS_ABL_NOBAR
#define VTTS_ABL_NOBAR 0
#endif
#ifndef VTTS_ABL_NOMFMA
#define VTTS_ABL_NOMFMA 0
#endif
// experiment: stream the weight slabs global -> VGPR -> ds_write instead of LDS-DMA
#ifndef VTTS_SLAB_VIA_REGS
#define VTTS_SLAB_VIA_REGS 0
#endif

namespace vtts {

template <int C_, int XC_, int CKC_, int KS_, int N1_, int WM_, int WN_, int TG_>
struct PTile {
    static constexpr int C = C_, XC = XC_, CKC = CKC_, KS = KS_, N1 = N1_, WM = WM_, WN = WN_, TG = TG_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int MR = C / WM / 32, NR = N1 / WN / 32;
    static constexpr int H2 = (KS - 1) / 2;             // c2 halo (rate 1)
    static constexpr int PA1 = (KS - 1) / 2 * 5;        // staged c1 halo per side (rate <= 5)
    static constexpr int NT2 = N1 - 2 * H2;             // outputs per workgroup
    static constexpr int NXC = C / XC, NCK1 = XC / CKC, NCK2 = C / CKC;
    static constexpr int SPR1 = XC / 8, P1 = XC * 2;    // X tile: slots / bytes per row
    static constexpr int SPR2 = C / 8, P2 = C * 2;      // xt tile
    static constexpr int ROWSX = N1 + 2 * PA1;
    static constexpr int ROWST = N1 + 2 * H2;           // xt rows incl. the tail only discarded columns read
    static constexpr int X_BYTES = ROWSX * P1, XT_BYTES = ROWST * P2;
    static constexpr int REGION_A = X_BYTES > XT_BYTES ? X_BYTES : XT_BYTES;  // xt overwrites the dead X tile
    static constexpr int KSTEPS = CKC / 16;
    static constexpr int NSL = (KS + TG - 1) / TG;
    static constexpr int NS1 = NXC * NCK1 * NSL, NS2 = NCK2 * NSL, NSTOT = NS1 + NS2;
    static constexpr int MB = C / 32;
    static constexpr int SLAB_BYTES = C * TG * CKC * 2;
    static constexpr int SLAB_UNITS = SLAB_BYTES / 16;
    static constexpr int APT = (SLAB_UNITS + THREADS - 1) / THREADS;
    static constexpr int XPT = (ROWSX * SPR1 + THREADS - 1) / THREADS;
    static constexpr int EP_PITCH = C * 4 + 16;
    static constexpr int EP_BYTES = N1 * EP_PITCH;
    static constexpr int MAIN_BYTES = REGION_A + 2 * SLAB_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES > EP_BYTES ? MAIN_BYTES : EP_BYTES;
    static_assert(C % (WM * 32) == 0 && N1 % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(C % XC == 0 && XC % CKC == 0 && CKC % 16 == 0, "channel tiling");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(SLAB_UNITS % 64 == 0, "slab = whole wave-instructions of LDS-DMA");
};

template <class T>
__global__ __launch_bounds__(T::THREADS) void resblock_pair_bf16_k(BConvArgs a) {
    constexpr int C = T::C, XC = T::XC, CKC = T::CKC, KS = T::KS, N1 = T::N1, WN = T::WN, TG = T::TG;
    constexpr int THREADS = T::THREADS, MR = T::MR, NR = T::NR, H2 = T::H2, PA1 = T::PA1, NT2 = T::NT2;
    constexpr int NXC = T::NXC, NCK1 = T::NCK1, NCK2 = T::NCK2, SPR1 = T::SPR1, P1 = T::P1, SPR2 = T::SPR2, P2 = T::P2;
    constexpr int ROWSX = T::ROWSX, KSTEPS = T::KSTEPS, NSL = T::NSL, NSTOT = T::NSTOT, MB = T::MB, APT = T::APT, XPT = T::XPT;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xt = lds;                 // X tile, later the xt tile
    unsigned char* ab = lds + T::REGION_A;   // two weight-slab buffers

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
    const int h1 = a.pad;                    // c1's symmetric pad = (K-1)/2 * rate

    const uint4* __restrict__ wsl = reinterpret_cast<const uint4*>(a.wp);
    [[maybe_unused]] const int wg_lin = blockIdx.z * gridDim.x + blockIdx.x;
    VTTS_TL(a, wg_lin, 0);
    VTTS_TL_ID(a, wg_lin);

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

    auto issue_slab = [&](int s, int buf) {
        const uint4* src = wsl + (size_t)s * T::SLAB_UNITS;
        unsigned char* dst = ab + buf * T::SLAB_BYTES + (size_t)(wave * 64) * 16;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int u0 = wave * 64 + i * THREADS;
            if (!VTTS_ABL_NODMA && (APT * THREADS == T::SLAB_UNITS || u0 < T::SLAB_UNITS))
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + u0 + lane), (lds_ptr_t)(dst + (size_t)i * THREADS * 16), 16, 0, 0);
        }
    };
    uint4 areg[APT];
    auto load_slab_regs = [&](int s) {
        const uint4* src = wsl + (size_t)s * T::SLAB_UNITS;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int u = tid + i * THREADS;
            if (APT * THREADS == T::SLAB_UNITS || u < T::SLAB_UNITS) areg[i] = src[u];
        }
        __builtin_amdgcn_sched_barrier(0);  // the loads stay here, a whole slab of MFMAs ahead of their use
    };
    auto write_slab_regs = [&](int buf) {
        uint4* dst = reinterpret_cast<uint4*>(ab + buf * T::SLAB_BYTES);
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int u = tid + i * THREADS;
            if (APT * THREADS == T::SLAB_UNITS || u < T::SLAB_UNITS) dst[u] = areg[i];
        }
    };
    // X rows [t0 - H2 - PA1, +ROWSX): LeakyReLU + zero padding in registers, swizzled ds_write_b128
    auto stage_x = [&](int xc) {
        uint4 v[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * THREADS;
            const int row = u / SPR1, c = u % SPR1;
            const int t = t0 - H2 - PA1 + row;
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (u < ROWSX * SPR1 && t >= 0 && t < L)
                v[i] = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(a.x) + ((size_t)b * L + t) * C + xc * XC + c * 8);
        }
        const float s_in = a.slope_in;
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            v[i].x = lrelu_bf16x2(v[i].x, s_in);
            v[i].y = lrelu_bf16x2(v[i].y, s_in);
            v[i].z = lrelu_bf16x2(v[i].z, s_in);
            v[i].w = lrelu_bf16x2(v[i].w, s_in);
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * THREADS;
            const int row = u / SPR1, c = u % SPR1;
            if (u < ROWSX * SPR1) *reinterpret_cast<uint4*>(xt + row * P1 + ((c ^ swz_of<SPR1>(row)) << 4)) = v[i];
        }
    };

    // One weight slab's worth of MFMAs.  B fragments come from tile `tile` (row pitch PB, SPRB slots per
    // row); output column n of this wave reads tile row  n + (tap index)*dl + row_shift.
    auto mma_slab = [&](const unsigned char* abuf_, int ntaps, int tap0, int slot0, int dl, int row_shift, auto spr_tag) {
        constexpr int SPRB = decltype(spr_tag)::value;
        constexpr int PB = SPRB * 16;
        const unsigned char* abuf = abuf_ + (size_t)(wm * MR) * 1024 + lane * 16;
        const int rowbase0 = wn * (N1 / WN) + l31 + row_shift;
        int rowoff[NR], rowswz[NR];
        auto set_rows = [&](int tj) {
            const int rowb = rowbase0 + (tap0 + tj) * dl;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int row = rowb + nr * 32;
                rowoff[nr] = row * PB;
                rowswz[nr] = swz_of<SPRB>(row);
            }
        };
        bf16x8 af[MR], bf[NR], afn[MR], bfn[NR];
        set_rows(0);
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) bf[nr] = *reinterpret_cast<const bf16x8*>(xt + rowoff[nr] + ((slot0 ^ rowswz[nr]) << 4));
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) af[mr] = *reinterpret_cast<const bf16x8*>(abuf + mr * 1024);
        for (int tj = 0; tj < ntaps; ++tj) {
            const unsigned char* aslab = abuf + (size_t)(tj * KSTEPS) * MB * 1024;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks + 1 < KSTEPS) {
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        if (!VTTS_ABL_NOB) bfn[nr] = *reinterpret_cast<const bf16x8*>(xt + rowoff[nr] + (((slot0 + (ks + 1) * 2) ^ rowswz[nr]) << 4));
                        else bfn[nr] = bf[nr];
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
                        if (!VTTS_ABL_NOA) afn[mr] = *reinterpret_cast<const bf16x8*>(aslab + ((ks + 1) * MB + mr) * 1024);
                        else afn[mr] = af[mr];
                } else if (tj + 1 < ntaps) {
                    set_rows(tj + 1);
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        if (!VTTS_ABL_NOB) bfn[nr] = *reinterpret_cast<const bf16x8*>(xt + rowoff[nr] + ((slot0 ^ rowswz[nr]) << 4));
                        else bfn[nr] = bf[nr];
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr)
                        if (!VTTS_ABL_NOA) afn[mr] = *reinterpret_cast<const bf16x8*>(aslab + (KSTEPS * MB + mr) * 1024);
                        else afn[mr] = af[mr];
                }
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        if (!VTTS_ABL_NOMFMA) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mr], bf[nr], acc[mr][nr], 0, 0, 0);
                        else acc[mr][nr][0] += (float)af[mr][0] * (float)bf[nr][1];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) af[mr] = afn[mr];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bf[nr] = bfn[nr];
            }
        }
    };

    if (VTTS_SLAB_VIA_REGS) {
        load_slab_regs(0);
        stage_x(0);
        write_slab_regs(0);
    } else {
        issue_slab(0, 0);
        stage_x(0);
    }
    __syncthreads();
    VTTS_TL(a, wg_lin, 1);

    // ---------------- phase 1: xt = c1(lrelu(x)) over N1 rows starting at time t0 - H2 ----------------
    int s = 0;
    for (int xc = 0; xc < NXC; ++xc) {
        if (xc > 0) {
            stage_x(xc);
            __syncthreads();
        }
        for (int ck = 0; ck < NCK1; ++ck) {
            for (int sl = 0; sl < NSL; ++sl, ++s) {
                if ((s + 1) < NSTOT) {
                    if (VTTS_SLAB_VIA_REGS) load_slab_regs(s + 1);
                    else issue_slab(s + 1, (s + 1) & 1);
                }
                const int ntaps = (KS - sl * TG) < TG ? (KS - sl * TG) : TG;
                // column n <-> xt time t0 - H2 + n; tap j reads x time t0 - H2 + n + j*dil - h1 = X row n + j*dil + (PA1 - h1)
                mma_slab(ab + (s & 1) * T::SLAB_BYTES, ntaps, sl * TG, ck * (CKC / 8) + lh, dil, PA1 - h1,
                         std::integral_constant<int, SPR1>{});
                if (VTTS_SLAB_VIA_REGS && (s + 1) < NSTOT) write_slab_regs((s + 1) & 1);
                if (!VTTS_ABL_NOBAR) __syncthreads();
            }
        }
    }

    VTTS_TL(a, wg_lin, 2);
    // ---------------- epilogue 1: bias, LeakyReLU(0.1), bf16, zero outside [0, L) -> xt tile in LDS ----------------
    // (all waves are past the barrier that ended the last c1 slab: the X tile is dead)
    {
        const float* __restrict__ bias1 = a.bias;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int co = wm * (C / T::WM) + mr * 32 + 8 * rq + 4 * lh;
                const float4 bv = *reinterpret_cast<const float4*>(bias1 + co);
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int row = wn * (N1 / WN) + nr * 32 + l31;
                    const int tt = t0 - H2 + row;
                    uint2 pk = make_uint2(0u, 0u);
                    if (tt >= 0 && tt < L) {
                        pk.x = pack_bf16x2(lrelu_f(acc[mr][nr][4 * rq + 0] + bv.x, 0.1f), lrelu_f(acc[mr][nr][4 * rq + 1] + bv.y, 0.1f));
                        pk.y = pack_bf16x2(lrelu_f(acc[mr][nr][4 * rq + 2] + bv.z, 0.1f), lrelu_f(acc[mr][nr][4 * rq + 3] + bv.w, 0.1f));
                    }
                    const int slot = co >> 3;  // 16-byte slot of channels co..co+7; this lane owns half of it
                    *reinterpret_cast<uint2*>(xt + row * P2 + ((slot ^ swz_of<SPR2>(row)) << 4) + (co & 4) * 2) = pk;
                }
            }
        }
        // rows N1 .. N1 + 2*H2 - 1 are only read by the discarded output columns: keep them finite
        for (int u = tid; u < 2 * H2 * SPR2; u += THREADS) {
            const int row = N1 + u / SPR2, c = u % SPR2;
            *reinterpret_cast<uint4*>(xt + row * P2 + (c << 4)) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    zero_acc();
    __syncthreads();
    VTTS_TL(a, wg_lin, 3);

    // ---------------- phase 2: c2 over the xt tile (rate 1): column n <-> time t0 + n, tap j reads xt row n + j ----------------
    for (int ck = 0; ck < NCK2; ++ck) {
        for (int sl = 0; sl < NSL; ++sl, ++s) {
            if ((s + 1) < NSTOT) {
                if (VTTS_SLAB_VIA_REGS) load_slab_regs(s + 1);
                else issue_slab(s + 1, (s + 1) & 1);
            }
            const int ntaps = (KS - sl * TG) < TG ? (KS - sl * TG) : TG;
            mma_slab(ab + (s & 1) * T::SLAB_BYTES, ntaps, sl * TG, ck * (CKC / 8) + lh, 1, 0, std::integral_constant<int, SPR2>{});
            if (VTTS_SLAB_VIA_REGS && (s + 1) < NSTOT) write_slab_regs((s + 1) & 1);
            if (!VTTS_ABL_NOBAR) __syncthreads();
        }
    }

    VTTS_TL(a, wg_lin, 4);
    // ---------------- epilogue 2 ----------------
    float* ep = reinterpret_cast<float*>(lds);
    constexpr int EPF = T::EP_PITCH / 4;
    const float* __restrict__ bias2 = a.bias + C;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int col = wm * (C / T::WM) + mr * 32 + 8 * rq + 4 * lh;
            const float4 bv = *reinterpret_cast<const float4*>(bias2 + col);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int row = wn * (N1 / WN) + nr * 32 + l31;
                float4 v;
                v.x = acc[mr][nr][4 * rq + 0] + bv.x;
                v.y = acc[mr][nr][4 * rq + 1] + bv.y;
                v.z = acc[mr][nr][4 * rq + 2] + bv.z;
                v.w = acc[mr][nr][4 * rq + 3] + bv.w;
                *reinterpret_cast<float4*>(ep + row * EPF + col) = v;
            }
        }
    }
    __syncthreads();
    VTTS_TL(a, wg_lin, 5);

    constexpr int UPR = C / 8;
    const float s_out = a.slope_out;
    unsigned short* __restrict__ y = static_cast<unsigned short*>(a.y);
    const unsigned short* __restrict__ xres = static_cast<const unsigned short*>(a.x);  // residual = the pair's own input, raw
    for (int u = tid; u < NT2 * UPR; u += THREADS) {
        const int row = u / UPR, c8 = u % UPR;
        const int t = t0 + row;
        if (t >= L) continue;
        const float4 p0 = *reinterpret_cast<const float4*>(ep + row * EPF + c8 * 8);
        const float4 p1 = *reinterpret_cast<const float4*>(ep + row * EPF + c8 * 8 + 4);
        float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        const size_t g = ((size_t)b * L + t) * C + c8 * 8;
        {  // x = xt + x  (model.py:50)
            const uint4 r = *reinterpret_cast<const uint4*>(xres + g);
            v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
            v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
        }
        if (a.acc_add) {  // MRF  xs += rb(x)  (model.py:118-120)
            const uint4 o = *reinterpret_cast<const uint4*>(y + g);
            v[0] = bf16_lo(o.x) + v[0]; v[1] = bf16_hi(o.x) + v[1]; v[2] = bf16_lo(o.y) + v[2]; v[3] = bf16_hi(o.y) + v[3];
            v[4] = bf16_lo(o.z) + v[4]; v[5] = bf16_hi(o.z) + v[5]; v[6] = bf16_lo(o.w) + v[6]; v[7] = bf16_hi(o.w) + v[7];
        }
        if (a.div != 1.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] / a.div;
        }
        if (s_out != 1.0f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = lrelu_f(v[e], s_out);
        }
        *reinterpret_cast<uint4*>(y + g) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
    VTTS_TL(a, wg_lin, 6);
}

// ---- tile table -------------------------------------------------------------------------------------
//                                      C    XC  CKC  KS  N1  WM WN TG
// (the C = 128 / 64 geometries of this generation — 8 waves, weight slabs double-buffered through LDS with one
// s_barrier per slab — are superseded by kernels_bf16_rbg.hip; C = 32 still runs fastest here)
template <int KS> using PRes32 = PTile<32, 32, 32, KS, 512, 1, 8, KS>;

template <class T>
static hipError_t launch_p(const BConvArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_bf16_k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid((a.L + T::NT2 - 1) / T::NT2, 1, a.B);
    hipLaunchKernelGGL(resblock_pair_bf16_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_p_ks(const BConvArgs& a, int K, hipStream_t s) {
    switch (K) {
        case 3: return launch_p<TT<3>>(a, s);
        case 7: return launch_p<TT<7>>(a, s);
        case 11: return launch_p<TT<11>>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pair_lds_bf16(int C, int K, const BConvArgs& a, hipStream_t s) {
    if (C == 32) return launch_p_ks<PRes32>(a, K, s);
    return hipErrorInvalidValue;
}

// packing geometry of ONE of the pair's two convolutions (all C output rows in one m-tile)
BPackGeom pair_lds_pack_geom(int C, int K) {
    if (C == 32) return BPackGeom{32, 32, 32, K, 32, K};
    return BPackGeom{0, 0, 0, 0, 0, 0};
}

const char* pair_lds_kernel_name(int C, int K) {
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "resblock_pair_bf16_k<PTile<%d, %d, %d, %d,", C, C, C, K);
    return buf;
}

}  // namespace vtts
