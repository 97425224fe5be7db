// bf16-operand / fp32-accumulate convolutions on v_mfma_f32_32x32x16_bf16 (BASELINE configs[2]).
//
// One kernel template carries every GEMM-shaped layer of the generator in the bf16 path:
//   * the 72 ResBlock1 convolutions (vietTTS/hifigan/model.py:21-28, :33-40),
//   * conv_pre (model.py:83; reads the fp32 mel as given and converts while staging),
//   * the 4 transposed convolutions (model.py:88-94): with CHANNELS-LAST activations the polyphase form
//     of ConvTranspose1d(k = 2s) is literally Conv1d(Cin -> s*Cout, k = 3, pad 1) followed by a free
//     reshape [L][s*Cout] -> [s*L][Cout] (unused (phase, frame) taps are zero
//     weights), so no separate kernel and no strided stores.
//
// Layout: activations [B][L][C] bf16 (the reference's NWC).  A workgroup owns MT output channels x NT
// time steps of one utterance:
//   * X tile  : (NT + 2*PA) rows x CKC channels staged global -> VGPR -> LDS once per input-channel
//               chunk, LeakyReLU (if the producer stored raw values) and zero padding applied in
//               registers.  Rows are 16-byte-slot XOR-swizzled so the MFMA B fragment (lane = time row,
//               8 consecutive channels = one ds_read_b128) is bank-conflict-free for every row pitch.
//               The K taps of the convolution are K row-shifted views of this one tile.
//   * A slabs : weights pre-packed on the host in A-fragment order, streamed L2 -> LDS by LDS-DMA
//               (global_load_lds_dwordx4) in slabs of TG taps x CKC channels (<= 32 KiB), double-buffered:
//               slab s+1 is in flight while slab s feeds the MFMAs (one barrier per slab).
//               Weight traffic from L2 is 2 B / (2*NT) FLOP -> NT = 256..512.
//   * epilogue: accumulators + bias go through LDS as an fp32 [NT][MT] tile so that every global access
//               (residual read, MRF accumulator read, output write) is a full-row coalesced 16-byte
//               access; residual add, MRF sum / mean, the consumer's LeakyReLU and the bf16 rounding
//               happen there, in fp32, in the reference's order of operations.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdio.h>
#include <string.h>

#include "bf16_common.h"

namespace vtts {

template <int CINP_, int XC_, int CKC_, int COUTP_, int KS_, int MT_, int NT_, int WM_, int WN_, int TG_, int PA_, bool IN_F32_>
struct BTile {
    static constexpr int CINP = CINP_, XC = XC_, CKC = CKC_, COUTP = COUTP_, KS = KS_, MT = MT_, NT = NT_, WM = WM_, WN = WN_;
    static constexpr int TG = TG_, PA = PA_;
    static constexpr bool IN_F32 = IN_F32_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int MR = MT / WM / 32, NR = NT / WN / 32;
    static constexpr int NXC = CINP / XC;               // X-tile chunks (re-staged per chunk)
    static constexpr int NCK = XC / CKC;                // weight-slab channel chunks per X chunk
    static constexpr int SPR = XC / 8;                  // 16-byte slots per X row
    static constexpr int P = XC * 2;                    // X row pitch in bytes
    static constexpr int RPB = 16 / SPR;                // X rows per 256-byte LDS bank row
    static constexpr int ROWS = NT + 2 * PA;
    static constexpr int X_BYTES = ROWS * P;
    static constexpr int KSTEPS = CKC / 16;             // MFMA k-steps per tap per slab chunk
    static constexpr int NSL = (KS + TG - 1) / TG;      // slabs per channel chunk
    static constexpr int NSTOT = NXC * NCK * NSL;       // slabs per workgroup
    static constexpr int MB = MT / 32;                  // m-blocks per tile
    static constexpr int SLAB_BYTES = MT * TG * CKC * 2;
    static constexpr int SLAB_UNITS = SLAB_BYTES / 16;
    static constexpr int APT = (SLAB_UNITS + THREADS - 1) / THREADS;  // 16-byte slab units per thread
    static constexpr int NBUF = (NSTOT > 1) ? 2 : 1;
    static constexpr int XPT = (ROWS * SPR + THREADS - 1) / THREADS;  // X units per thread
    static constexpr int EP_PITCH = MT * 4 + 16;        // fp32 epilogue tile row pitch (bytes), conflict-free
    static constexpr int EP_BYTES = NT * EP_PITCH;
    static constexpr int MAIN_BYTES = X_BYTES + NBUF * SLAB_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES > EP_BYTES ? MAIN_BYTES : EP_BYTES;
    static_assert(MT % (WM * 32) == 0 && NT % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(CINP % XC == 0 && XC % CKC == 0 && CKC % 16 == 0 && COUTP % MT == 0, "channel tiling");
    static_assert(SPR == 4 || SPR == 8 || SPR == 16, "row pitch 64/128/256 B");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(SLAB_UNITS % 64 == 0, "slab = whole wave-instructions of LDS-DMA");
};

template <class T>
__global__ __launch_bounds__(T::THREADS) void conv_bf16_k(BConvArgs a) {
    constexpr int XC = T::XC, CKC = T::CKC, COUTP = T::COUTP, KS = T::KS, MT = T::MT, NT = T::NT, WN = T::WN, TG = T::TG, PA = T::PA;
    constexpr int THREADS = T::THREADS, MR = T::MR, NR = T::NR, NXC = T::NXC, NCK = T::NCK, SPR = T::SPR, P = T::P, RPB = T::RPB;
    constexpr int ROWS = T::ROWS, KSTEPS = T::KSTEPS, NSL = T::NSL, NSTOT = T::NSTOT, MB = T::MB, APT = T::APT, XPT = T::XPT;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xt = lds;
    unsigned char* ab = lds + T::X_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;
    const int t0 = blockIdx.x * NT;
    const int mtile = blockIdx.y;
    const int b = blockIdx.z;
    const int Lp = a.L;                                      // rows allocated per utterance
    const int L = a.lens ? min(max(a.lens[b], 0) * a.len_mul, a.L) : a.L;  // valid rows of this utterance, clamped to its slot (ragged batch: the rest reads as zero padding)
    if (t0 >= L) return;  // a tile past this utterance's end

    const uint4* __restrict__ wsl = reinterpret_cast<const uint4*>(a.wp) + (size_t)mtile * NSTOT * T::SLAB_UNITS;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;

    // A slab s -> LDS buffer `buf`, asynchronously (LDS-DMA: global_load_lds_dwordx4, no VGPR round trip).
    // Destination = wave-uniform base + lane*16, which is exactly the fragment-ordered slab image.
    auto issue_slab = [&](int s, int buf) {
        const uint4* src = wsl + (size_t)s * T::SLAB_UNITS;
        unsigned char* dst = ab + buf * T::SLAB_BYTES + (size_t)(wave * 64) * 16;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int u0 = wave * 64 + i * THREADS;  // wave-uniform
            if (APT * THREADS == T::SLAB_UNITS || u0 < T::SLAB_UNITS)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + u0 + lane), (lds_ptr_t)(dst + (size_t)i * THREADS * 16), 16, 0, 0);
        }
    };
    auto stage_x = [&](int cc) {
        uint4 v[XPT];
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * THREADS;
            const int row = u / SPR, c = u % SPR;
            const int t = t0 - PA + row;
            v[i] = make_uint4(0u, 0u, 0u, 0u);
            if (u < ROWS * SPR && t >= 0 && t < L) {
                const int ch = cc * XC + c * 8;
                if constexpr (T::IN_F32) {
                    if (ch < a.cin_real) {
                        const float4* src = reinterpret_cast<const float4*>(static_cast<const float*>(a.x) + ((size_t)b * Lp + t) * a.x_pitch + ch);
                        const float4 f0 = src[0], f1 = src[1];
                        v[i] = make_uint4(pack_bf16x2(f0.x, f0.y), pack_bf16x2(f0.z, f0.w), pack_bf16x2(f1.x, f1.y), pack_bf16x2(f1.z, f1.w));
                    }
                } else {
                    v[i] = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(a.x) + ((size_t)b * Lp + t) * a.x_pitch + ch);
                }
            }
        }
        const float s_in = a.slope_in;
        if (!T::IN_F32 && s_in != 1.0f) {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                v[i].x = lrelu_bf16x2(v[i].x, s_in);
                v[i].y = lrelu_bf16x2(v[i].y, s_in);
                v[i].z = lrelu_bf16x2(v[i].z, s_in);
                v[i].w = lrelu_bf16x2(v[i].w, s_in);
            }
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int u = tid + i * THREADS;
            const int row = u / SPR, c = u % SPR;
            if (u < ROWS * SPR) *reinterpret_cast<uint4*>(xt + row * P + ((c ^ ((row / RPB) & (SPR - 1))) << 4)) = v[i];
        }
    };

    // A transposed convolution run as Conv1d(k = 3) has one structurally zero tap per output phase (engine.hip: phases
    // r < s/2 read frames q-1, q = taps 0, 1; the others q, q+1 = taps 1, 2): an M tile inside one half of the rows skips
    // that tap's slabs altogether (a third of its MFMAs and weight traffic).
    const bool halves = a.convt_halves != 0 && KS == 3 && TG == 1 && (COUTP / 2) % MT == 0;
    const int tap0 = (halves && mtile * MT >= COUTP / 2) ? 1 : 0;
    const int nsl = halves ? 2 : NSL;            // slabs (taps) visited per channel chunk
    const int nit = NXC * NCK * nsl;             // ... per workgroup
    auto slab_of = [&](int it) { return (it / nsl) * NSL + tap0 + it % nsl; };
    issue_slab(slab_of(0), 0);
    stage_x(0);
    __syncthreads();  // also drains the LDS-DMA (the barrier's release waits vmcnt(0))

    const int dil = a.dil;
    const int rowbase0 = wn * (NT / WN) + l31 - a.pad + PA;
    int s = 0;  // slabs visited so far
    for (int xc = 0; xc < NXC; ++xc) {
        if (xc > 0) {
            stage_x(xc);  // every wave passed the barrier that ended the previous slab: the old X is dead
            __syncthreads();
        }
        for (int ck = 0; ck < NCK; ++ck) {
            for (int sli = 0; sli < nsl; ++sli, ++s) {
                const int sl = tap0 + sli;
                // the next slab streams into the other buffer while this one feeds the MFMAs; every wave is past the
                // barrier that ended the previous one, so nobody still reads that buffer
                if ((s + 1) < nit) issue_slab(slab_of(s + 1), (s + 1) & 1);
                const unsigned char* abuf = ab + (T::NBUF == 2 ? (s & 1) * T::SLAB_BYTES : 0) + (size_t)(wm * MR) * 1024 + lane * 16;
                const int ntaps = (KS - sl * TG) < TG ? (KS - sl * TG) : TG;
                const int slot0 = ck * (CKC / 8) + lh;

                // software pipeline over (tap, k-step): fragments of step q+1 are read while step q's MFMAs issue
                int rowoff[NR], rowswz[NR];
                auto set_rows = [&](int tj) {
                    const int rowb = rowbase0 + (sl * TG + tj) * dil;
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        const int row = rowb + nr * 32;
                        rowoff[nr] = row * P;
                        rowswz[nr] = (row / RPB) & (SPR - 1);
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
                                bfn[nr] = *reinterpret_cast<const bf16x8*>(xt + rowoff[nr] + (((slot0 + (ks + 1) * 2) ^ rowswz[nr]) << 4));
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr) afn[mr] = *reinterpret_cast<const bf16x8*>(aslab + ((ks + 1) * MB + mr) * 1024);
                        } else if (tj + 1 < ntaps) {
                            set_rows(tj + 1);
#pragma unroll
                            for (int nr = 0; nr < NR; ++nr) bfn[nr] = *reinterpret_cast<const bf16x8*>(xt + rowoff[nr] + ((slot0 ^ rowswz[nr]) << 4));
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr) afn[mr] = *reinterpret_cast<const bf16x8*>(aslab + (KSTEPS * MB + mr) * 1024);
                        }
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                            for (int nr = 0; nr < NR; ++nr)
                                acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mr], bf[nr], acc[mr][nr], 0, 0, 0);
#pragma unroll
                        for (int mr = 0; mr < MR; ++mr) af[mr] = afn[mr];
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) bf[nr] = bfn[nr];
                    }
                }
                __syncthreads();  // slab s consumed by all waves, slab s+1 landed (vmcnt(0) before the barrier)
            }
        }
    }

    // ---- epilogue 1: accumulators + bias -> fp32 tile [NT][MT] in LDS (all MFMA reads are behind the barrier)
    float* ep = reinterpret_cast<float*>(lds);
    constexpr int EPF = T::EP_PITCH / 4;
    const float* __restrict__ bias = a.bias + mtile * MT;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int col = wm * (MT / T::WM) + mr * 32 + 8 * rq + 4 * lh;
            const float4 bv = *reinterpret_cast<const float4*>(bias + col);
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int row = wn * (NT / WN) + nr * 32 + l31;
                float4 v;
                v.x = vadd_raw(acc[mr][nr][4 * rq + 0], bv.x);
                v.y = vadd_raw(acc[mr][nr][4 * rq + 1], bv.y);
                v.z = vadd_raw(acc[mr][nr][4 * rq + 2], bv.z);
                v.w = vadd_raw(acc[mr][nr][4 * rq + 3], bv.w);
                *reinterpret_cast<float4*>(ep + row * EPF + col) = v;
            }
        }
    }
    __syncthreads();

    // ---- epilogue 2: coalesced pass, 8 channels (16 bytes of bf16) per thread-iteration
    constexpr int UPR = MT / 8;  // units per row
    const float s_out = a.slope_out;
    const float rdiv = mrf_recip(a.div);
    unsigned short* __restrict__ y = static_cast<unsigned short*>(a.y);
    const unsigned short* __restrict__ res = static_cast<const unsigned short*>(a.res);
    for (int u = tid; u < NT * UPR; u += THREADS) {
        const int row = u / UPR, c8 = u % UPR;
        const int t = t0 + row;
        if (t >= L) continue;
        const float4 p0 = *reinterpret_cast<const float4*>(ep + row * EPF + c8 * 8);
        const float4 p1 = *reinterpret_cast<const float4*>(ep + row * EPF + c8 * 8 + 4);
        float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        const size_t g = ((size_t)b * Lp + t) * COUTP + mtile * MT + c8 * 8;
        if (res) {  // ResBlock residual  x = xt + x  (model.py:50)
            const uint4 r = *reinterpret_cast<const uint4*>(res + g);
            v[0] = vadd_raw(bf16_lo(r.x), v[0]); v[1] = vadd_raw(bf16_hi(r.x), v[1]); v[2] = vadd_raw(bf16_lo(r.y), v[2]); v[3] = vadd_raw(bf16_hi(r.y), v[3]);
            v[4] = vadd_raw(bf16_lo(r.z), v[4]); v[5] = vadd_raw(bf16_hi(r.z), v[5]); v[6] = vadd_raw(bf16_lo(r.w), v[6]); v[7] = vadd_raw(bf16_hi(r.w), v[7]);
        }
        if (a.acc_add) {  // MRF  xs += rb(x)  (model.py:118-120)
            const uint4 o = *reinterpret_cast<const uint4*>(y + g);
            v[0] = vadd_raw(bf16_lo(o.x), v[0]); v[1] = vadd_raw(bf16_hi(o.x), v[1]); v[2] = vadd_raw(bf16_lo(o.y), v[2]); v[3] = vadd_raw(bf16_hi(o.y), v[3]);
            v[4] = vadd_raw(bf16_lo(o.z), v[4]); v[5] = vadd_raw(bf16_hi(o.z), v[5]); v[6] = vadd_raw(bf16_lo(o.w), v[6]); v[7] = vadd_raw(bf16_hi(o.w), v[7]);
        }
        if (a.div != 1.0f) {  // x = xs / num_kernels  (model.py:121)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * rdiv;
        }
        if (s_out != 1.0f) {  // the (only) consumer's LeakyReLU, applied once by the producer
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = lrelu_f(v[e], s_out);
        }
        *reinterpret_cast<uint4*>(y + g) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    }
}

// ---- tile table ------------------------------------------------------------------------------------
// Two workgroups per CU (<= 80 KiB LDS each) so that one workgroup's HBM phases (X staging, epilogue)
// overlap the other's MFMA phase.
//                        CINP XC  CKC COUTP KS  MT   NT  WM WN TG PA  IN_F32
template <int KS> using BRes256 = BTile<256, 128, 128, 256, KS, 128, 256, 2, 4, 1, (KS - 1) / 2 * 5, false>;  // 8 waves: stage 1 runs un-fused
template <int KS> using BRes128 = BTile<128, 128, 64, 128, KS, 128, 128, 2, 2, 1, (KS - 1) / 2 * 5, false>;
template <int KS> using BRes64 = BTile<64, 64, 64, 64, KS, 64, 256, 1, 4, 2, (KS - 1) / 2 * 5, false>;
template <int KS> using BRes32 = BTile<32, 32, 32, 32, KS, 32, 256, 1, 4, KS, (KS - 1) / 2 * 5, false>;
using BPre = BTile<128, 128, 64, 512, 7, 128, 128, 2, 2, 1, 3, true>;    // conv_pre: 80 (padded to 128) -> 512
using BUp0 = BTile<512, 128, 64, 2048, 3, 128, 128, 2, 2, 1, 1, false>;  // ups_0 as Conv1d(512 -> 8*256, k=3)
using BUp1 = BTile<256, 128, 64, 1024, 3, 128, 128, 2, 2, 1, 1, false>;  // ups_1 as Conv1d(256 -> 8*128, k=3)
using BUp2 = BTile<128, 128, 64, 128, 3, 128, 128, 2, 2, 1, 1, false>;   // ups_2 as Conv1d(128 -> 2*64,  k=3)
using BUp3 = BTile<64, 64, 64, 64, 3, 64, 256, 1, 4, 3, 1, false>;       // ups_3 as Conv1d(64  -> 2*32,  k=3)

template <class T>
static hipError_t launch_b(const BConvArgs& a, hipStream_t s) {
    static DynLdsOnce once;  // per device (vtts_internal.h)
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&conv_bf16_k<T>), T::LDS_BYTES, once); e != hipSuccess) return e;
    dim3 grid((a.L + T::NT - 1) / T::NT, T::COUTP / T::MT, a.B);
    hipLaunchKernelGGL(conv_bf16_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

template <template <int> class TT>
static hipError_t launch_b_ks(const BConvArgs& a, int K, hipStream_t s) {
    switch (K) {
        case 3: return launch_b<TT<3>>(a, s);
        case 7: return launch_b<TT<7>>(a, s);
        case 11: return launch_b<TT<11>>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv_bf16(int cls, int K, const BConvArgs& a, hipStream_t s) {
    switch (cls) {
        case BCLS_RES256: return launch_b_ks<BRes256>(a, K, s);
        case BCLS_RES128: return launch_b_ks<BRes128>(a, K, s);
        case BCLS_RES64: return launch_b_ks<BRes64>(a, K, s);
        case BCLS_RES32: return launch_b_ks<BRes32>(a, K, s);
        case BCLS_PRE: return launch_b<BPre>(a, s);
        case BCLS_UP0: return launch_b<BUp0>(a, s);
        case BCLS_UP1: return launch_b<BUp1>(a, s);
        case BCLS_UP2: return launch_b<BUp2>(a, s);
        case BCLS_UP3: return launch_b<BUp3>(a, s);
    }
    return hipErrorInvalidValue;
}

// tile geometry the host-side packer needs, per class
BPackGeom bf16_pack_geom(int cls, int K) {
    auto mk = [](int cinp, int ckc, int coutp, int ks, int mt, int tg) { return BPackGeom{cinp, ckc, coutp, ks, mt, tg}; };
    (void)K;
    switch (cls) {
        case BCLS_RES256: return mk(256, 128, 256, K, 128, 1);
        case BCLS_RES128: return mk(128, 64, 128, K, 128, 1);
        case BCLS_RES64: return mk(64, 64, 64, K, 64, 2);
        case BCLS_RES32: return mk(32, 32, 32, K, 32, K);
        case BCLS_PRE: return mk(128, 64, 512, 7, 128, 1);
        case BCLS_UP0: return mk(512, 64, 2048, 3, 128, 1);
        case BCLS_UP1: return mk(256, 64, 1024, 3, 128, 1);
        case BCLS_UP2: return mk(128, 64, 128, 3, 128, 1);
        case BCLS_UP3: return mk(64, 64, 64, 3, 64, 3);
    }
    return mk(0, 0, 0, 0, 0, 0);
}

// name prefix of the instantiation, as rocprofv3 prints it
const char* bf16_kernel_name(int cls, int K) {
    static thread_local char buf[96];
    const BPackGeom g = bf16_pack_geom(cls, K);
    const int xc = g.cinp < 128 ? g.cinp : 128;
    snprintf(buf, sizeof(buf), "conv_bf16_k<BTile<%d, %d, %d, %d, %d,", g.cinp, xc, g.ckc, g.coutp, g.ks);
    return buf;
}

static unsigned short f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

size_t bf16_packed_bytes(const BPackGeom& g) {
    const int ncc = g.cinp / g.ckc, nsl = (g.ks + g.tg - 1) / g.tg;
    return (size_t)(g.coutp / g.mt) * ncc * nsl * ((size_t)g.mt * g.tg * g.ckc * 2);
}

// Wc: conv weights [KS][cin_real][coutp] fp32 (Haiku layout).  Output: [mtile][cc][slab] slabs, each
// [tj][ks][mblk][lane][8] bf16 with  row = mblk*32 + (lane&31),  k = ks*16 + 8*(lane>>5) + e.
void bf16_pack(const float* Wc, int cin_real, const BPackGeom& g, unsigned short* out) {
    const int ncc = g.cinp / g.ckc, nsl = (g.ks + g.tg - 1) / g.tg, ksteps = g.ckc / 16, mb = g.mt / 32;
    size_t o = 0;
    for (int mtile = 0; mtile < g.coutp / g.mt; ++mtile)
        for (int cc = 0; cc < ncc; ++cc)
            for (int sl = 0; sl < nsl; ++sl)
                for (int tj = 0; tj < g.tg; ++tj)
                    for (int ks = 0; ks < ksteps; ++ks)
                        for (int mblk = 0; mblk < mb; ++mblk)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 8; ++e) {
                                    const int j = sl * g.tg + tj;
                                    const int ci = cc * g.ckc + ks * 16 + 8 * (lane >> 5) + e;
                                    const int co = mtile * g.mt + mblk * 32 + (lane & 31);
                                    float w = 0.f;
                                    if (j < g.ks && ci < cin_real) w = Wc[((size_t)j * cin_real + ci) * g.coutp + co];
                                    out[o++] = f32_to_bf16_rne(w);
                                }
}

// =====================================================================================================
// conv_post in the bf16 path: Conv1d 32 -> 1, k = 7 on channels-last bf16 input (already LeakyReLU(0.01)-ed
// by its producer) + tanh, fp32 waveform out (model.py:122-124).  Streaming, HBM-bound: 64 B in / 4 B out.
// =====================================================================================================
__global__ __launch_bounds__(256) void conv_post_bf16_k(BConvArgs a, float* __restrict__ wav, float* __restrict__ pre_act) {
    constexpr int C = 32, KS = 7, NT = 256, ROWS = NT + KS - 1, SPR = 4, P = 64;
    __shared__ __attribute__((aligned(16))) unsigned char xt[ROWS * P];
    __shared__ float ws[KS * C];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int Lp = a.L;                                      // rows allocated per utterance
    const int L = a.lens ? min(max(a.lens[b], 0) * a.len_mul, a.L) : a.L;  // valid rows of this utterance, clamped to its slot (ragged batch: the rest reads as zero padding)
    const long t0 = (long)blockIdx.x * NT;
    if (t0 >= L) return;  // samples past this utterance's end: zero by the caller's memset
    for (int i = tid; i < KS * C; i += 256) ws[i] = a.wf[i];  // Haiku [K][Cin][1] fp32
    const unsigned short* __restrict__ xb = static_cast<const unsigned short*>(a.x) + (size_t)b * Lp * C;
    for (int u = tid; u < ROWS * SPR; u += 256) {
        const int row = u / SPR, c = u % SPR;
        const long t = t0 - 3 + row;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (t >= 0 && t < L) v = *reinterpret_cast<const uint4*>(xb + t * C + c * 8);
        *reinterpret_cast<uint4*>(xt + row * P + ((c ^ ((row >> 2) & 3)) << 4)) = v;
    }
    __syncthreads();
    const long t = t0 + tid;
    if (t >= L) return;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        const int row = tid + j;
#pragma unroll
        for (int c = 0; c < SPR; ++c) {
            const uint4 v = *reinterpret_cast<const uint4*>(xt + row * P + ((c ^ ((row >> 2) & 3)) << 4));
            const float* w = ws + j * C + c * 8;
            acc = fmaf(w[0], bf16_lo(v.x), acc); acc = fmaf(w[1], bf16_hi(v.x), acc);
            acc = fmaf(w[2], bf16_lo(v.y), acc); acc = fmaf(w[3], bf16_hi(v.y), acc);
            acc = fmaf(w[4], bf16_lo(v.z), acc); acc = fmaf(w[5], bf16_hi(v.z), acc);
            acc = fmaf(w[6], bf16_lo(v.w), acc); acc = fmaf(w[7], bf16_hi(v.w), acc);
        }
    }
    const float p = acc + a.bias[0];
    const size_t idx = (size_t)b * Lp + t;
    if (pre_act) pre_act[idx] = p;
    wav[idx] = tanhf(p);
}

hipError_t launch_conv_post_bf16(const BConvArgs& a, float* wav, float* pre_act, hipStream_t s) {
    dim3 grid((a.L + 255) / 256, a.B);
    hipLaunchKernelGGL(conv_post_bf16_k, grid, dim3(256), 0, s, a, wav, pre_act);
    return hipGetLastError();
}

// bf16 [n] -> fp32 [n] (test taps)
__global__ void bf16_to_f32_k(const unsigned short* __restrict__ in, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = __builtin_bit_cast(float, (unsigned)in[i] << 16);
}
hipError_t launch_bf16_to_f32(const void* in, float* out, size_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(bf16_to_f32_k, dim3(blocks), dim3(256), 0, s, static_cast<const unsigned short*>(in), out, n);
    return hipGetLastError();
}
// fp32 [n] -> bf16 [n] (run_module inputs)
__global__ void f32_to_bf16_k(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (unsigned short)(pack_bf16x2(in[i], 0.f) & 0xffffu);
}
hipError_t launch_f32_to_bf16(const float* in, void* out, size_t n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(f32_to_bf16_k, dim3(blocks), dim3(256), 0, s, in, static_cast<unsigned short*>(out), n);
    return hipGetLastError();
}

}  // namespace vtts
