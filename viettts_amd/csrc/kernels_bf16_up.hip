// The four transposed convolutions of the generator in bf16 (vietTTS/hifigan/model.py:88-94, :112-114) on the
// register-streamed structure of the fused-pair kernel (kernels_bf16_rbg.hip), second generation of the upsampler.
//
// With channels-last activations the polyphase form of ConvTranspose1d(k = 2s, stride s) is two small GEMMs per input frame q
// (SURVEY.md Appendix A.2):  output rows (phase r, co) with r <  s/2 read frames (q - 1, q),
//                            output rows (phase r, co) with r >= s/2 read frames (q, q + 1),
// and the [L][s * Cout] result IS the [s * L][Cout] tensor.  The first generation ran this as ONE 3-tap convolution whose
// third tap per half is all zeros, with the weights double-buffered through LDS slabs behind one s_barrier per slab
// (MfmaUtil 0.21-0.33, 6-15 M LDS bank-conflict cycles per launch, profiles/r01_k_pmc_bf16.md).  Here:
//   * the input tile (N1 + 2 frames x Cin, already LeakyReLU-ed by its producer: the MRF mean / conv_pre epilogues store the
//     activated tensor) is staged once per workgroup, XOR-swizzled;
//   * the output rows are walked in chunks of WM * MR * 32 rows that never straddle the two halves: a chunk runs exactly its
//     two non-zero taps (no zero weights are multiplied), accumulators start from the bias block (C operand of the first MFMA);
//   * weight fragments stream L2 -> registers through a ring (buffer loads, k-step in an SGPR offset), activation fragments
//     from LDS with one row address + swizzle term per tap (the lean addressing of the pair kernel); no workgroup
//     synchronisation after the tile is staged;
//   * the epilogue packs to bf16 and stores 16 bytes per lane straight from the accumulator layout (v_permlane32_swap).
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "bf16_common.h"

#ifndef VTTS_UP_STAGE_OUT  // LDS-staged, row-contiguous output of the two-chunk tiles (A/B switch)
#define VTTS_UP_STAGE_OUT 1
#endif

namespace vtts {

// TAPS / HALVES: the transposed convolutions run two taps per chunk, rows of the first half of the chunks on frames (q - 1, q), the others on
// (q, q + 1) (HALVES); conv_pre (Conv1d 80 -> 512, k = 7, model.py:83,110) is the same structure with all TAPS = 7 taps in every chunk, frames
// q - 3 .. q + 3, and its fp32 mel rows (CINR = 80 channels) converted to bf16 while staging (INF32; channels 80 .. 127 of the tile are zero and
// meet zero weights).
template <int CIN_, int M_, int N1_, int WM_, int WN_, int MR_, int PA_, int MINWG_, int TAPS_ = 2, bool HALVES_ = true, int CINR_ = CIN_, bool INF32_ = false>
struct UTile {
    static constexpr int CIN = CIN_, M = M_, N1 = N1_, WM = WM_, WN = WN_, MR = MR_, PA = PA_, MINWG = MINWG_;
    static constexpr int TAPS = TAPS_, CINR = CINR_;
    static constexpr bool HALVES = HALVES_, INF32 = INF32_;
    static constexpr int HL = HALVES ? 1 : (TAPS - 1) / 2;   // tile row 0 holds frame t0 - HL
    static constexpr int PT = HALVES ? 3 : TAPS;             // taps in the packed weights (the transposed convolutions' 3-tap form)
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int NR = N1 / WN / 32;
    static constexpr int MC = WM * MR * 32;             // output rows per chunk
    static constexpr int NCH = M / MC;                  // chunks; the first NCH / 2 read frames (q - 1, q), the others (q, q + 1)
    static constexpr int SPR = CIN / 8, P = CIN * 2;    // 16-byte slots / bytes per tile row
    static constexpr int ROWS = N1 + (HALVES ? 2 : TAPS - 1);
    static constexpr int KSTEPS = CIN / 16;             // k-steps per tap
    static constexpr int NQ = TAPS * KSTEPS;            // k-steps per chunk
    static constexpr int MB = M / 32;
    static constexpr int RA = PA + 1;
    static constexpr int UB = KSTEPS < 8 ? KSTEPS : 8;  // k-steps per block: a block never straddles taps
    static constexpr int XPT = (ROWS * SPR + THREADS - 1) / THREADS;
    static constexpr int LDS_BYTES = ROWS * P;
    static_assert(N1 % (WN * 32) == 0 && M % MC == 0 && (!HALVES || (M / 2) % MC == 0), "chunks must not straddle the two halves");
    static_assert(HALVES ? TAPS == 2 : (TAPS % 2 == 1), "two taps per half, or a centred odd convolution");
    static_assert(!INF32 || (CINR % 8 == 0 && CINR <= CIN), "fp32 rows are converted 8 channels at a time");
    static_assert(RA == 4 && UB % RA == 0 && KSTEPS % UB == 0, "ring slot / B parity are a step's position in its block");
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    static_assert((NCH & (NCH - 1)) == 0, "the launcher splits the chunks over 1, 2, 4 ... workgroups");
    // ups_2 / ups_3 (two chunks, M = Cin): the epilogue above stores 32 bytes per frame and instruction (an output row is M * 2 = 128 /
    // 256 bytes, a lane owns 16 of them), which is what kept these two HBM-bound layers on the first-generation kernel (round 2: 684 vs
    // 687 us, 597 vs 455).  Both chunks' packed results fit in registers (64 VGPRs), so they are held until the last chunk is done, go
    // through the — by then dead — input tile in LDS, and leave as whole rows: a wave stores 1 KiB contiguous per instruction.
    static constexpr int SPRO = M / 8;                  // 16-byte slots per OUTPUT row
    static constexpr bool STAGE_OUT = VTTS_UP_STAGE_OUT && HALVES && NCH == 2 && tile_rows16(N1) * M * 2 <= LDS_BYTES && NCH * MR * NR * 8 <= 64 &&
                                      (N1 * SPRO) % THREADS == 0;
};

template <class T>
__global__ __launch_bounds__(T::THREADS, (T::MINWG * T::THREADS + 255) / 256) void convt_g_bf16_k(BConvArgs a) {
    constexpr int CIN = T::CIN, M = T::M, N1 = T::N1, WN = T::WN, MR = T::MR, NR = T::NR, PA = T::PA, RA = T::RA;
    constexpr int THREADS = T::THREADS, MC = T::MC, NCH = T::NCH, SPR = T::SPR, P = T::P, ROWS = T::ROWS, KSTEPS = T::KSTEPS;
    constexpr int NQ = T::NQ, MB = T::MB, UB = T::UB, XPT = T::XPT;
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) unsigned char xt[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int t0 = blockIdx.x * N1;  // first input frame of this workgroup
    const int b = a.zrev ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z;
    const int Lp = a.L;
    const int L = a.lens ? min(max(a.lens[b], 0) * a.len_mul, a.L) : a.L;  // valid input rows of this utterance (ragged batches)
    if (t0 >= L) return;
    const unsigned short* __restrict__ xg = static_cast<const unsigned short*>(a.x) + (size_t)b * Lp * CIN;  // bf16 input rows (!INF32)
    [[maybe_unused]] const float* __restrict__ xg32 = static_cast<const float*>(a.x) + (size_t)b * Lp * T::CINR;  // fp32 input rows (INF32: conv_pre's mel)
    unsigned short* __restrict__ yg = static_cast<unsigned short*>(a.y) + (size_t)b * Lp * M;

    // ---------------- input tile: frames t0 - 1 .. t0 + N1 (zero outside the utterance: lax "SAME"), swizzled ds_write_b128 ----------------
    {
        const float sin_ = a.slope_in;
        auto act2 = [&](unsigned u) { return sin_ == 1.0f ? u : lrelu_bf16x2(u, sin_); };
        // every load of the tile in flight at once (nothing else is live yet; a batch of 4 per round trip left a 128 -> 2 x 64
        // workgroup 10 us in staging for 2.5 us of MFMAs): unconditional, from clamped addresses, masked afterwards
        constexpr int XB = XPT;
#pragma unroll 1
        for (int i0 = 0; i0 < XPT; i0 += XB) {
            uint4 v[XB];
            bool ok[XB];
            int row[XB], c[XB];
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                const int u = tid + (i0 + i) * THREADS;
                row[i] = u / SPR;
                c[i] = u % SPR;
                const int t = t0 - T::HL + row[i];
                ok[i] = u < ROWS * SPR && t >= 0 && t < L;
                const int tc = t < 0 ? 0 : (t >= L ? L - 1 : t);
                if constexpr (T::INF32) {  // 8 fp32 channels -> 4 bf16 pairs (round to nearest even, as every other producer of the bf16 path)
                    const bool real = c[i] * 8 < T::CINR;
                    const float4 lo = *reinterpret_cast<const float4*>(xg32 + (size_t)tc * T::CINR + (real ? c[i] * 8 : 0));
                    const float4 hi = *reinterpret_cast<const float4*>(xg32 + (size_t)tc * T::CINR + (real ? c[i] * 8 + 4 : 4));
                    v[i] = make_uint4(pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w));
                    ok[i] = ok[i] && real;
                } else {
                    v[i] = *reinterpret_cast<const uint4*>(xg + (size_t)tc * CIN + c[i] * 8);
                }
            }
#pragma unroll
            for (int i = 0; i < XB; ++i) {
                if (!ok[i]) v[i] = make_uint4(0u, 0u, 0u, 0u);
                v[i].x = act2(v[i].x);
                v[i].y = act2(v[i].y);
                v[i].z = act2(v[i].z);
                v[i].w = act2(v[i].w);
                if (tid + (i0 + i) * THREADS < ROWS * SPR) *reinterpret_cast<uint4*>(xt + row[i] * P + ((c[i] ^ swz_of<SPR>(row[i])) << 4)) = v[i];
            }
        }
    }
    __syncthreads();

    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, T::PT * CIN * M * 2, 0x00020000);
    const unsigned a_voff = (unsigned)((wm * MR) * 64 + lane) * 16;  // lane's bytes inside a k-step's [MB][64][16 B], chunk 0
    const int rowbase0 = wn * (N1 / WN) + l31;                         // this lane's column of block 0 = tile row of frame q - HL

    f32x16 acc[MR][NR];
    f32x16 bblk[MR];
    bf16x8 af[RA][MR], bf[2][NR];

    auto load_bias = [&](int c) {  // rows c*MC + wm*MR*32 + mr*32 + 8*rq + 4*lh + i  (r = 4*rq + i)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef float f32x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const float* __restrict__ bp = a.bias + c * MC + (wm * MR + mr) * 32 + 4 * lh;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(bp), q1 = *reinterpret_cast<const f32x4*>(bp + 8);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(bp + 16), q3 = *reinterpret_cast<const f32x4*>(bp + 24);
            const f32x8 lo = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7);
            bblk[mr] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
        }
    };
    // A fragments of flat step g = c * NQ + s (chunk c, step s = tap * KSTEPS + ks): the weight stream runs across the chunks
    auto load_a = [&](int g, int slot) {
        const int gc = g < NCH * NQ ? g : NCH * NQ - 1;  // the last look-aheads re-read the last step (inside the blob)
        const int c = gc / NQ, sidx = gc - c * NQ;
        const int h = T::HALVES && c >= NCH / 2 ? 1 : 0;
        const int f = h + sidx / KSTEPS, ks = sidx % KSTEPS;
        const int soff = ((f * KSTEPS + ks) * MB + c * (MC / 32)) * 1024;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_voff + mr * 1024, soff, 0);
            af[slot][mr] = __builtin_bit_cast(bf16x8, v);
        }
    };
    auto tap_terms = [&](int f, unsigned& tapaddr, unsigned& xs) {
        const int row = rowbase0 + f;
        tapaddr = (unsigned)row * P;
        xs = (unsigned)(swz_of<SPR>(row) ^ lh) << 4;
    };
    auto load_b = [&](unsigned tapaddr, unsigned xs, int ks, int par) {
        const unsigned addr = tapaddr + (xs ^ (unsigned)(ks << 5));
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) bf[par][nr] = *reinterpret_cast<const bf16x8*>(xt + addr + nr * 32 * P);
    };
    auto pin_step = [&]() {
        constexpr int NM = MR * NR, MEM = MR + NR;
        int done = 0;
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            const int upto = (i + 1) * MEM / NM;
            for (; done < upto; ++done) {
                if (done < MR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };
    auto swap_pair = [](unsigned& pd, unsigned& qd) {
        auto r = __builtin_amdgcn_permlane32_swap(pd, qd, false, false);
        pd = r[0];
        qd = r[1];
    };

    // small launches (batch-1 latency) split the chunks over gridDim.y workgroups per tile (each stages the tile itself)
    const int cpw = NCH / (int)gridDim.y, c_lo = (int)blockIdx.y * cpw, c_hi = c_lo + cpw;
#pragma unroll
    for (int s = 0; s < PA; ++s) load_a(c_lo * NQ + s, s % RA);
    load_bias(c_lo);

    // one chunk: its two taps' MFMAs, then `sink(mr, p, nr, rb, t, packed)` per 8 output rows of a frame
    auto run_chunk = [&](int c, bool more, auto&& sink) {
        const int h = T::HALVES && c >= NCH / 2 ? 1 : 0;
        {
            unsigned ta, xs;
            tap_terms(h, ta, xs);
            load_b(ta, xs, 0, 0);
        }
        // blocks of UB k-steps; block bi covers steps bi*UB .. of this chunk: tap = (bi*UB) / KSTEPS
        auto block = [&](int bi, auto first_tag) {
            const int s0 = bi * UB;
            const int tp = s0 / KSTEPS, ksb = s0 - tp * KSTEPS;
            unsigned ta, xs, tn, xn;
            tap_terms(h + tp, ta, xs);
            const bool wrap = ksb + UB >= KSTEPS;  // the block's last look-ahead B fragment is the next tap's first
            tap_terms(h + (wrap ? (tp + 1 < T::TAPS ? tp + 1 : T::TAPS - 1) : tp), tn, xn);
            const int ksn = wrap ? 0 : ksb + UB;
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                load_a(c * NQ + s0 + i + PA, (i + PA) % RA);
                if (i + 1 < UB) load_b(ta, xs, ksb + i + 1, (i + 1) & 1);
                else load_b(tn, xn, ksn, (i + 1) & 1);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i % RA][mr], bf[i & 1][nr],
                                                                              (decltype(first_tag)::value && i == 0) ? bblk[mr] : acc[mr][nr], 0, 0, 0);
                pin_step();
            }
        };
        block(0, std::true_type{});
#pragma unroll 1
        for (int bi = 1; bi < NQ / UB; ++bi) block(bi, std::false_type{});
        if (more) load_bias(c + 1);  // lands while this chunk's epilogue runs

        // ---------------- epilogue: [consumer's LeakyReLU] -> bf16, 8 consecutive rows (phase, co) of frame t0 + n per lane ----------------
        const float s_out = a.slope_out;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int rb = c * MC + (wm * MR + mr) * 32 + 16 * p;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const int t = t0 + wn * (N1 / WN) + nr * 32 + l31;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[mr][nr][8 * p + e];
                    if (s_out != 1.0f) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = lrelu_f(v[e], s_out);
                    }
                    unsigned p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
                    unsigned q0 = pack_bf16x2(v[4], v[5]), q1 = pack_bf16x2(v[6], v[7]);
                    swap_pair(p0, q0);
                    swap_pair(p1, q1);
                    sink(mr, p, nr, rb, t, make_uint4(p0, p1, q0, q1));
                }
            }
        }
    };
    auto store_direct = [&](int, int, int, int rb, int t, const uint4& v) {  // 16-byte stores straight from the accumulator layout
        if (t < L) *reinterpret_cast<uint4*>(yg + (size_t)t * M + rb + 8 * lh) = v;
    };

    if constexpr (T::STAGE_OUT) {
        if (gridDim.y == 1) {  // both chunks in this workgroup
            constexpr int SPRO = T::SPRO;
            uint4 pk[2][MR][2][NR];
            run_chunk(0, true, [&](int mr, int p, int nr, int, int, const uint4& v) { pk[0][mr][p][nr] = v; });
            run_chunk(1, false, [&](int mr, int p, int nr, int, int, const uint4& v) { pk[1][mr][p][nr] = v; });
            __syncthreads();  // every wave is done reading the input tile: it becomes the output tile [frame][M], same tile_off layout
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int nr = 0; nr < NR; ++nr) {
                            const int n = wn * (N1 / WN) + nr * 32 + l31;
                            const int slot = (ci * MC + (wm * MR + mr) * 32 + 16 * p) / 8 + lh;
                            *reinterpret_cast<uint4*>(xt + tile_off<SPRO>(n, slot)) = pk[ci][mr][p][nr];
                        }
            __syncthreads();
            constexpr int UPT = N1 * SPRO / THREADS;  // whole rows out: consecutive lanes = consecutive 16-byte units of consecutive frames
            uint4 ov[UPT];
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const int u = tid + i * THREADS;
                ov[i] = *reinterpret_cast<const uint4*>(xt + tile_off<SPRO>(u / SPRO, u % SPRO));
            }
#pragma unroll
            for (int i = 0; i < UPT; ++i) {
                const int u = tid + i * THREADS, n = u / SPRO;
                if (t0 + n < L) *reinterpret_cast<uint4*>(yg + (size_t)(t0 + n) * M + (u % SPRO) * 8) = ov[i];
            }
            return;
        }
    }
#pragma unroll 1
    for (int c = c_lo; c < c_hi; ++c) run_chunk(c, c + 1 < c_hi, store_direct);
}

// ---- tile table (Cin, rows = stride * Cout, frames per workgroup, waves, m-blocks per wave) ------------------------------
//                        CIN    M   N1  WM WN MR PA MINWG
using UT0 = UTile<512, 2048, 64, 4, 1, 2, 3, 2>;   // ups_0: 512 -> 8 x 256, k = 16   (LDS 66 KiB)
using UT1 = UTile<256, 1024, 128, 4, 1, 2, 3, 2>;  // ups_1: 256 -> 8 x 128, k = 16   (LDS 65 KiB)
using UT2 = UTile<128, 128, 256, 1, 4, 2, 3, 2>;   // ups_2: 128 -> 2 x 64, k = 4     (LDS 64.5 KiB)
using UT3 = UTile<64, 64, 512, 1, 4, 1, 3, 2>;     // ups_3: 64 -> 2 x 32, k = 4      (LDS 64.3 KiB)
using UTP = UTile<128, 512, 128, 4, 1, 2, 3, 2, 7, false, 80, true>;  // conv_pre: 80 (-> 128) -> 512, k = 7, fp32 mel in (LDS 33.5 KiB)

template <class T>
static hipError_t launch_u(const BConvArgs& a, hipStream_t s) {
    static DynLdsOnce once;  // per device (vtts_internal.h)
    if (hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(&convt_g_bf16_k<T>), T::LDS_BYTES, once); e != hipSuccess) return e;
    const long tiles = (long)((a.L + T::N1 - 1) / T::N1) * a.B;
    int gy = 1;
    while (gy < T::NCH && tiles * gy < 512) gy *= 2;  // NCH is a power of two
    dim3 grid((a.L + T::N1 - 1) / T::N1, gy, a.B);
    hipLaunchKernelGGL(convt_g_bf16_k<T>, grid, dim3(T::THREADS), T::LDS_BYTES, s, a);
    return hipGetLastError();
}

// a.x = input bf16 [B][L][Cin]; a.wp = the 3-tap form's weights in pair_g_pack_geom-style order [tap][k-step][m-block][lane][8]
// (convt_g_pack_geom); a.bias = fp32 [stride * Cout]; a.y = bf16 [B][L][stride * Cout]
hipError_t launch_convt_g_bf16(int cls, const BConvArgs& a, hipStream_t s) {
    switch (cls) {
        case BCLS_UP0: return launch_u<UT0>(a, s);
        case BCLS_UP1: return launch_u<UT1>(a, s);
        case BCLS_UP2: return launch_u<UT2>(a, s);
        case BCLS_UP3: return launch_u<UT3>(a, s);
        case BCLS_PRE: return launch_u<UTP>(a, s);
    }
    return hipErrorInvalidValue;
}

BPackGeom convt_g_pack_geom(int cls) {
    switch (cls) {
        case BCLS_UP0: return BPackGeom{512, 512, 2048, 3, 2048, 1};
        case BCLS_UP1: return BPackGeom{256, 256, 1024, 3, 1024, 1};
        case BCLS_UP2: return BPackGeom{128, 128, 128, 3, 128, 1};
        case BCLS_UP3: return BPackGeom{64, 64, 64, 3, 64, 1};
        case BCLS_PRE: return BPackGeom{128, 128, 512, 7, 512, 1};  // the plain convolution's 7 taps, input channels padded 80 -> 128 with zeros
    }
    return BPackGeom{0, 0, 0, 0, 0, 0};
}

const char* convt_g_kernel_name(int cls) {
    static thread_local char buf[64];
    if (cls == BCLS_PRE) snprintf(buf, sizeof(buf), "convt_g_bf16_k<UTile<128, 512,");
    else snprintf(buf, sizeof(buf), "convt_g_bf16_k<UTile<%d,", cls == BCLS_UP0 ? 512 : cls == BCLS_UP1 ? 256 : cls == BCLS_UP2 ? 128 : 64);
    return buf;
}

}  // namespace vtts
