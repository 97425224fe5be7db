// fp32 implicit-GEMM dilated Conv1d on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// This kernel family is 96.8 % of the generator's FLOPs: the 72 ResBlock1 convolutions
// (vietTTS/hifigan/model.py:21-28 convs1 with rate d, :33-40 convs2 with rate 1), C -> C with
// C in {256,128,64,32}, K in {3,7,11}.
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

namespace vtts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int round_up4(int v) { return (v + 3) & ~3; }
constexpr int MAX_DIL = 5;  // resblock_dilation_sizes max in V1; larger rates fall back to the generic kernel

template <int C_, int KS_, int MT_, int NT_, int WM_, int WN_, int CK_>
struct ConvTile {
    static constexpr int C = C_, KS = KS_, MT = MT_, NT = NT_, WM = WM_, WN = WN_, CK = CK_;
    static constexpr int MR = MT / WM / 32;           // 32x32 accumulator blocks per wave along M
    static constexpr int NR = NT / WN / 32;           // ... along N
    static constexpr int PA = round_up4((KS - 1) / 2 * MAX_DIL);  // staged halo per side
    static constexpr int W = NT + 2 * PA;             // staged columns (multiple of 4)
    static constexpr int RS = W;                      // LDS row stride in floats
    static constexpr int NCH = C / CK;                // input-channel chunks
    static constexpr int CQ = CK / 8;                 // float4 A loads per tap per chunk
    static constexpr int NIT = KS * CQ;               // main-loop iterations per chunk
    static constexpr int LDS_BYTES = CK * RS * 4;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(MT % (WM * 32) == 0 && NT % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(C % MT == 0 && C % CK == 0 && CK % 8 == 0, "channel tiling");
};

// Packed weight layout (floats): [mblk = C/32][chunk][j][cq][lane = 64][e = 4]
//   element = W_hk[j][ci = chunk*CK + 2*(4*cq + e) + (lane >> 5)][co = mblk*32 + (lane & 31)]
// i.e. for k-step (j, cp = 4*cq + e) lane l holds A[i = l&31][k = l>>5] of the 32x32x2 MFMA.
template <class T>
__global__ __launch_bounds__(256) void conv1d_f32_mfma_k(ConvArgs a) {
    constexpr int C = T::C, KS = T::KS, MT = T::MT, NT = T::NT, WN = T::WN, CK = T::CK;
    constexpr int MR = T::MR, NR = T::NR, PA = T::PA, W = T::W, RS = T::RS, NCH = T::NCH, CQ = T::CQ, NIT = T::NIT;

    extern __shared__ __attribute__((aligned(16))) float xs[];  // [CK][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int t0 = blockIdx.x * NT;
    const int m0 = blockIdx.y * MT + wm * (MT / T::WM);  // first output channel of this wave
    const int b = blockIdx.z;
    const int L = a.L;
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

    // A operands for the first iteration
    float4 a_cur[MR], a_nxt[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) a_cur[mr] = wbase[mr][0];

    const int dil = a.dil;
    const int colbase = wn * (NT / WN) + l31 - a.pad + PA;  // + j*dil + nr*32

    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk) __syncthreads();  // all waves done reading the previous chunk
        // ---- stage CK input channels x W columns, LeakyReLU fused, zero outside [0, L) ----
        {
            constexpr int W4 = W / 4;
            const float* __restrict__ xc = xb + (long)(chunk * CK) * a.x_sc;
            for (int idx = tid; idx < CK * W4; idx += 256) {
                const int row = idx / W4;
                const int c4 = idx - row * W4;
                const int t = t0 - PA + 4 * c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t >= 0 && t < L) v = *reinterpret_cast<const float4*>(xc + (long)row * a.x_sc + t);
                v.x = lrelu(v.x, slope);
                v.y = lrelu(v.y, slope);
                v.z = lrelu(v.z, slope);
                v.w = lrelu(v.w, slope);
                *reinterpret_cast<float4*>(&xs[row * RS + 4 * c4]) = v;
            }
        }
        __syncthreads();

        // ---- MFMA over (tap j, channel pairs) of this chunk ----
        for (int it = 0; it < NIT; ++it) {
            const int j = it / CQ;
            const int cq = it - j * CQ;
            // prefetch the next iteration's A operands (next chunk's first one included)
            const long nxt = (long)chunk * NIT + it + 1;
            const bool has_next = nxt < (long)NCH * NIT;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) a_nxt[mr] = has_next ? wbase[mr][nxt * 64] : make_float4(0.f, 0.f, 0.f, 0.f);

            const float* xrow = &xs[(cq * 8 + lh) * RS + colbase + j * dil];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bf[NR];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bf[nr] = xrow[e * 2 * RS + nr * 32];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    const float av = e == 0 ? a_cur[mr].x : e == 1 ? a_cur[mr].y : e == 2 ? a_cur[mr].z : a_cur[mr].w;
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[nr], acc[mr][nr], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) a_cur[mr] = a_nxt[mr];
        }
    }

    // ---- epilogue: C/D layout col = lane&31 (time), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel) ----
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int t = t0 + wn * (NT / WN) + nr * 32 + l31;
            if (t < L) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + mr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const long idx = ((long)b * C + co) * L + t;
                    epilogue_store(a, idx, acc[mr][nr][r] + a.bias[co]);
                }
            }
        }
    }
}

// ---- tile selection --------------------------------------------------------------------------
//                        C   KS  MT   NT  WM WN CK
template <int KS> using Tile256 = ConvTile<256, KS, 128, 128, 2, 2, 64>;
template <int KS> using Tile128 = ConvTile<128, KS, 128, 128, 2, 2, 64>;
template <int KS> using Tile64 = ConvTile<64, KS, 64, 128, 2, 2, 64>;
template <int KS> using Tile32 = ConvTile<32, KS, 32, 256, 1, 4, 32>;

template <class T>
static hipError_t launch_tile(const ConvArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_f32_mfma_k<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid((a.L + T::NT - 1) / T::NT, T::C / T::MT, a.B);
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

bool conv1d_f32_mfma_supported(int C, int K, int dil, int L) {
    if (!(C == 256 || C == 128 || C == 64 || C == 32)) return false;
    if (!(K == 3 || K == 7 || K == 11)) return false;
    if (dil < 1 || dil > MAX_DIL) return false;
    if (L % 4 != 0) return false;  // float4 staging
    return true;
}

static int chunk_of(int C) { return C >= 64 ? 64 : 32; }

size_t conv1d_f32_mfma_packed_floats(int C, int K) { return (size_t)C * C * K; }

void conv1d_f32_mfma_pack(const float* w_hk, int C, int K, float* out) {
    const int CK = chunk_of(C), NCH = C / CK, CQ = CK / 8;
    size_t o = 0;
    for (int mblk = 0; mblk < C / 32; ++mblk)
        for (int chunk = 0; chunk < NCH; ++chunk)
            for (int j = 0; j < K; ++j)
                for (int cq = 0; cq < CQ; ++cq)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int ci = chunk * CK + 2 * (4 * cq + e) + (lane >> 5);
                            const int co = mblk * 32 + (lane & 31);
                            out[o++] = w_hk[((size_t)j * C + ci) * C + co];
                        }
}

hipError_t launch_conv1d_f32_mfma(const ConvArgs& a, hipStream_t s) {
    switch (a.Cin) {
        case 256: return launch_ks<Tile256>(a, s);
        case 128: return launch_ks<Tile128>(a, s);
        case 64: return launch_ks<Tile64>(a, s);
        case 32: return launch_ks<Tile32>(a, s);
    }
    return hipErrorInvalidValue;
}

const char* conv1d_f32_mfma_kernel_name(int C, int K) {
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "conv1d_f32_mfma_k<ConvTile<%d, %d", C, K);
    return buf;
}

// transposed convolution on MFMA: not instantiated yet — the engine uses the generic polyphase kernel.
bool convT1d_f32_mfma_supported(int, int, int, int, int, int) { return false; }
size_t convT1d_f32_mfma_packed_floats(int, int, int) { return 0; }
void convT1d_f32_mfma_pack(const float*, int, int, int, int, int, float*) {}
hipError_t launch_convT1d_f32_mfma(const ConvArgs&, hipStream_t) { return hipErrorNotSupported; }

}  // namespace vtts
