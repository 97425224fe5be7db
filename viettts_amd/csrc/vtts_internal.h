// Internal declarations shared by the engine and the kernel translation units.
// gfx950 only: wave64, MFMA, 160 KiB LDS.  No portability layer on purpose.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vtts {

// engine.hip: record the calling thread's last error message (vtts_last_error()) and return `code`
int set_error(int code, const char* msg);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function ON A DEVICE: a process that drives several GPUs (a
// second Generator on cuda:1 after cuda:0) must set it once per device, not once per process.  `done` = the caller's static
// per-kernel table; devices beyond it just set the attribute on every launch (cheap next to a launch).
struct DynLdsOnce {
    bool done[32] = {};
};
inline hipError_t set_max_dynamic_lds(const void* fn, int bytes, DynLdsOnce& once) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < 32;
    if (tracked && once.done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && tracked) once.done[dev] = true;
    return e;
}

// How a convolution's result is combined with what is already in memory.  This is where
// ResBlock1's residual (model.py:50 `x = xt + x`) and the MRF mean (model.py:115-121
// `xs = rb0; xs += rb1; xs += rb2; x = xs / 3`) are fused into the producing kernel.
//   v = acc + bias;  if (res) v = v + res[i];
//   ACC_STORE : out[i] = v
//   ACC_ADD   : out[i] = out[i] + v
//   ACC_MEAN  : out[i] = (out[i] + v) / div      (true division, as the reference)
enum AccMode : int { ACC_STORE = 0, ACC_ADD = 1, ACC_MEAN = 2 };

struct ConvArgs {
    const float* x;      // input activations
    long x_sb, x_sc, x_st;  // element strides of x: batch, channel, time  (NCW: C*L, L, 1)
    const float* w;      // plain weights, Haiku layout [K][Cin][Cout]  (convT: [K][Cout][Cin])
    const void* wp;      // MFMA-packed weights (or nullptr)
    const float* bias;   // [Cout]
    const float* res;    // optional residual [B][Cout][L] (may alias y) or nullptr
    float* y;            // output [B][Cout][Lout]
    int B, Cin, Cout, K;
    int dil, pad;        // convolution: rate and symmetric zero pad
    int stride, pad_a;   // transposed convolution: stride and left pad of the zero-stuffed input
    int L;               // input length (time)
    int Lout;            // output length
    float slope_in;      // LeakyReLU slope applied to x on load (1.0f = identity)
    int acc_mode;        // AccMode
    float div;           // divisor for ACC_MEAN
    int tanh_out;        // apply tanh to the result (conv_post)
    float* pre_act;      // optional copy of the pre-tanh value (same indexing as y) or nullptr
    int tile_pref;       // MFMA time-tile choice: 0 = by problem size, 1 = wide, 2 = narrow (tests)
    int zrev;            // MFMA kernels: 1 = utterance = gridDim.z - 1 - blockIdx.z (engine.hip: next_zrev)
    // ragged batch (vtts_hifigan_forward_ragged on the fp32 / bf16x3 engines): utterance b's valid INPUT length is lens[b] * len_mul of the L
    // allocated (L stays the row pitch); everything past it reads as the reference's zero padding (model.py:8-10, lax "SAME"), nothing past
    // it is stored, tiles past it exit at once.  nullptr = all L columns valid.  device_common.h: valid_len
    const int* lens;     // [B] mel frames per utterance (device memory)
    int len_mul;         // input columns of this layer per mel frame
};

// ---- generic (any shape) fp32 kernels: kernels_generic.hip -------------------------------
hipError_t launch_conv1d_generic(const ConvArgs& a, hipStream_t s);
hipError_t launch_convT1d_generic(const ConvArgs& a, hipStream_t s);

// ---- fp32 MFMA implicit-GEMM convolution: kernels_f32_mfma.hip ------------------------------
// ResBlock shapes (Cin == Cout in {32,64,128,256}, K in {3,7,11}, channel-major input) and conv_pre
// (80 -> 512, K = 7, time-major input: nwc = true).
bool conv1d_f32_mfma_supported(int Cin, int Cout, int K, int dil, int L, bool nwc);
size_t conv1d_f32_mfma_packed_floats(int Cin, int Cout, int K);
// host-side re-layout: Haiku [K][Cin][Cout] -> MFMA A-fragment order
void conv1d_f32_mfma_pack(const float* w_hk, int Cin, int Cout, int K, float* out);
hipError_t launch_conv1d_f32_mfma(const ConvArgs& a, hipStream_t s);
const char* conv1d_f32_mfma_kernel_name(int C, int K);

// ---- fp32 fused ResBlock1 pair  x' = c2(lrelu(c1(lrelu(x)))) + x  (kernels_f32_pair.hip) ----
// a = c1's ConvArgs (x, wp = c1's packed weights, bias = b1, dil, slope_in, B, L, zrev) with the OUTPUT side filled in (y, res = x,
// acc_mode, div); wp2 / bias2 = c2's.  Bit-identical to launching the two convolutions one after the other.
bool pair_f32_supported(int C, int K, int dil, int L);
hipError_t launch_pair_f32(const ConvArgs& a, const void* wp2, const float* bias2, hipStream_t s);

// ---- split-operand ("bf16x3") fused ResBlock1 pair on the bf16 matrix pipe, fp32 channel-major activations (kernels_x3.hip) ----
// a as for launch_pair_f32 (a.wp unused); w1 / w2 = the two convolutions' packed weights, each [hi fragments][lo fragments] (pair_x3_pack)
bool pair_x3_supported(int C, int K, int dil, int L);
size_t pair_x3_conv_bytes(int C, int K);
void pair_x3_pack(const float* w_hk, int C, int K, unsigned short* out);
hipError_t launch_pair_x3(const ConvArgs& a, const void* w1, const void* w2, const float* bias2, hipStream_t s);
// ... and the four transposed convolutions (polyphase, k = 2 * stride): a.x [B][Cin][L] -> a.y [B][Cout][stride * L], a.wp = convt_x3_pack's output
bool convt_x3_supported(int Cin, int Cout, int K, int stride, int pad_a, int L);
size_t convt_x3_bytes(int Cin, int Cout, int stride);
void convt_x3_pack(const float* w_hk, int Cin, int Cout, int K, int stride, int pad_a, unsigned short* out);
hipError_t launch_convt_x3(const ConvArgs& a, hipStream_t s);
// ... and conv_pre (80 -> 512, k = 7, mel NWC in): a.x [B][L][80] (a.x_sb = the batch stride), a.y [B][512][L], a.wp = conv_pre_x3_pack's output
bool conv_pre_x3_supported(int Cin, int Cout, int K, int dil);
size_t conv_pre_x3_bytes();
void conv_pre_x3_pack(const float* w_hk, unsigned short* out);
hipError_t launch_conv_pre_x3(const ConvArgs& a, hipStream_t s);

// ---- whole ResBlock1 with split operands (kernels_x3_rb.hip): three pairs + the MRF bookkeeping in one launch, bit-identical to three launch_pair_x3 ----
// a = ConvArgs of the ResBlock (x = stage input [B][C][L], y = output with acc_mode / div, Cin = C, K, B, L, lens / len_mul, slope_in, zrev);
// dils = the three rates; w[q] / bias[q] = the six convolutions (c1_0, c2_0, c1_1, ...), weights as pair_x3_pack lays them out
bool resblock_x3_supported(int C, int K, const int* dils, int L);
bool resblock_x3_preferred(int C, int K);
hipError_t launch_resblock_x3(const ConvArgs& a, const int* dils, const void* const* w, const float* const* bias, hipStream_t s);

// ---- fp32 polyphase transposed convolution on MFMA (k == 2*stride) -------------------------
bool convT1d_f32_mfma_supported(int Cin, int Cout, int K, int stride, int pad_a, int L);
size_t convT1d_f32_mfma_packed_floats(int Cin, int Cout, int K);
void convT1d_f32_mfma_pack(const float* w_hk, int Cin, int Cout, int K, int stride, int pad_a, float* out);
hipError_t launch_convT1d_f32_mfma(const ConvArgs& a, hipStream_t s);

// ---- streaming conv_post (32 -> 1, K = 7, fused LeakyReLU + tanh) ----------------------------
bool conv_post_fast_supported(int Cin, int Cout, int K, int L);
hipError_t launch_conv_post_fast(const ConvArgs& a, hipStream_t s);

// =================================================================================================
// bf16 path: kernels_bf16.hip  (channels-last activations, v_mfma_f32_32x32x16_bf16)
// =================================================================================================
enum BClass : int { BCLS_NONE = -1, BCLS_RES256 = 0, BCLS_RES128, BCLS_RES64, BCLS_RES32, BCLS_PRE, BCLS_UP0, BCLS_UP1, BCLS_UP2, BCLS_UP3 };

struct BConvArgs {
    const void* x;        // input activations: bf16 [B][L][x_pitch]  (fp32 for conv_pre)
    const void* wp;       // packed bf16 weights (A-fragment-ordered slabs)
    const float* wf;      // plain fp32 weights (conv_post only)
    const float* bias;    // fp32 [COUTP]
    const void* res;      // optional residual, bf16 [B][L][COUTP], or nullptr
    void* y;              // output bf16 [B][L][COUTP]
    int B, L;             // utterances; rows (time steps) ALLOCATED per utterance in x / y
    const int* lens;      // ragged batch: valid mel frames per utterance (device, [B]) or nullptr = all L rows valid
    int len_mul;          // rows of this layer per mel frame (valid rows of utterance b = lens[b] * len_mul)
    int x_pitch;          // elements per input row in global memory
    int cin_real;         // valid input channels (conv_pre: 80)
    int dil, pad;
    int convt_halves;     // conv_bf16_k on a transposed convolution's 3-tap form: skip each row half's all-zero tap
    int dils[3];          // whole-ResBlock kernel: the three pairs' rates
    float slope_in;       // LeakyReLU applied to the input while staging (1 = producer already did it)
    float slope_out;      // LeakyReLU applied to the stored output (the consumer's activation), 1 = none
    int acc_add;          // y = y + v   (MRF accumulate)
    int tile_pref;        // fused-pair tile width: 0 = by launch size, 1 = wide, 2 = narrow (option "tiles"; tests run both on short inputs)
    float div;            // then the MRF mean v / div as v * (1 / div) (bf16_common.h: mrf_recip), 1 = none
    int zrev;             // 1 = utterance = gridDim.z - 1 - blockIdx.z: the launch walks the batch backwards (engine.hip: next_zrev)
    unsigned long long* dbg;  // kernel-development builds only (-DVTTS_TIMELINE): per-workgroup s_memtime stamps; else unused
    // stage-4 tail fused into the last pair launch (kernels_bf16_rbg.hip: GTail): conv_post's plain fp32 weights [7][32] and bias, the waveform
    // [B][L] fp32 (pitch = this layer's L); nullptr = a plain pair launch
    float* tail_wav;
    const float* tail_wf;
    const float* tail_bias;
};

struct BPackGeom { int cinp, ckc, coutp, ks, mt, tg; };

hipError_t launch_conv_bf16(int cls, int K, const BConvArgs& a, hipStream_t s);
BPackGeom bf16_pack_geom(int cls, int K);
const char* bf16_kernel_name(int cls, int K);
size_t bf16_packed_bytes(const BPackGeom& g);
void bf16_pack(const float* Wc, int cin_real, const BPackGeom& g, unsigned short* out);
// fused ResBlock1 pair (c1 -> lrelu -> c2 -> + x): kernels_bf16_rbg.hip.  a.wp = [c1 fragments][c2 fragments],
// a.bias = [b1 (C)][b2 (C)], a.x = raw pair input (also the residual), a.dil/a.pad = c1's rate / pad.
bool pair_bf16_supported(int C, int K, int dil);
BPackGeom pair_pack_geom(int C, int K);
hipError_t launch_pair_bf16(int C, int K, const BConvArgs& a, hipStream_t s);
bool pair_tail_bf16_supported(int C, int K, int post_cin, int post_cout, int post_k);
const char* pair_kernel_name(int C, int K);
// kernels_bf16_rbg.hip: weights straight from L2 into register rings, no workgroup sync in the main loops
hipError_t launch_pair_g_bf16(int C, int K, const BConvArgs& a, hipStream_t s);
BPackGeom pair_g_pack_geom(int C, int K);
const char* pair_g_kernel_name(int C, int K);
// whole ResBlock1 in one kernel (kernels_bf16_rbk.hip): C = 32 (k = 3, 7, 11), C = 64 / 128 (k = 3)
bool resblock_bf16_supported(int C, int K, const int* dils);
bool resblock_bf16_preferred(int C, int K);
hipError_t launch_resblock_bf16(int C, int K, const BConvArgs& a, hipStream_t s);
const char* resblock_kernel_name(int C, int K);
// the generator's whole LAST stage — three ResBlock1 of C = 32 from one LDS-resident window, the MRF mean, LeakyReLU(0.01), conv_post, tanh — in one
// launch (kernels_bf16_stage.hip), bit-identical to the launch-per-ResBlock path
struct BStageArgs {
    const void* x;            // stage input (ups_3's output), raw bf16 [B][L][32]
    const void* wp[3];        // per ResBlock: its six convolutions' A fragments (Layer::off_rw: pair_g_pack_geom order, contiguous)
    const float* bias[3];     // per ResBlock: 6 x [32] fp32 (Layer::off_rb)
    int dils[3][3];           // per ResBlock: the three pairs' rates
    void* s;                  // scratch bf16 [B][L][32]: the MRF sum across the third ResBlock (only the rows a window reads back are written)
    float* wav;               // [B][L] fp32
    const float* post_w;      // conv_post's plain fp32 weights [7][32] (Haiku [K][Cin][1]) and bias
    const float* post_b;
    int B, L;                 // utterances; rows allocated per utterance
    const int* lens;          // ragged batch (as BConvArgs)
    int len_mul;
    int zrev;
    int margin;               // stage_margin(): rows per window side whose dependency cone leaves the window (+ conv_post's 3)
    float div, slope_out;     // MRF divisor (num_kernels), the tail's LeakyReLU slope (model.py:121-122)
    unsigned long long* dbg;  // kernel-development builds only (-DVTTS_TIMELINE=1, VTTS_ST_TL=<file>): per-workgroup s_memtime stamps; nullptr otherwise
};
bool stage_bf16_supported(int C, int nk, const int* ks, const int (*dils)[3], int post_cin, int post_cout, int post_k);
int stage_margin(const int* ks, const int (*dils)[3]);
hipError_t launch_stage_bf16(const BStageArgs& a, hipStream_t s);
const char* stage_kernel_name();
// the four transposed convolutions on the register-streamed structure (kernels_bf16_up.hip): cls = BCLS_UP0..3,
// a.wp = the 3-tap polyphase form's weights packed with convt_g_pack_geom(cls) (bf16_pack)
hipError_t launch_convt_g_bf16(int cls, const BConvArgs& a, hipStream_t s);
BPackGeom convt_g_pack_geom(int cls);
const char* convt_g_kernel_name(int cls);
hipError_t launch_conv_post_bf16(const BConvArgs& a, float* wav, float* pre_act, hipStream_t s);
hipError_t launch_bf16_to_f32(const void* in, float* out, size_t n, hipStream_t s);
hipError_t launch_f32_to_bf16(const float* in, void* out, size_t n, hipStream_t s);

}  // namespace vtts
