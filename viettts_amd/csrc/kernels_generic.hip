// Generic fp32 convolution kernels: any channel count, kernel size, dilation, stride.
//
// These are the shape-agnostic HIP path (VALU fmaf chains, weights read wave-uniformly through
// the scalar cache).  They carry conv_pre (80->512, K=7: 0.09 % of the FLOPs), conv_post
// (32->1 + tanh: 0.02 %), architectures the MFMA kernels have no instantiation for (e.g. the
// TINY fixture config), and serve as the on-device cross-check of the MFMA kernels.
//
// Reference semantics: hk.Conv1D / hk.Conv1DTranspose as restated in SURVEY.md Appendix A.1/A.2
// (call sites vietTTS/hifigan/model.py:83,88-94,107; :21-28,33-40).
#include "device_common.h"

namespace vtts {

// One thread = one output time step x COT consecutive output channels.
// grid = (ceil(Lout/256), ceil(Cout/COT), B)
template <int COT>
__global__ __launch_bounds__(256) void conv1d_generic_k(ConvArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT;
    const int b = blockIdx.z;
    const float* __restrict__ xb = a.x + (long)b * a.x_sb;
    const int Lv = valid_len(a, b);  // ragged batches: this utterance's columns (a.L = a.Lout stay the row pitch)

    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = 0.0f;

    for (int j = 0; j < a.K; ++j) {
        const int ti = t + j * a.dil - a.pad;
        const bool ok = (t < Lv) && (ti >= 0) && (ti < Lv);
        const float* xp = xb + (long)ti * a.x_st;
        const float* wj = a.w + (long)j * a.Cin * a.Cout + co0;
        for (int ci = 0; ci < a.Cin; ++ci) {
            float v = ok ? xp[(long)ci * a.x_sc] : 0.0f;
            v = lrelu(v, a.slope_in);
            const float* wr = wj + (long)ci * a.Cout;  // wave-uniform address
#pragma unroll
            for (int c = 0; c < COT; ++c) {
                const float wv = (co0 + c < a.Cout) ? wr[c] : 0.0f;
                acc[c] = fmaf(wv, v, acc[c]);
            }
        }
    }
    if (t >= Lv) return;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        const int co = co0 + c;
        if (co < a.Cout) {
            const long idx = ((long)b * a.Cout + co) * a.Lout + t;
            epilogue_store(a, idx, acc[c] + a.bias[co]);
        }
    }
}

hipError_t launch_conv1d_generic(const ConvArgs& a, hipStream_t s) {
    if (a.Cout >= 4) {
        dim3 grid((a.Lout + 255) / 256, (a.Cout + 3) / 4, a.B);
        hipLaunchKernelGGL(conv1d_generic_k<4>, grid, dim3(256), 0, s, a);
    } else {
        dim3 grid((a.Lout + 255) / 256, a.Cout, a.B);
        hipLaunchKernelGGL(conv1d_generic_k<1>, grid, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// Transposed convolution in polyphase form.  With the zero-stuffed,
// (pad_a, pad_b)-padded input xd (xd[pad_a + s*t] = x[t]) and y[p] = b + sum_j w[j] . xd[p + j],
// output p = s*q + r only meets taps j == (pad_a - r) mod s, i.e. j = j0 + m*s, reading input
// frame t = (p + j - pad_a)/s.  One block handles one phase r (wave-uniform tap set).
// grid = (ceil(L/256), ceil(Cout/COT) * stride, B); thread = one q.
template <int COT>
__global__ __launch_bounds__(256) void convT1d_generic_k(ConvArgs a) {
    const int s = a.stride;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y % s;
    const int co0 = (blockIdx.y / s) * COT;
    const int b = blockIdx.z;
    const float* __restrict__ xb = a.x + (long)b * a.x_sb;
    const int Lv = valid_len(a, b);  // ragged batches: this utterance's input frames

    float acc[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c] = 0.0f;

    int j0 = (a.pad_a - r) % s;
    if (j0 < 0) j0 += s;
    for (int j = j0; j < a.K; j += s) {
        const int ti = q + (r + j - a.pad_a) / s;  // exact: r + j - pad_a is a multiple of s
        const bool ok = (q < Lv) && (ti >= 0) && (ti < Lv);
        const float* xp = xb + (long)ti * a.x_st;
        const float* wj = a.w + ((long)j * a.Cout + co0) * a.Cin;  // [K][Cout][Cin]
        for (int ci = 0; ci < a.Cin; ++ci) {
            float v = ok ? xp[(long)ci * a.x_sc] : 0.0f;
            v = lrelu(v, a.slope_in);
#pragma unroll
            for (int c = 0; c < COT; ++c) {
                const float wv = (co0 + c < a.Cout) ? wj[(long)c * a.Cin + ci] : 0.0f;
                acc[c] = fmaf(wv, v, acc[c]);
            }
        }
    }
    if (q >= Lv) return;
    const int p = q * s + r;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        const int co = co0 + c;
        if (co < a.Cout) {
            const long idx = ((long)b * a.Cout + co) * a.Lout + p;
            epilogue_store(a, idx, acc[c] + a.bias[co]);
        }
    }
}

hipError_t launch_convT1d_generic(const ConvArgs& a, hipStream_t s) {
    dim3 grid((a.L + 255) / 256, ((a.Cout + 3) / 4) * a.stride, a.B);
    hipLaunchKernelGGL(convT1d_generic_k<4>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace vtts
