// Device-side helpers shared by all kernels (gfx950).
#pragma once

#include "vtts_internal.h"

namespace vtts {

__device__ __forceinline__ float lrelu(float v, float slope) {
    // jax.nn.leaky_relu / F.leaky_relu: where(x >= 0, x, slope * x)  (model.py:46,48,112,122)
    return v >= 0.0f ? v : v * slope;
}

// Valid input length of utterance b (ragged batches: ConvArgs::lens; clamped into the slot, so a bad count never leaves it) — a.L otherwise.
__device__ __forceinline__ int valid_len(const ConvArgs& a, int b) {
    return a.lens ? min(max(a.lens[b], 0) * a.len_mul, a.L) : a.L;
}
// Four consecutive columns t .. t + 3 loaded as one float4 from a row whose valid length L is not a multiple of 4 (a ragged utterance's
// conv_pre output): the columns past L read as zero (select, not multiply: the slot's tail holds whatever the last pass left there).
__device__ __forceinline__ float4 mask_tail4(float4 v, int t, int L) {
    if (t + 1 >= L) v.y = 0.0f;
    if (t + 2 >= L) v.z = 0.0f;
    if (t + 3 >= L) v.w = 0.0f;
    return v;
}

// Combine a finished convolution value with memory according to ConvArgs::acc_mode and store it.
// `v` already holds acc + bias.  Order of operations follows the reference:
//   residual   xt + x            (model.py:50)
//   MRF        xs += rb(x)       (model.py:118-120), x = xs / num_kernels (model.py:121)
//   tail       tanh(conv_post)   (model.py:123-124)
__device__ __forceinline__ void epilogue_store(const ConvArgs& a, long idx, float v) {
    if (a.res) v = v + a.res[idx];
    if (a.acc_mode == ACC_ADD) {
        v = a.y[idx] + v;
    } else if (a.acc_mode == ACC_MEAN) {
        v = (a.y[idx] + v) / a.div;
    }
    if (a.tanh_out) {
        if (a.pre_act) a.pre_act[idx] = v;
        v = tanhf(v);
    }
    a.y[idx] = v;
}

}  // namespace vtts
