// NAT duration model on MI355X behind the C ABI of include/vtts_nat.h.
//
// Reference: vietTTS/nat/model.py — TokenEncoder (:9-50: Embed, 3 x [Conv1D(k=3, SAME) + BatchNorm(eval) + ReLU],
// forward LSTM, backward LSTM) and DurationModel (:53-70: Linear -> gelu -> Linear(1) -> softplus), called with batch 1
// by text2mel.py:22-34.  All arithmetic fp32 (the reference's dtype); one sentence per workgroup row, rows independent.
//
// This is a latency path (a sentence is ~100 tokens x 256 channels), not an MFMA path: the recurrence is sequential in
// time and each step is a [1 x 512] x [512 x 1024] product.  Mapping:
//   * front end: one kernel per layer; a workgroup owns TL time steps x all D output channels of one sentence, the
//     (TL + 2) input rows staged in LDS, weights read coalesced along the output channel;
//   * LSTM: one persistent workgroup per (sentence, direction), one thread per gate column (4D = 1024 threads); per
//     step every thread walks its column of the [2D x 4D] weight matrix (coalesced across threads, L2-resident: all
//     workgroups read the same 2 MB), [x_t ; h] broadcast from LDS, cell state in registers of the first D threads;
//   * head: Linear(2D -> D) + tanh-form gelu + Linear(D -> 1) + softplus per token, block reduction for the last dot.
#include "../../include/vtts_nat.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vtts_hifigan.h"
#include "vtts_internal.h"

#define VTTS_API extern "C" __attribute__((visibility("default")))

namespace {

int failf(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return vtts::set_error(code, buf);
}

#define HIP_TRYN(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return failf(VTTS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct Arr {
    std::string module, name;
    std::vector<int64_t> shape;
    std::vector<float> host;
    bool have = false;
    size_t off = 0;  // byte offset in the packed blob
    size_t elems() const {
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        return n;
    }
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct vtts_nat_duration {
    vtts_nat_duration_cfg cfg;
    int device = 0;
    std::vector<Arr> arrs;
    size_t blob_bytes = 0;
    char* blob = nullptr;
    int find(const char* module, const char* name) const {
        for (size_t i = 0; i < arrs.size(); ++i)
            if (arrs[i].module == module && arrs[i].name == name) return (int)i;
        return -1;
    }
    const float* dev(const char* module, const char* name) const { return reinterpret_cast<const float*>(blob + arrs[find(module, name)].off); }
};

// ================================================ kernels ================================================
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// x0[b][t][:] = embeddings[tokens[b][t]][:]   (model.py:27); rows past the sentence's length are zero
__global__ void nat_embed_k(const int* __restrict__ tokens, const int* __restrict__ lengths, const float* __restrict__ emb,
                            float* __restrict__ out, int Lmax, int D, int V) {
    const int b = blockIdx.y, t = blockIdx.x;
    const int len = lengths[b];
    int tok = tokens[(size_t)b * Lmax + t];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    for (int c = threadIdx.x; c < D; c += blockDim.x) out[((size_t)b * Lmax + t) * D + c] = t < len ? emb[(size_t)tok * D + c] : 0.0f;
}

// y = relu(batchnorm_eval(conv1d_same_k3(x)))   (model.py:28, :31, :34; hk.Conv1D w[3][D][D], cross-correlation)
// inv = scale * rsqrt(var + eps) is precomputed at pack time; rows at or past the sentence's length read as zero
// (the reference runs each sentence alone, so its SAME padding sees zeros there) and are written as zero.
template <int TL>
__global__ __launch_bounds__(256) void nat_conv3_bn_relu_k(const float* __restrict__ x, const int* __restrict__ lengths,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ inv, const float* __restrict__ mean,
                                                           const float* __restrict__ offset, float* __restrict__ y, int Lmax, int D) {
    extern __shared__ float xs[];  // (TL + 2) x D
    const int b = blockIdx.y, t0 = blockIdx.x * TL;
    const int len = lengths[b];
    for (int i = threadIdx.x; i < (TL + 2) * D; i += blockDim.x) {
        const int r = i / D, c = i % D;
        const int t = t0 - 1 + r;
        xs[i] = (t >= 0 && t < len) ? x[((size_t)b * Lmax + t) * D + c] : 0.0f;
    }
    __syncthreads();
    for (int co = threadIdx.x; co < D; co += blockDim.x) {
        float acc[TL];
        const float bv = bias[co];
#pragma unroll
        for (int i = 0; i < TL; ++i) acc[i] = bv;
        for (int j = 0; j < 3; ++j) {
            const float* __restrict__ wj = w + (size_t)j * D * D + co;
            for (int ci = 0; ci < D; ++ci) {
                const float wv = wj[(size_t)ci * D];
#pragma unroll
                for (int i = 0; i < TL; ++i) acc[i] = fmaf(xs[(i + j) * D + ci], wv, acc[i]);
            }
        }
        const float iv = inv[co], mv = mean[co], ov = offset[co];
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            const int t = t0 + i;
            if (t < Lmax) {
                const float v = (acc[i] - mv) * iv + ov;
                y[((size_t)b * Lmax + t) * D + co] = t < len ? fmaxf(v, 0.0f) : 0.0f;
            }
        }
    }
}

// hk.LSTM over one sentence in one direction (model.py:39-45).  grid = (B, 2); blockDim = 4*D (one thread per gate
// column, order i, g, f, o; forget gate +1).  w [2D][4D] (rows: x then h), b [4D].  out[b][t][dir*D + j] = h_t[j]; the
// backward direction walks t = len-1 .. 0 and stores at t, which IS jnp.flip of its outputs (:46).  hk.ResetCore's
// reset falls on the backward pass's first step(s), where the state still is the initial state.
__global__ __launch_bounds__(1024) void nat_lstm_k(const float* __restrict__ x, const int* __restrict__ lengths, const float* __restrict__ wf,
                           const float* __restrict__ bf, const float* __restrict__ wb, const float* __restrict__ bb,
                           float* __restrict__ out, int Lmax, int D) {
    extern __shared__ float sm[];  // xh[2D], gates[4D]
    float* xh = sm;
    float* gates = sm + 2 * D;
    const int b = blockIdx.x, dir = blockIdx.y;
    const int g = threadIdx.x;  // gate column
    const int len = lengths[b];
    const float* __restrict__ w = dir ? wb : wf;
    const float bias = (dir ? bb : bf)[g];
    float c = 0.0f;
    if (g < D) xh[D + g] = 0.0f;
    for (int s = 0; s < len; ++s) {
        const int t = dir ? len - 1 - s : s;
        if (g < D) xh[g] = x[((size_t)b * Lmax + t) * D + g];
        __syncthreads();
        float acc = bias;
        const float* __restrict__ wc = w + g;
        const int W4 = 4 * D;
#pragma unroll 8
        for (int k = 0; k < 2 * D; ++k) acc = fmaf(xh[k], wc[(size_t)k * W4], acc);
        gates[g] = acc;
        __syncthreads();
        if (g < D) {
            const float gi = gates[g], gg = gates[D + g], gf = gates[2 * D + g], go = gates[3 * D + g];
            c = sigmoidf_(gf + 1.0f) * c + sigmoidf_(gi) * tanhf(gg);
            const float h = sigmoidf_(go) * tanhf(c);
            xh[D + g] = h;
            out[((size_t)b * Lmax + t) * (2 * D) + dir * D + g] = h;
        }
        // the next step's barrier orders these writes before the next reads of xh[D..] / gates
    }
}

// durations = softplus(Linear(D->1)(gelu(Linear(2D->D)(enc))))   (model.py:64-70); blockDim = D, one token per block
__global__ void nat_duration_head_k(const float* __restrict__ enc, const int* __restrict__ lengths, const float* __restrict__ w1,
                                    const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                    float* __restrict__ dur, int Lmax, int D) {
    extern __shared__ float sh[];  // enc row [2D], partial sums [blockDim/64]
    float* e = sh;
    float* part = sh + 2 * D;
    const int b = blockIdx.y, t = blockIdx.x, j = threadIdx.x;
    const int len = lengths[b];
    if (t >= len) {
        if (j == 0) dur[(size_t)b * Lmax + t] = 0.0f;
        return;
    }
    for (int i = j; i < 2 * D; i += blockDim.x) e[i] = enc[((size_t)b * Lmax + t) * (2 * D) + i];
    __syncthreads();
    float acc = b1[j];
    for (int k = 0; k < 2 * D; ++k) acc = fmaf(e[k], w1[(size_t)k * D + j], acc);
    // jax.nn.gelu(approximate=True)
    const float u = 0.7978845608028654f * (acc + 0.044715f * acc * acc * acc);
    float v = 0.5f * acc * (1.0f + tanhf(u)) * w2[j];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((j & 63) == 0) part[j >> 6] = v;
    __syncthreads();
    if (j == 0) {
        float s = b2[0];
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i];
        // jax.nn.softplus = logaddexp(s, 0) = max(s, 0) + log1p(exp(-|s|))
        dur[(size_t)b * Lmax + t] = fmaxf(s, 0.0f) + log1pf(expf(-fabsf(s)));
    }
}

}  // namespace

// ================================================ C ABI ================================================
VTTS_API int vtts_nat_duration_create(const vtts_nat_duration_cfg* cfg, int device, vtts_nat_duration** out) {
    if (!cfg || !out) return failf(VTTS_ERR_INVALID, "null argument");
    const int D = cfg->lstm_dim, V = cfg->vocab_size;
    if (D < 64 || D > 256 || D % 64 != 0 || V < 1)
        return failf(VTTS_ERR_INVALID, "duration model: lstm_dim must be 64, 128, 192 or 256 (one thread per gate column, 4*dim <= 1024) and vocab_size >= 1 (got %d, %d)", D, V);
    auto* h = new (std::nothrow) vtts_nat_duration();
    if (!h) return failf(VTTS_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->device = device;
    auto add = [&](const std::string& m, const char* n, std::vector<int64_t> shp) {
        Arr a;
        a.module = m;
        a.name = n;
        a.shape = std::move(shp);
        h->arrs.push_back(a);
    };
    const std::string te = "token_encoder/~/";
    add(te + "embed", "embeddings", {V, D});
    for (int i = 0; i < 3; ++i) {
        const std::string sfx = i ? "_" + std::to_string(i) : "";
        add(te + "conv1_d" + sfx, "w", {3, D, D});
        add(te + "conv1_d" + sfx, "b", {D});
        add(te + "batch_norm" + sfx, "scale", {1, 1, D});
        add(te + "batch_norm" + sfx, "offset", {1, 1, D});
        add(te + "batch_norm" + sfx + "/~/mean_ema", "average", {1, 1, D});
        add(te + "batch_norm" + sfx + "/~/var_ema", "average", {1, 1, D});
    }
    for (const char* l : {"lstm/linear", "lstm_1/linear"}) {
        add(te + l, "w", {2 * D, 4 * D});
        add(te + l, "b", {4 * D});
    }
    add("linear", "w", {2 * D, D});
    add("linear", "b", {D});
    add("linear_1", "w", {D, 1});
    add("linear_1", "b", {1});
    size_t off = 0;
    for (auto& a : h->arrs) {
        a.off = off;
        off = align_up(off + a.elems() * sizeof(float), 256);
    }
    // three derived arrays: inv[i] = scale * rsqrt(var + eps), appended after the checkpoint arrays
    h->blob_bytes = off + 3 * align_up((size_t)D * sizeof(float), 256);
    *out = h;
    return VTTS_OK;
}

VTTS_API void vtts_nat_duration_destroy(vtts_nat_duration* h) { delete h; }

VTTS_API int vtts_nat_duration_num_params(const vtts_nat_duration* h, int* n) {
    if (!h || !n) return failf(VTTS_ERR_INVALID, "null argument");
    *n = (int)h->arrs.size();
    return VTTS_OK;
}

VTTS_API int vtts_nat_duration_param_info(const vtts_nat_duration* h, int i, const char** module, const char** name, int64_t shape[3],
                                          int* ndim) {
    if (!h || i < 0 || i >= (int)h->arrs.size()) return failf(VTTS_ERR_INVALID, "parameter index out of range");
    const Arr& a = h->arrs[i];
    if (module) *module = a.module.c_str();
    if (name) *name = a.name.c_str();
    if (shape)
        for (int d = 0; d < 3; ++d) shape[d] = d < (int)a.shape.size() ? a.shape[d] : 1;
    if (ndim) *ndim = (int)a.shape.size();
    return VTTS_OK;
}

VTTS_API int vtts_nat_duration_set_param(vtts_nat_duration* h, const char* module, const char* name, const float* host,
                                         const int64_t* shape, int ndim) {
    if (!h || !module || !name || !host || !shape) return failf(VTTS_ERR_INVALID, "null argument");
    const int i = h->find(module, name);
    if (i < 0) return failf(VTTS_ERR_INVALID, "duration model has no array '%s' in module '%s'", name, module);
    Arr& a = h->arrs[i];
    if (ndim != (int)a.shape.size()) return failf(VTTS_ERR_SHAPE, "%s/%s: expected %zu dimensions, got %d", module, name, a.shape.size(), ndim);
    for (int d = 0; d < ndim; ++d)
        if (shape[d] != a.shape[d]) return failf(VTTS_ERR_SHAPE, "%s/%s: dimension %d is %lld, expected %lld", module, name, d, (long long)shape[d], (long long)a.shape[d]);
    a.host.assign(host, host + a.elems());
    a.have = true;
    return VTTS_OK;
}

VTTS_API int vtts_nat_duration_packed_bytes(const vtts_nat_duration* h, size_t* bytes) {
    if (!h || !bytes) return failf(VTTS_ERR_INVALID, "null argument");
    *bytes = h->blob_bytes;
    return VTTS_OK;
}

VTTS_API int vtts_nat_duration_pack(vtts_nat_duration* h, void* dev_blob, size_t blob_bytes, void* stream) {
    if (!h || !dev_blob) return failf(VTTS_ERR_INVALID, "null argument");
    if (blob_bytes < h->blob_bytes) return failf(VTTS_ERR_NOMEM, "blob too small: %zu < %zu bytes", blob_bytes, h->blob_bytes);
    for (auto& a : h->arrs)
        if (!a.have) return failf(VTTS_ERR_MISSING, "array %s/%s was never set", a.module.c_str(), a.name.c_str());
    std::vector<char> img(h->blob_bytes, 0);
    size_t last = 0;
    for (auto& a : h->arrs) {
        memcpy(img.data() + a.off, a.host.data(), a.elems() * sizeof(float));
        last = align_up(a.off + a.elems() * sizeof(float), 256);
    }
    const int D = h->cfg.lstm_dim;
    for (int i = 0; i < 3; ++i) {
        const std::string sfx = i ? "_" + std::to_string(i) : "";
        const Arr& sc = h->arrs[h->find(("token_encoder/~/batch_norm" + sfx).c_str(), "scale")];
        const Arr& var = h->arrs[h->find(("token_encoder/~/batch_norm" + sfx + "/~/var_ema").c_str(), "average")];
        float* inv = reinterpret_cast<float*>(img.data() + last + (size_t)i * align_up((size_t)D * sizeof(float), 256));
        for (int c = 0; c < D; ++c) inv[c] = sc.host[c] / std::sqrt(var.host[c] + 1e-5f);  // hk.BatchNorm eps
    }
    HIP_TRYN(hipMemcpyAsync(dev_blob, img.data(), h->blob_bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    HIP_TRYN(hipStreamSynchronize(static_cast<hipStream_t>(stream)));  // img is a stack-lifetime staging buffer
    h->blob = static_cast<char*>(dev_blob);
    return VTTS_OK;
}

VTTS_API int vtts_nat_duration_bind_packed(vtts_nat_duration* h, void* dev_blob, size_t blob_bytes) {
    if (!h || !dev_blob) return failf(VTTS_ERR_INVALID, "null argument");
    if (blob_bytes < h->blob_bytes) return failf(VTTS_ERR_NOMEM, "blob too small: %zu < %zu bytes", blob_bytes, h->blob_bytes);
    h->blob = static_cast<char*>(dev_blob);
    return VTTS_OK;
}

VTTS_API int vtts_nat_duration_workspace_bytes(const vtts_nat_duration* h, int B, int Lmax, size_t* bytes) {
    if (!h || !bytes) return failf(VTTS_ERR_INVALID, "null argument");
    if (B <= 0 || Lmax <= 0) return failf(VTTS_ERR_INVALID, "B and Lmax must be positive (got %d, %d)", B, Lmax);
    const size_t D = h->cfg.lstm_dim;
    // two ping-pong [B][Lmax][D] buffers + the encoder output [B][Lmax][2D]
    *bytes = 2 * align_up((size_t)B * Lmax * D * 4, 256) + align_up((size_t)B * Lmax * 2 * D * 4, 256);
    return VTTS_OK;
}

VTTS_API int vtts_nat_duration_forward(vtts_nat_duration* h, const int32_t* tokens_dev, const int32_t* lengths_dev, int B, int Lmax,
                                       float* durations_dev, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !tokens_dev || !lengths_dev || !durations_dev) return failf(VTTS_ERR_INVALID, "null argument");
    if (!h->blob) return failf(VTTS_ERR_STATE, "forward() before pack()/bind_packed()");
    size_t need = 0;
    int rc = vtts_nat_duration_workspace_bytes(h, B, Lmax, &need);
    if (rc) return rc;
    if (!workspace || workspace_bytes < need) return failf(VTTS_ERR_NOMEM, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int D = h->cfg.lstm_dim, V = h->cfg.vocab_size;
    const size_t per = align_up((size_t)B * Lmax * D * 4, 256);
    float* bufA = reinterpret_cast<float*>(static_cast<char*>(workspace));
    float* bufB = reinterpret_cast<float*>(static_cast<char*>(workspace) + per);
    float* enc = reinterpret_cast<float*>(static_cast<char*>(workspace) + 2 * per);
    const size_t inv_base = h->blob_bytes - 3 * align_up((size_t)D * sizeof(float), 256);

    hipLaunchKernelGGL(nat_embed_k, dim3(Lmax, B), dim3(256), 0, s, tokens_dev, lengths_dev, h->dev("token_encoder/~/embed", "embeddings"), bufA,
                       Lmax, D, V);
    constexpr int TL = 8;
    float* cur = bufA;
    float* nxt = bufB;
    for (int i = 0; i < 3; ++i) {
        const std::string sfx = i ? "_" + std::to_string(i) : "";
        const std::string cv = "token_encoder/~/conv1_d" + sfx, bn = "token_encoder/~/batch_norm" + sfx;
        const float* inv = reinterpret_cast<const float*>(h->blob + inv_base + (size_t)i * align_up((size_t)D * sizeof(float), 256));
        hipLaunchKernelGGL(nat_conv3_bn_relu_k<TL>, dim3((Lmax + TL - 1) / TL, B), dim3(256), (TL + 2) * D * sizeof(float), s, cur, lengths_dev,
                           h->dev(cv.c_str(), "w"), h->dev(cv.c_str(), "b"), inv, h->dev((bn + "/~/mean_ema").c_str(), "average"),
                           h->dev(bn.c_str(), "offset"), nxt, Lmax, D);
        std::swap(cur, nxt);
    }
    hipLaunchKernelGGL(nat_lstm_k, dim3(B, 2), dim3(4 * D), 6 * D * sizeof(float), s, cur, lengths_dev, h->dev("token_encoder/~/lstm/linear", "w"),
                       h->dev("token_encoder/~/lstm/linear", "b"), h->dev("token_encoder/~/lstm_1/linear", "w"),
                       h->dev("token_encoder/~/lstm_1/linear", "b"), enc, Lmax, D);
    hipLaunchKernelGGL(nat_duration_head_k, dim3(Lmax, B), dim3(D), (2 * D + 16) * sizeof(float), s, enc, lengths_dev, h->dev("linear", "w"),
                       h->dev("linear", "b"), h->dev("linear_1", "w"), h->dev("linear_1", "b"), durations_dev, Lmax, D);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "duration model launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}
