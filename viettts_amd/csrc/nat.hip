// NAT duration and acoustic models on MI355X behind the C ABI of include/vtts_nat.h.
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
// Acoustic model (model.py:73-151, inference path): the same TokenEncoder, Gaussian upsampling to frames (one block per
// frame), the autoregressive decoder as ONE persistent 1024-thread workgroup per sentence (prenet, two LSTM-512 with skip
// connections, mel projection; every thread owns two gate columns; the 25 MB of weights stream from L2 / Infinity Cache each
// frame — a functional first cut: the weights-stationary form, gate columns spread over all CUs with a grid barrier per
// frame, is what comes next), then the 5-layer postnet with the generic Conv1D + BatchNorm + activation kernel.
// The prenet's always-on dropout (model.py:95-100) takes explicit keep masks: JAX's threefry stream is not restated.
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

// Arrays a model takes from the checkpoint (Haiku module tail + array name), their place in the packed device blob, and
// the derived per-BatchNorm vectors inv = scale * rsqrt(var + eps) appended behind them.
struct NatModel {
    const char* what = "model";
    int device = 0;
    std::vector<Arr> arrs;
    std::vector<std::pair<std::string, int>> bns;  // (BatchNorm module tail, channels)
    std::vector<size_t> bn_off;
    size_t blob_bytes = 0;
    char* blob = nullptr;

    void add(const std::string& m, const char* n, std::vector<int64_t> shp) {
        Arr a;
        a.module = m;
        a.name = n;
        a.shape = std::move(shp);
        arrs.push_back(a);
    }
    void add_bn(const std::string& m, int C) {
        add(m, "scale", {1, 1, C});
        add(m, "offset", {1, 1, C});
        add(m + "/~/mean_ema", "average", {1, 1, C});
        add(m + "/~/var_ema", "average", {1, 1, C});
        bns.emplace_back(m, C);
    }
    // TokenEncoder (model.py:12-24): Embed, 3 x (Conv1D k=3 + BatchNorm), forward LSTM, backward LSTM
    void add_token_encoder(const std::string& te, int V, int D) {
        add(te + "embed", "embeddings", {V, D});
        for (int i = 0; i < 3; ++i) {
            const std::string sfx = i ? "_" + std::to_string(i) : "";
            add(te + "conv1_d" + sfx, "w", {3, D, D});
            add(te + "conv1_d" + sfx, "b", {D});
            add_bn(te + "batch_norm" + sfx, D);
        }
        for (const char* l : {"lstm/linear", "lstm_1/linear"}) {
            add(te + l, "w", {2 * D, 4 * D});
            add(te + l, "b", {4 * D});
        }
    }
    void layout() {
        size_t off = 0;
        for (auto& a : arrs) {
            a.off = off;
            off = align_up(off + a.elems() * sizeof(float), 256);
        }
        for (auto& b : bns) {
            bn_off.push_back(off);
            off = align_up(off + (size_t)b.second * sizeof(float), 256);
        }
        blob_bytes = off;
    }
    int find(const std::string& module, const char* name) const {
        for (size_t i = 0; i < arrs.size(); ++i)
            if (arrs[i].module == module && arrs[i].name == name) return (int)i;
        return -1;
    }
    const float* dev(const std::string& module, const char* name) const { return reinterpret_cast<const float*>(blob + arrs[find(module, name)].off); }
    const float* inv(const std::string& bn) const {
        for (size_t i = 0; i < bns.size(); ++i)
            if (bns[i].first == bn) return reinterpret_cast<const float*>(blob + bn_off[i]);
        return nullptr;
    }

    int param_info(int i, const char** module, const char** name, int64_t shape[3], int* ndim) const {
        if (i < 0 || i >= (int)arrs.size()) return failf(VTTS_ERR_INVALID, "parameter index out of range");
        const Arr& a = arrs[i];
        if (module) *module = a.module.c_str();
        if (name) *name = a.name.c_str();
        if (shape)
            for (int d = 0; d < 3; ++d) shape[d] = d < (int)a.shape.size() ? a.shape[d] : 1;
        if (ndim) *ndim = (int)a.shape.size();
        return VTTS_OK;
    }
    int set_param(const char* module, const char* name, const float* host, const int64_t* shape, int ndim) {
        if (!module || !name || !host || !shape) return failf(VTTS_ERR_INVALID, "null argument");
        const int i = find(module, name);
        if (i < 0) return failf(VTTS_ERR_INVALID, "%s has no array '%s' in module '%s'", what, name, module);
        Arr& a = arrs[i];
        if (ndim != (int)a.shape.size()) return failf(VTTS_ERR_SHAPE, "%s/%s: expected %zu dimensions, got %d", module, name, a.shape.size(), ndim);
        for (int d = 0; d < ndim; ++d)
            if (shape[d] != a.shape[d])
                return failf(VTTS_ERR_SHAPE, "%s/%s: dimension %d is %lld, expected %lld", module, name, d, (long long)shape[d], (long long)a.shape[d]);
        a.host.assign(host, host + a.elems());
        a.have = true;
        return VTTS_OK;
    }
    int pack(void* dev_blob, size_t bytes, void* stream) {
        if (!dev_blob) return failf(VTTS_ERR_INVALID, "null argument");
        if (bytes < blob_bytes) return failf(VTTS_ERR_NOMEM, "blob too small: %zu < %zu bytes", bytes, blob_bytes);
        for (auto& a : arrs)
            if (!a.have) return failf(VTTS_ERR_MISSING, "array %s/%s was never set", a.module.c_str(), a.name.c_str());
        std::vector<char> img(blob_bytes, 0);
        for (auto& a : arrs) memcpy(img.data() + a.off, a.host.data(), a.elems() * sizeof(float));
        for (size_t i = 0; i < bns.size(); ++i) {
            const Arr& sc = arrs[find(bns[i].first, "scale")];
            const Arr& var = arrs[find(bns[i].first + "/~/var_ema", "average")];
            float* iv = reinterpret_cast<float*>(img.data() + bn_off[i]);
            for (int c = 0; c < bns[i].second; ++c) iv[c] = sc.host[c] / std::sqrt(var.host[c] + 1e-5f);  // hk.BatchNorm eps
        }
        HIP_TRYN(hipMemcpyAsync(dev_blob, img.data(), blob_bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
        HIP_TRYN(hipStreamSynchronize(static_cast<hipStream_t>(stream)));  // img dies at return
        blob = static_cast<char*>(dev_blob);
        return VTTS_OK;
    }
    int bind(void* dev_blob, size_t bytes) {
        if (!dev_blob) return failf(VTTS_ERR_INVALID, "null argument");
        if (bytes < blob_bytes) return failf(VTTS_ERR_NOMEM, "blob too small: %zu < %zu bytes", bytes, blob_bytes);
        blob = static_cast<char*>(dev_blob);
        return VTTS_OK;
    }
};

}  // namespace

struct vtts_nat_duration : NatModel {
    vtts_nat_duration_cfg cfg;
};
struct vtts_nat_acoustic : NatModel {
    vtts_nat_acoustic_cfg cfg;
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

// y = act(batchnorm_eval(conv1d_same(x))) [+ res]   hk.Conv1D(Cout, K, padding="SAME"): w[K][Cin][Cout], cross-correlation,
// pads ((K-1)/2, K/2).  Token encoder: K = 3, BatchNorm + ReLU (model.py:28-34); postnet: K = 5, BatchNorm + tanh, the
// last layer plain and added to its input's source (model.py:113-121, :151).  inv = scale * rsqrt(var + eps) comes from
// pack time (nullptr = no BatchNorm).  Rows at or past the sequence's length read as zero (the reference runs each
// sequence alone, so its SAME padding sees zeros there) and are written as zero.
enum { NAT_ACT_NONE = 0, NAT_ACT_RELU = 1, NAT_ACT_TANH = 2 };
template <int K, int TL>
__global__ __launch_bounds__(256) void nat_conv_bn_act_k(const float* __restrict__ x, const int* __restrict__ lengths, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ inv,
                                                         const float* __restrict__ mean, const float* __restrict__ offset,
                                                         const float* __restrict__ res, float* __restrict__ y, int Lmax, int Cin, int Cout,
                                                         int act) {
    extern __shared__ float xs[];  // (TL + K - 1) x Cin
    constexpr int PL = (K - 1) / 2;
    const int b = blockIdx.y, t0 = blockIdx.x * TL;
    const int len = lengths[b];
    for (int i = threadIdx.x; i < (TL + K - 1) * Cin; i += blockDim.x) {
        const int r = i / Cin, c = i % Cin;
        const int t = t0 - PL + r;
        xs[i] = (t >= 0 && t < len) ? x[((size_t)b * Lmax + t) * Cin + c] : 0.0f;
    }
    __syncthreads();
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
        float acc[TL];
        const float bv = bias[co];
#pragma unroll
        for (int i = 0; i < TL; ++i) acc[i] = bv;
        for (int j = 0; j < K; ++j) {
            const float* __restrict__ wj = w + (size_t)j * Cin * Cout + co;
            for (int ci = 0; ci < Cin; ++ci) {
                const float wv = wj[(size_t)ci * Cout];
#pragma unroll
                for (int i = 0; i < TL; ++i) acc[i] = fmaf(xs[(i + j) * Cin + ci], wv, acc[i]);
            }
        }
        const bool bn = inv != nullptr;
        const float iv = bn ? inv[co] : 1.0f, mv = bn ? mean[co] : 0.0f, ov = bn ? offset[co] : 0.0f;
#pragma unroll
        for (int i = 0; i < TL; ++i) {
            const int t = t0 + i;
            if (t < Lmax) {
                float v = bn ? (acc[i] - mv) * iv + ov : acc[i];
                if (act == NAT_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (act == NAT_ACT_TANH) v = tanhf(v);
                const size_t o = ((size_t)b * Lmax + t) * Cout + co;
                if (res) v = res[o] + v;
                y[o] = t < len ? v : 0.0f;
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

// AcousticModel.upsample (model.py:102-111): cond[b][f][:] = sum_j softmax_j(-(mid_j - f)^2 / 10) * enc[b][j][:], with
// mid = cumsum(d) - d/2, d in frames.  One block per (frame, sentence); blockDim = 256; E = 2D encoder channels.
__global__ __launch_bounds__(256) void nat_upsample_k(const float* __restrict__ enc, const int* __restrict__ lengths, const float* __restrict__ dur,
                                                      const int* __restrict__ nframes, float* __restrict__ cond, int Lmax, int Fmax, int E) {
    extern __shared__ float sw[];  // mid[Lmax] then weights[Lmax], + 8 for reductions
    float* mid = sw;
    float* wgt = sw + Lmax;
    float* red = sw + 2 * Lmax;
    const int b = blockIdx.y, f = blockIdx.x, j = threadIdx.x;
    const int len = lengths[b];
    if (f >= nframes[b]) {
        for (int c = j; c < E; c += blockDim.x) cond[((size_t)b * Fmax + f) * E + c] = 0.0f;
        return;
    }
    if (j == 0) {  // jnp.cumsum: sequential fp32 prefix sum
        float end = 0.0f;
        for (int k = 0; k < len; ++k) {
            const float d = dur[(size_t)b * Lmax + k];
            end += d;
            mid[k] = end - d / 2.0f;
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int k = j; k < len; k += blockDim.x) {
        const float z = mid[k] - (float)f;
        const float v = -(z * z) / 10.0f;
        wgt[k] = v;
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
    if ((j & 63) == 0) red[j >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
    for (int k = j; k < len; k += blockDim.x) {
        const float e = expf(wgt[k] - mx);
        wgt[k] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    __syncthreads();  // everyone has read red[] (the max) before it is reused
    if ((j & 63) == 0) red[4 + (j >> 6)] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    for (int c = j; c < E; c += blockDim.x) {
        float acc = 0.0f;
        for (int k = 0; k < len; ++k) acc = fmaf(wgt[k] / sum, enc[((size_t)b * Lmax + k) * E + c], acc);
        cond[((size_t)b * Fmax + f) * E + c] = acc;
    }
}

// AcousticModel.inference's scan body (model.py:134-141), one persistent 1024-thread workgroup per sentence:
//   p = dropout(relu(dropout(relu(prev @ f1)) @ f2))           prenet, no bias, rate 0.5, ALWAYS on (model.py:95-100);
//                                                               keep[b][f][0|1][PN] bytes (1 = keep, value * 2), nullptr = none
//   x = [cond_f ; p];  h1 = LSTM1([x ; h1]);  h2 = LSTM2([[h1 ; x] ; h2])      hk.deep_rnn_with_skip_connections
//   mel_f = [h1 ; h2] @ wp + bp;  prev = mel_f
// Gate columns: thread g owns columns g and g + 1024 of each [in x 4H] matrix (coalesced across threads).
__global__ __launch_bounds__(1024) void nat_decoder_k(const float* __restrict__ cond, const int* __restrict__ nframes, const float* __restrict__ f1,
                                                      const float* __restrict__ f2, const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ wp,
                                                      const float* __restrict__ bp, const unsigned char* __restrict__ keep,
                                                      float* __restrict__ mel, int Fmax, int E, int PN, int H, int MEL) {
    extern __shared__ float sd[];
    float* xin = sd;                 // [E + PN]
    float* h1 = xin + E + PN;        // [H]
    float* h2 = h1 + H;              // [H]
    float* gates = h2 + H;           // [4H]
    float* prev = gates + 4 * H;     // [MEL]
    float* p1 = prev + MEL;          // [PN]
    float* part = p1 + PN;           // [8 * MEL]
    const int b = blockIdx.x, g = threadIdx.x;
    const int nf = nframes[b];
    const int X = E + PN, G4 = 4 * H;
    float c1 = 0.0f, c2 = 0.0f;
    if (g < H) {
        h1[g] = 0.0f;
        h2[g] = 0.0f;
    }
    if (g < MEL) prev[g] = 0.0f;
    const float b1a = b1[g], b1b = b1[g + 1024], b2a = b2[g], b2b = b2[g + 1024];
    __syncthreads();
    for (int f = 0; f < nf; ++f) {
        const unsigned char* kp = keep ? keep + ((size_t)b * Fmax + f) * 2 * PN : nullptr;
        if (g < PN) {
            float a = 0.0f;
            for (int k = 0; k < MEL; ++k) a = fmaf(prev[k], f1[(size_t)k * PN + g], a);
            a = fmaxf(a, 0.0f);
            if (kp) a = kp[g] ? a * 2.0f : 0.0f;
            p1[g] = a;
        } else if (g < PN + E) {
            xin[g - PN] = cond[((size_t)b * Fmax + f) * E + (g - PN)];
        }
        __syncthreads();
        if (g < PN) {
            float a = 0.0f;
            for (int k = 0; k < PN; ++k) a = fmaf(p1[k], f2[(size_t)k * PN + g], a);
            a = fmaxf(a, 0.0f);
            if (kp) a = kp[PN + g] ? a * 2.0f : 0.0f;
            xin[E + g] = a;
        }
        __syncthreads();
        {  // LSTM 1: rows [x (E+PN) ; h1 (H)]
            float a0 = b1a, a1 = b1b;
            const float* __restrict__ wc = w1 + g;
            for (int k = 0; k < X; ++k) {
                const float v = xin[k];
                a0 = fmaf(v, wc[(size_t)k * G4], a0);
                a1 = fmaf(v, wc[(size_t)k * G4 + 1024], a1);
            }
            wc += (size_t)X * G4;
            for (int k = 0; k < H; ++k) {
                const float v = h1[k];
                a0 = fmaf(v, wc[(size_t)k * G4], a0);
                a1 = fmaf(v, wc[(size_t)k * G4 + 1024], a1);
            }
            gates[g] = a0;
            gates[g + 1024] = a1;
        }
        __syncthreads();
        if (g < H) {
            c1 = sigmoidf_(gates[2 * H + g] + 1.0f) * c1 + sigmoidf_(gates[g]) * tanhf(gates[H + g]);
            h1[g] = sigmoidf_(gates[3 * H + g]) * tanhf(c1);
        }
        __syncthreads();
        {  // LSTM 2: rows [h1 (H) ; x (E+PN) ; h2 (H)]
            float a0 = b2a, a1 = b2b;
            const float* __restrict__ wc = w2 + g;
            for (int k = 0; k < H; ++k) {
                const float v = h1[k];
                a0 = fmaf(v, wc[(size_t)k * G4], a0);
                a1 = fmaf(v, wc[(size_t)k * G4 + 1024], a1);
            }
            wc += (size_t)H * G4;
            for (int k = 0; k < X; ++k) {
                const float v = xin[k];
                a0 = fmaf(v, wc[(size_t)k * G4], a0);
                a1 = fmaf(v, wc[(size_t)k * G4 + 1024], a1);
            }
            wc += (size_t)X * G4;
            for (int k = 0; k < H; ++k) {
                const float v = h2[k];
                a0 = fmaf(v, wc[(size_t)k * G4], a0);
                a1 = fmaf(v, wc[(size_t)k * G4 + 1024], a1);
            }
            gates[g] = a0;  // LSTM 1's gate values were consumed before the barrier above
            gates[g + 1024] = a1;
        }
        __syncthreads();
        if (g < H) {
            c2 = sigmoidf_(gates[2 * H + g] + 1.0f) * c2 + sigmoidf_(gates[g]) * tanhf(gates[H + g]);
            h2[g] = sigmoidf_(gates[3 * H + g]) * tanhf(c2);
        }
        __syncthreads();
        if (g < 8 * MEL) {  // projection: 8 partial dot products of 2H/8 terms per mel bin
            const int m = g % MEL, ch = g / MEL, per = 2 * H / 8;
            float a = 0.0f;
            for (int k = ch * per; k < (ch + 1) * per; ++k) a = fmaf(k < H ? h1[k] : h2[k - H], wp[(size_t)k * MEL + m], a);
            part[g] = a;
        }
        __syncthreads();
        if (g < MEL) {
            float a = bp[g];
            for (int ch = 0; ch < 8; ++ch) a += part[ch * MEL + g];
            prev[g] = a;
            mel[((size_t)b * Fmax + f) * MEL + g] = a;
        }
        __syncthreads();
    }
    for (size_t i = g; i < (size_t)(Fmax - nf) * MEL; i += 1024) mel[((size_t)b * Fmax + nf) * MEL + i] = 0.0f;  // rows past the end
}

// ---- shared host-side sequence: TokenEncoder of `m` under module prefix `te` -> enc [B][Lmax][2D] ------------------
int run_token_encoder(const NatModel& m, const std::string& te, int V, int D, const int32_t* tokens, const int32_t* lengths, int B, int Lmax,
                      float* bufA, float* bufB, float* enc, hipStream_t s) {
    hipLaunchKernelGGL(nat_embed_k, dim3(Lmax, B), dim3(256), 0, s, tokens, lengths, m.dev(te + "embed", "embeddings"), bufA, Lmax, D, V);
    constexpr int TL = 8;
    float* cur = bufA;
    float* nxt = bufB;
    for (int i = 0; i < 3; ++i) {
        const std::string sfx = i ? "_" + std::to_string(i) : "";
        const std::string cv = te + "conv1_d" + sfx, bn = te + "batch_norm" + sfx;
        hipLaunchKernelGGL((nat_conv_bn_act_k<3, TL>), dim3((Lmax + TL - 1) / TL, B), dim3(256), (TL + 2) * D * sizeof(float), s, cur, lengths,
                           m.dev(cv, "w"), m.dev(cv, "b"), m.inv(bn), m.dev(bn + "/~/mean_ema", "average"), m.dev(bn, "offset"), nullptr, nxt, Lmax,
                           D, D, (int)NAT_ACT_RELU);
        std::swap(cur, nxt);
    }
    hipLaunchKernelGGL(nat_lstm_k, dim3(B, 2), dim3(4 * D), 6 * D * sizeof(float), s, cur, lengths, m.dev(te + "lstm/linear", "w"),
                       m.dev(te + "lstm/linear", "b"), m.dev(te + "lstm_1/linear", "w"), m.dev(te + "lstm_1/linear", "b"), enc, Lmax, D);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "token encoder launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}

int check_encoder_dims(const char* what, int D, int V) {
    if (D < 64 || D > 256 || D % 64 != 0 || V < 1)
        return failf(VTTS_ERR_INVALID, "%s: encoder width must be 64, 128, 192 or 256 (one thread per gate column, 4*width <= 1024) and vocab_size >= 1 (got %d, %d)",
                     what, D, V);
    return VTTS_OK;
}

}  // namespace

// ================================================ C ABI: duration model ================================================
VTTS_API int vtts_nat_duration_create(const vtts_nat_duration_cfg* cfg, int device, vtts_nat_duration** out) {
    if (!cfg || !out) return failf(VTTS_ERR_INVALID, "null argument");
    const int D = cfg->lstm_dim, V = cfg->vocab_size;
    if (int rc = check_encoder_dims("duration model", D, V)) return rc;
    auto* h = new (std::nothrow) vtts_nat_duration();
    if (!h) return failf(VTTS_ERR_NOMEM, "host allocation failed");
    h->what = "duration model";
    h->cfg = *cfg;
    h->device = device;
    h->add_token_encoder("token_encoder/~/", V, D);
    h->add("linear", "w", {2 * D, D});
    h->add("linear", "b", {D});
    h->add("linear_1", "w", {D, 1});
    h->add("linear_1", "b", {1});
    h->layout();
    *out = h;
    return VTTS_OK;
}
VTTS_API void vtts_nat_duration_destroy(vtts_nat_duration* h) { delete h; }
VTTS_API int vtts_nat_duration_num_params(const vtts_nat_duration* h, int* n) {
    if (!h || !n) return failf(VTTS_ERR_INVALID, "null argument");
    *n = (int)h->arrs.size();
    return VTTS_OK;
}
VTTS_API int vtts_nat_duration_param_info(const vtts_nat_duration* h, int i, const char** module, const char** name, int64_t shape[3], int* ndim) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->param_info(i, module, name, shape, ndim);
}
VTTS_API int vtts_nat_duration_set_param(vtts_nat_duration* h, const char* module, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->set_param(module, name, host, shape, ndim);
}
VTTS_API int vtts_nat_duration_packed_bytes(const vtts_nat_duration* h, size_t* bytes) {
    if (!h || !bytes) return failf(VTTS_ERR_INVALID, "null argument");
    *bytes = h->blob_bytes;
    return VTTS_OK;
}
VTTS_API int vtts_nat_duration_pack(vtts_nat_duration* h, void* dev_blob, size_t blob_bytes, void* stream) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->pack(dev_blob, blob_bytes, stream);
}
VTTS_API int vtts_nat_duration_bind_packed(vtts_nat_duration* h, void* dev_blob, size_t blob_bytes) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->bind(dev_blob, blob_bytes);
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
    rc = run_token_encoder(*h, "token_encoder/~/", V, D, tokens_dev, lengths_dev, B, Lmax, bufA, bufB, enc, s);
    if (rc) return rc;
    hipLaunchKernelGGL(nat_duration_head_k, dim3(Lmax, B), dim3(D), (2 * D + 16) * sizeof(float), s, enc, lengths_dev, h->dev("linear", "w"),
                       h->dev("linear", "b"), h->dev("linear_1", "w"), h->dev("linear_1", "b"), durations_dev, Lmax, D);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "duration model launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}

// ================================================ C ABI: acoustic model ================================================
VTTS_API int vtts_nat_acoustic_create(const vtts_nat_acoustic_cfg* cfg, int device, vtts_nat_acoustic** out) {
    if (!cfg || !out) return failf(VTTS_ERR_INVALID, "null argument");
    const int D = cfg->encoder_dim, V = cfg->vocab_size, H = cfg->decoder_dim, PN = cfg->prenet_dim, MEL = cfg->mel_dim, PD = cfg->postnet_dim;
    if (int rc = check_encoder_dims("acoustic model", D, V)) return rc;
    if (H != 512 || PN < 1 || PN > 256 || MEL < 1 || MEL > 128 || PD < 1 || PD > 1024 || 2 * D + PN > 1024)
        return failf(VTTS_ERR_INVALID, "acoustic model: decoder_dim must be 512 (two gate columns per thread of a 1024-thread workgroup), prenet_dim <= 256, mel_dim <= 128, postnet_dim <= 1024");
    auto* h = new (std::nothrow) vtts_nat_acoustic();
    if (!h) return failf(VTTS_ERR_NOMEM, "host allocation failed");
    h->what = "acoustic model";
    h->cfg = *cfg;
    h->device = device;
    const int X = 2 * D + PN;
    h->add_token_encoder("token_encoder/~/", V, D);
    h->add("lstm/linear", "w", {X + H, 4 * H});          // decoder layer 1: [x ; h1]
    h->add("lstm/linear", "b", {4 * H});
    h->add("lstm_1/linear", "w", {H + X + H, 4 * H});    // decoder layer 2: [[h1 ; x] ; h2]   (skip connection)
    h->add("lstm_1/linear", "b", {4 * H});
    h->add("linear", "w", {2 * H, MEL});                 // projection of concat(h1, h2)
    h->add("linear", "b", {MEL});
    h->add("linear_1", "w", {MEL, PN});                  // prenet_fc1 (no bias)
    h->add("linear_2", "w", {PN, PN});                   // prenet_fc2 (no bias)
    for (int i = 0; i < 5; ++i) {
        const std::string sfx = i ? "_" + std::to_string(i) : "";
        const int cin = i == 0 ? MEL : PD, cout = i == 4 ? MEL : PD;
        h->add("conv1_d" + sfx, "w", {5, cin, cout});
        h->add("conv1_d" + sfx, "b", {cout});
        if (i < 4) h->add_bn("batch_norm" + sfx, PD);
    }
    h->layout();
    *out = h;
    return VTTS_OK;
}
VTTS_API void vtts_nat_acoustic_destroy(vtts_nat_acoustic* h) { delete h; }
VTTS_API int vtts_nat_acoustic_num_params(const vtts_nat_acoustic* h, int* n) {
    if (!h || !n) return failf(VTTS_ERR_INVALID, "null argument");
    *n = (int)h->arrs.size();
    return VTTS_OK;
}
VTTS_API int vtts_nat_acoustic_param_info(const vtts_nat_acoustic* h, int i, const char** module, const char** name, int64_t shape[3], int* ndim) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->param_info(i, module, name, shape, ndim);
}
VTTS_API int vtts_nat_acoustic_set_param(vtts_nat_acoustic* h, const char* module, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->set_param(module, name, host, shape, ndim);
}
VTTS_API int vtts_nat_acoustic_packed_bytes(const vtts_nat_acoustic* h, size_t* bytes) {
    if (!h || !bytes) return failf(VTTS_ERR_INVALID, "null argument");
    *bytes = h->blob_bytes;
    return VTTS_OK;
}
VTTS_API int vtts_nat_acoustic_pack(vtts_nat_acoustic* h, void* dev_blob, size_t blob_bytes, void* stream) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->pack(dev_blob, blob_bytes, stream);
}
VTTS_API int vtts_nat_acoustic_bind_packed(vtts_nat_acoustic* h, void* dev_blob, size_t blob_bytes) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    return h->bind(dev_blob, blob_bytes);
}
VTTS_API int vtts_nat_acoustic_workspace_bytes(const vtts_nat_acoustic* h, int B, int Lmax, int Fmax, size_t* bytes) {
    if (!h || !bytes) return failf(VTTS_ERR_INVALID, "null argument");
    if (B <= 0 || Lmax <= 0 || Fmax <= 0) return failf(VTTS_ERR_INVALID, "B, Lmax and Fmax must be positive (got %d, %d, %d)", B, Lmax, Fmax);
    const size_t D = h->cfg.encoder_dim, PD = h->cfg.postnet_dim, MEL = h->cfg.mel_dim;
    *bytes = 2 * align_up((size_t)B * Lmax * D * 4, 256) + align_up((size_t)B * Lmax * 2 * D * 4, 256)  // encoder ping-pong + output
             + align_up((size_t)B * Fmax * 2 * D * 4, 256)                                                 // cond
             + align_up((size_t)B * Fmax * MEL * 4, 256)                                                   // decoder mel
             + 2 * align_up((size_t)B * Fmax * PD * 4, 256);                                               // postnet ping-pong
    return VTTS_OK;
}
VTTS_API int vtts_nat_acoustic_forward(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev, const float* durations_dev,
                                       const int32_t* nframes_dev, int B, int Lmax, int Fmax, const uint8_t* keep_dev, float* mel_dev,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !tokens_dev || !lengths_dev || !durations_dev || !nframes_dev || !mel_dev) return failf(VTTS_ERR_INVALID, "null argument");
    if (!h->blob) return failf(VTTS_ERR_STATE, "forward() before pack()/bind_packed()");
    size_t need = 0;
    int rc = vtts_nat_acoustic_workspace_bytes(h, B, Lmax, Fmax, &need);
    if (rc) return rc;
    if (!workspace || workspace_bytes < need) return failf(VTTS_ERR_NOMEM, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    if (Lmax > 2048) return failf(VTTS_ERR_INVALID, "at most 2048 tokens per sentence (upsampling weights live in LDS)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int D = h->cfg.encoder_dim, V = h->cfg.vocab_size, H = h->cfg.decoder_dim, PN = h->cfg.prenet_dim, MEL = h->cfg.mel_dim, PD = h->cfg.postnet_dim;
    const int E = 2 * D;
    char* p = static_cast<char*>(workspace);
    auto take = [&](size_t bytes) {
        float* r = reinterpret_cast<float*>(p);
        p += align_up(bytes, 256);
        return r;
    };
    float* bufA = take((size_t)B * Lmax * D * 4);
    float* bufB = take((size_t)B * Lmax * D * 4);
    float* enc = take((size_t)B * Lmax * E * 4);
    float* cond = take((size_t)B * Fmax * E * 4);
    float* mel0 = take((size_t)B * Fmax * MEL * 4);
    float* pA = take((size_t)B * Fmax * PD * 4);
    float* pB = take((size_t)B * Fmax * PD * 4);
    rc = run_token_encoder(*h, "token_encoder/~/", V, D, tokens_dev, lengths_dev, B, Lmax, bufA, bufB, enc, s);  // model.py:131
    if (rc) return rc;
    hipLaunchKernelGGL(nat_upsample_k, dim3(Fmax, B), dim3(256), (2 * Lmax + 8) * sizeof(float), s, enc, lengths_dev, durations_dev, nframes_dev, cond,
                       Lmax, Fmax, E);  // :132
    const size_t dec_lds = ((size_t)(E + PN) + 2 * H + 4 * H + MEL + PN + 8 * MEL) * sizeof(float);
    hipLaunchKernelGGL(nat_decoder_k, dim3(B), dim3(1024), dec_lds, s, cond, nframes_dev, h->dev("linear_1", "w"), h->dev("linear_2", "w"),
                       h->dev("lstm/linear", "w"), h->dev("lstm/linear", "b"), h->dev("lstm_1/linear", "w"), h->dev("lstm_1/linear", "b"),
                       h->dev("linear", "w"), h->dev("linear", "b"), keep_dev, mel0, Fmax, E, PN, H, MEL);  // :134-150
    // postnet (:113-121) + residual (:151): 4 x (Conv1D(PD, 5) + BatchNorm + tanh), Conv1D(MEL, 5), mel + .
    constexpr int TL = 8;
    const float* cur = mel0;
    float* bufs[2] = {pA, pB};
    for (int i = 0; i < 5; ++i) {
        const std::string sfx = i ? "_" + std::to_string(i) : "";
        const std::string cv = "conv1_d" + sfx, bn = "batch_norm" + sfx;
        const int cin = i == 0 ? MEL : PD, cout = i == 4 ? MEL : PD;
        float* dst = i == 4 ? mel_dev : bufs[i & 1];
        hipLaunchKernelGGL((nat_conv_bn_act_k<5, TL>), dim3((Fmax + TL - 1) / TL, B), dim3(256), (TL + 4) * cin * sizeof(float), s, cur, nframes_dev,
                           h->dev(cv, "w"), h->dev(cv, "b"), i < 4 ? h->inv(bn) : nullptr, i < 4 ? h->dev(bn + "/~/mean_ema", "average") : nullptr,
                           i < 4 ? h->dev(bn, "offset") : nullptr, i == 4 ? mel0 : nullptr, dst, Fmax, cin, cout, i < 4 ? (int)NAT_ACT_TANH : (int)NAT_ACT_NONE);
        cur = dst;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "acoustic model launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}
