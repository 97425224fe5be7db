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
// frame), the autoregressive decoder as one launch per layer per frame over ALL sentences of the batch (two LSTM-512 with
// skip connections on the fp32 matrix cores, cell update in registers; mel projection + next frame's prenet), then the
// 5-layer postnet as fp32 MFMA convolutions.
// The prenet's always-on dropout (model.py:95-100) takes explicit keep masks, which vtts_nat_acoustic_keep_masks can draw on
// the device with jax.random's cipher (Threefry-2x32-20) from per-sentence seeds, and vtts_nat_acoustic_keep_masks_haiku as the
// reference itself draws them from the checkpoint's rng (classic jax.random layout + Haiku's key chain, restated in round 2).
#include "../../include/vtts_nat.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/vtts_hifigan.h"
#include "bf16_common.h"
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
    struct Extra {  // a kernel-private re-layout of checkpoint arrays, built at pack() time behind the plain arrays
        std::string key;
        size_t bytes = 0, off = 0;
        std::function<void(const NatModel&, float*)> fill;
    };
    std::vector<Extra> extras;
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
            add_lstm_mfma(te + l, 2 * D, D);  // rows [x ; h] as hk.LSTM concatenates them
        }
    }
    // an hk.LSTM's [K][4H] matrix in MFMA A-fragment order for nat_dec_lstm_k: [slice = 8 units][K/8][lane][4],
    // element i of lane = W[hrow(8*kb + 4*(lane/32) + i)][gate*H + 8*slice + unit], (unit, gate) = ((lane%32)/4, (lane%32)%4);
    // hrow maps a row of the kernel's state order to the row of the Haiku matrix (identity unless the caller permutes)
    void add_lstm_mfma(const std::string& mod, int K, int H, std::function<int(int)> hrow = nullptr) {
        add_extra(mod + "#mfma", (size_t)K * 4 * H * sizeof(float), [mod, K, H, hrow](const NatModel& m, float* out) {
            const std::vector<float>& W = m.arrs[m.find(mod, "w")].host;
            const int NIT = K / 8;
            for (int sl = 0; sl < H / 8; ++sl)
                for (int kb = 0; kb < NIT; ++kb)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int mrow = lane & 31, lh = lane >> 5, col = (mrow & 3) * H + 8 * sl + (mrow >> 2);
                        for (int i = 0; i < 4; ++i) {
                            const int zr = 8 * kb + 4 * lh + i;
                            out[(((size_t)sl * NIT + kb) * 64 + lane) * 4 + i] = W[(size_t)(hrow ? hrow(zr) : zr) * 4 * H + col];
                        }
                    }
        });
    }
    void add_extra(const std::string& key, size_t bytes, std::function<void(const NatModel&, float*)> fill) {
        Extra e;
        e.key = key;
        e.bytes = bytes;
        e.fill = std::move(fill);
        extras.push_back(std::move(e));
    }
    const float* extra(const std::string& key) const {
        for (auto& e : extras)
            if (e.key == key) return reinterpret_cast<const float*>(blob + e.off);
        return nullptr;
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
        for (auto& e : extras) {
            e.off = off;
            off = align_up(off + e.bytes, 256);
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
        for (auto& e : extras) e.fill(*this, reinterpret_cast<float*>(img.data() + e.off));
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
// round-to-nearest-even bf16 of a float (host side of the bf16x3 split: v0 = bf16(v), v1 = bf16(v - v0); v - v0 is exact in fp32)
static inline unsigned short nat_bf16_rne(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float nat_bf16_to_float(unsigned short h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
struct vtts_nat_acoustic : NatModel {
    vtts_nat_acoustic_cfg cfg;
    int x3 = 0;  // option "bf16x3": LSTM steps, gate GEMM and postnet as three bf16 x bf16 terms per product on the bf16 matrix pipe
    int pp_split = 0;  // experiment builds (VTTS_NAT_PP_EXP) only: 1 = the decoder's projection + prenet step cut along its weights (measured slower; see nat_dec_proj_prenet_k)
    // forward_groups(): the postnet of a group of rows runs on `side` as soon as the decoder has produced the group's last frame
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_gates = nullptr;
    std::vector<hipEvent_t> ev_dec, ev_done;  // per group: decoder frames complete (recorded on the caller's stream) / mel rows complete (on `side`)
    int groups_valid = 0;
    ~vtts_nat_acoustic() {
        for (hipEvent_t e : ev_dec) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_done) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_gates) (void)hipEventDestroy(ev_gates);
        if (side) (void)hipStreamDestroy(side);
    }
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

// Postnet convolutions (model.py:113-121) and the hoisted gate GEMMs on the fp32 matrix cores: y = act(batchnorm_eval(conv1d_same(x))) [+ res],
// channels-last fp32 rows, same zero-beyond-the-length semantics as nat_conv_bn_act_k.  GEMM view: M = cout (A = weights, host-packed
// [mblk][32-channel step][tap][lane][16]: element i of lane = W[tap][32*cs + 16*(lane/32) + i][32*mblk + lane%32]), N = frame,
// k = (32-channel step, tap, channel).  A workgroup = 64 frames x (4 waves x MR m-blocks); a wave owns MR x 2 accumulator blocks.
// Round 4: the B operand goes through LDS.  Round 1-3 had every lane read 64 contiguous bytes of ITS frame's row per step and tap straight
// from L1 (no LDS, no barrier): 64 lanes x 16 bytes from 32 different rows per instruction, five times over for the five taps — the kernel
// sat at 64-80 TF/s, bound by the vector-memory pipeline (software-pipelining those loads changed nothing: profiles/r04_c_kernel_structure_findings.md).
// Now the 64 + K - 1 rows x 32 channels of a step are staged ONCE (coalesced float4 loads along the channels, next step's in flight under this
// step's MFMAs, two LDS buffers, one barrier per step), the taps are shifted views of the tile, and a fragment is one ds_read_b32 per lane
// (row stride 33 floats: 32 consecutive frames of one channel hit 32 banks).  The fmaf chains are the old ones, in the old order: same bits.
template <int K, int MR>
__global__ __launch_bounds__(256, 2) void nat_conv_mfma_k(const float* __restrict__ x, const int* __restrict__ lengths, const float4* __restrict__ wpk,
                                                       const float* __restrict__ bias, const float* __restrict__ inv, const float* __restrict__ mean,
                                                       const float* __restrict__ offset, const float* __restrict__ res, float* __restrict__ y, int Lmax,
                                                       int Cin, int Cout, int act, int tile0) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int NR = 2, PL = (K - 1) / 2, ROWS = 64 + K - 1, RS = 33, UNITS = ROWS * 8, UPT = (UNITS + 255) / 256;
    __shared__ float xs[2][ROWS * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, t0 = (blockIdx.x + tile0) * 64;  // tile0: a launch may cover the 64-frame tiles [tile0, tile0 + gridDim.x) only
    const int len = lengths[b];
    const int MB = (Cout + 31) / 32, NCS = (Cin + 31) / 32;
    const int mb0 = (blockIdx.y * 4 + wave) * MR;
    if (t0 >= len) return;               // rows at or past the length: zero by the caller's memset (last layer) or masked by the reader (uniform per workgroup)
    const bool mine = mb0 < MB;          // a wave without an m-block still helps staging and meets the barriers
    const bool two = len - t0 > 32;      // a sentence's last tile with <= 32 frames left: the second 32-frame block is skipped (uniform)
    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = 32 * (mb0 + mr) + 8 * rq + 4 * lh + i;
                const float bv = (mb0 + mr < MB && co < Cout) ? bias[co] : 0.0f;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) acc[mr][nr][4 * rq + i] = bv;
            }
    const float* __restrict__ xb = x + (size_t)b * Lmax * Cin;
    // staging of step cs: unit u = (row, 4-channel group); unconditional loads from clamped addresses, masked afterwards
    float4 sv[UPT];
    auto stage_load = [&](int cs) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = tid + q * 256, uc = u < UNITS ? u : UNITS - 1;
            const int row = uc >> 3, c = cs * 32 + 4 * (uc & 7);
            const int t = t0 + row - PL;
            const int tc = t < 0 ? 0 : (t >= len ? len - 1 : t);
            const int cc = c + 4 <= Cin ? c : Cin - 4;
            float4 v = *reinterpret_cast<const float4*>(xb + (size_t)tc * Cin + cc);
            if (t != tc || c != cc) v = make_float4(0.f, 0.f, 0.f, 0.f);
            sv[q] = v;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = tid + q * 256;
            if (u >= UNITS) continue;
            float* d = &xs[buf][(u >> 3) * RS + 4 * (u & 7)];
            d[0] = sv[q].x; d[1] = sv[q].y; d[2] = sv[q].z; d[3] = sv[q].w;
        }
    };
    // A operands (weights, L2-resident) one (step, tap) ahead in a second register set: with two workgroups per CU nothing else covers their round trip
    auto load_a = [&](int cs, int j, float4 (&av)[MR][4]) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int mb = mb0 + mr < MB ? mb0 + mr : MB - 1;  // a wave's spare block re-reads the last one; never stored
            const float4* __restrict__ ap = wpk + ((((size_t)mb * NCS + cs) * K + j) * 64 + lane) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) av[mr][q] = ap[q];
        }
    };
    auto mfma_tap = [&](int buf, int j, const float4 (&av)[MR][4]) {
        const float* xr = &xs[buf][(l31 + j) * RS + 16 * lh];  // tile row of frame t0 + l31 + j - PL, this half-wave's 16 channels
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bv[NR];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bv[nr] = (nr == 0 || two) ? xr[nr * 32 * RS + 4 * q + e] : 0.0f;
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    const float a1 = e == 0 ? av[mr][q].x : e == 1 ? av[mr][q].y : e == 2 ? av[mr][q].z : av[mr][q].w;
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        if (nr == 0 || two) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[nr], acc[mr][nr], 0, 0, 0);
                }
            }
        }
    };
    float4 avA[MR][4], avB[MR][4];
    stage_load(0);
    if (mine) load_a(0, 0, avA);
    stage_store(0);
    __syncthreads();
    // steps in pairs (cs0, cs0 + 1) so that LDS buffer and register-set parities are compile-time positions (K is odd: the parity of the first tap flips from
    // one step to the next)
#pragma unroll 1
    for (int cs0 = 0; cs0 < NCS; cs0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cs = cs0 + u;
            if (cs >= NCS) break;  // uniform
            if (cs + 1 < NCS) stage_load(cs + 1);  // in flight under this step's MFMAs
            if (mine) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const bool last = j + 1 == K;
                    const int ncs = last ? cs + 1 : cs, nj = last ? 0 : j + 1;
                    const bool more = ncs < NCS;
                    if (((u * K + j) & 1) == 0) {
                        if (more) load_a(ncs, nj, avB);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_tap(u, j, avA);
                    } else {
                        if (more) load_a(ncs, nj, avA);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_tap(u, j, avB);
                    }
                }
            }
            if (cs + 1 < NCS) {
                stage_store(u ^ 1);  // nobody reads that buffer any more: its last readers passed the barrier that ended step cs - 1
                __syncthreads();
            }
        }
    }
    if (!mine) return;
    const bool bn = inv != nullptr;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int co = 32 * (mb0 + mr) + 8 * rq + 4 * lh;
            if (mb0 + mr >= MB || co >= Cout) continue;  // Cout is a multiple of 4: a lane's 4 channels are in or out together
            float iv[4] = {1.f, 1.f, 1.f, 1.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, ov[4] = {0.f, 0.f, 0.f, 0.f};
            if (bn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    iv[i] = inv[co + i];
                    mv[i] = mean[co + i];
                    ov[i] = offset[co + i];
                }
            }
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = t0 + nr * 32 + l31;
                if (t >= Lmax || (nr == 1 && !two)) continue;
                const size_t o = ((size_t)b * Lmax + t) * Cout + co;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[mr][nr][4 * rq + i];
                    if (bn) v[i] = (v[i] - mv[i]) * iv[i] + ov[i];
                    if (act == NAT_ACT_RELU) v[i] = fmaxf(v[i], 0.0f);
                    else if (act == NAT_ACT_TANH) v[i] = tanhf(v[i]);
                }
                if (res) {
                    const float4 r = *reinterpret_cast<const float4*>(res + o);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                if (t >= len) v[0] = v[1] = v[2] = v[3] = 0.0f;
                *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
}

// The same convolution with every product as three bf16 x bf16 terms on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, fp32 accumulation):
// v = v0 + v1 with v0 = bf16(v), v1 = bf16(v - v0) (16 mantissa bits), x * w ~ x1 w0 + x0 w1 + x0 w0 (the dropped x1 w1 is 2^-18 of the
// product) — the split of the vocoder's bf16x3 engine (kernels_x3.hip; profiles/r04_b_split_findings.md).  An OPTION of the acoustic model
// (vtts_nat_acoustic_set_option "bf16x3"), not its default: the fp32 kernel above is what the parity tests against the reference pin at 5e-5;
// this one is for callers whose vocoder is bf16-class anyway (the text -> waveform pipeline).  Same tile, same epilogue; per 32-channel step
// and tap 6 matrix instructions of 32 cycles instead of 16 of 64.  Weights pre-split at pack time ("…#x3": [mblk][step][tap][16-channel
// half][hi | lo][lane][8] bf16, as many bytes as the fp32 fragments); the activations are split while they are staged: two bf16 LDS planes,
// rows of 32 channels padded to 80 bytes (eight lanes' 16-byte reads cover the 32 banks).
template <int K, int MR>
__global__ __launch_bounds__(256, 2) void nat_conv_x3_k(const float* __restrict__ x, const int* __restrict__ lengths, const uint4* __restrict__ wpk,
                                                     const float* __restrict__ bias, const float* __restrict__ inv, const float* __restrict__ mean,
                                                     const float* __restrict__ offset, const float* __restrict__ res, float* __restrict__ y, int Lmax,
                                                     int Cin, int Cout, int act, int tile0) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    using vtts::bf16x8;
    constexpr int NR = 2, PL = (K - 1) / 2, ROWS = 64 + K - 1, RSB = 40, UNITS = ROWS * 8, UPT = (UNITS + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned short xs[2][2][ROWS * RSB];  // [buffer][hi | lo][row][32 channels + 8 pad] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, t0 = (blockIdx.x + tile0) * 64;
    const int len = lengths[b];
    const int MB = (Cout + 31) / 32, NCS = (Cin + 31) / 32;
    const int mb0 = (blockIdx.y * 4 + wave) * MR;
    if (t0 >= len) return;
    const bool mine = mb0 < MB;
    const bool two = len - t0 > 32;
    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = 32 * (mb0 + mr) + 8 * rq + 4 * lh + i;
                const float bv = (mb0 + mr < MB && co < Cout) ? bias[co] : 0.0f;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) acc[mr][nr][4 * rq + i] = bv;
            }
    const float* __restrict__ xb = x + (size_t)b * Lmax * Cin;
    float4 sv[UPT];
    auto stage_load = [&](int cs) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = tid + q * 256, uc = u < UNITS ? u : UNITS - 1;
            const int row = uc >> 3, c = cs * 32 + 4 * (uc & 7);
            const int t = t0 + row - PL;
            const int tc = t < 0 ? 0 : (t >= len ? len - 1 : t);
            const int cc = c + 4 <= Cin ? c : Cin - 4;
            float4 v = *reinterpret_cast<const float4*>(xb + (size_t)tc * Cin + cc);
            if (t != tc || c != cc) v = make_float4(0.f, 0.f, 0.f, 0.f);
            sv[q] = v;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int u = tid + q * 256;
            if (u >= UNITS) continue;
            const unsigned h0 = vtts::pack_bf16x2(sv[q].x, sv[q].y), h1 = vtts::pack_bf16x2(sv[q].z, sv[q].w);
            const unsigned l0 = vtts::pack_bf16x2(sv[q].x - vtts::bf16_lo(h0), sv[q].y - vtts::bf16_hi(h0));
            const unsigned l1 = vtts::pack_bf16x2(sv[q].z - vtts::bf16_lo(h1), sv[q].w - vtts::bf16_hi(h1));
            const int o = (u >> 3) * RSB + 4 * (u & 7);
            *reinterpret_cast<uint2*>(&xs[buf][0][o]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(&xs[buf][1][o]) = make_uint2(l0, l1);
        }
    };
    // A fragments of one (step, tap): [16-channel half][hi | lo] per m-block, a (step, tap) ahead in a second register set
    auto load_a = [&](int cs, int j, bf16x8 (&av)[MR][2][2]) {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int mb = mb0 + mr < MB ? mb0 + mr : MB - 1;
            const uint4* __restrict__ ap = wpk + ((((size_t)mb * NCS + cs) * K + j) * 4) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) av[mr][ks][pl] = __builtin_bit_cast(bf16x8, ap[(ks * 2 + pl) * 64]);
        }
    };
    auto mfma_tap = [&](int buf, int j, const bf16x8 (&av)[MR][2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bh[NR], bl[NR];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int o = (l31 + j + 32 * nr) * RSB + 16 * ks + 8 * lh;  // frame t0 + 32 nr + l31 + j - PL, channels 16 ks + 8 lh .. + 7 of the step
                bh[nr] = *reinterpret_cast<const bf16x8*>(&xs[buf][0][o]);
                bl[nr] = *reinterpret_cast<const bf16x8*>(&xs[buf][1][o]);
            }
            // the small terms first
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    if (nr == 0 || two) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mr][ks][1], bh[nr], acc[mr][nr], 0, 0, 0);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    if (nr == 0 || two) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mr][ks][0], bl[nr], acc[mr][nr], 0, 0, 0);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    if (nr == 0 || two) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mr][ks][0], bh[nr], acc[mr][nr], 0, 0, 0);
        }
    };
    bf16x8 avA[MR][2][2], avB[MR][2][2];
    stage_load(0);
    if (mine) load_a(0, 0, avA);
    stage_store(0);
    __syncthreads();
#pragma unroll 1
    for (int cs0 = 0; cs0 < NCS; cs0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cs = cs0 + u;
            if (cs >= NCS) break;  // uniform
            if (cs + 1 < NCS) stage_load(cs + 1);
            if (mine) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const bool last = j + 1 == K;
                    const int ncs = last ? cs + 1 : cs, nj = last ? 0 : j + 1;
                    const bool more = ncs < NCS;
                    if (((u * K + j) & 1) == 0) {
                        if (more) load_a(ncs, nj, avB);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_tap(u, j, avA);
                    } else {
                        if (more) load_a(ncs, nj, avA);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_tap(u, j, avB);
                    }
                }
            }
            if (cs + 1 < NCS) {
                stage_store(u ^ 1);
                __syncthreads();
            }
        }
    }
    if (!mine) return;
    const bool bn = inv != nullptr;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int co = 32 * (mb0 + mr) + 8 * rq + 4 * lh;
            if (mb0 + mr >= MB || co >= Cout) continue;
            float iv[4] = {1.f, 1.f, 1.f, 1.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, ov[4] = {0.f, 0.f, 0.f, 0.f};
            if (bn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    iv[i] = inv[co + i];
                    mv[i] = mean[co + i];
                    ov[i] = offset[co + i];
                }
            }
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const int t = t0 + nr * 32 + l31;
                if (t >= Lmax || (nr == 1 && !two)) continue;
                const size_t o = ((size_t)b * Lmax + t) * Cout + co;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[mr][nr][4 * rq + i];
                    if (bn) v[i] = (v[i] - mv[i]) * iv[i] + ov[i];
                    if (act == NAT_ACT_RELU) v[i] = fmaxf(v[i], 0.0f);
                    else if (act == NAT_ACT_TANH) v[i] = tanhf(v[i]);
                }
                if (res) {
                    const float4 r = *reinterpret_cast<const float4*>(res + o);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                if (t >= len) v[0] = v[1] = v[2] = v[3] = 0.0f;
                *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
            }
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
// mid = cumsum(d) - d/2, d in frames; E = 2D encoder channels.
// The conditioning's share of the LSTM gates WITHOUT materialising the conditioning (round 4).  cond[b][f] = sum_k a[b][f][k] enc[b][k]
// only ever meets the first E rows of the two LSTMs' input matrices (x = [cond_f ; p], model.py:134-141), so
//   G_l[b][f] = b_l + cond[b][f] @ W_l[0:E] = b_l + sum_k a[b][f][k] (enc[b][k] @ W_l[0:E]):
// the GEMM runs over TOKENS (EG_l = enc @ W_l[0:E], nat_conv_mfma_k with one tap; a sixth of the frames' rows: 302 -> 37 GFLOP for 256 sentences)
// and this kernel mixes its rows into the frames' rows with the upsampling weights a = softmax_k(-(mid_k - f)^2 / 10), mid = cumsum(d) - d/2.
// A workgroup = NAT_MIX_FT frames x 1024 gate columns of one sentence and one layer: the weights of its frames in LDS ([token][frame]: four
// 16-byte broadcast reads per token), a thread's 4 columns x NAT_MIX_FT frames in registers, EG rows streamed from L2 (coalesced float4).
// Sums in token order, weights from the sentence's own durations: a row does not depend on its batch.
constexpr int NAT_MIX_FT = 16;
__global__ __launch_bounds__(256) void nat_gates_mix_k(const float* __restrict__ eg1, const float* __restrict__ eg2, const float* __restrict__ bias1,
                                                       const float* __restrict__ bias2, const int* __restrict__ lengths, const float* __restrict__ dur,
                                                       const int* __restrict__ nframes, float* __restrict__ G1, float* __restrict__ G2, int Lmax, int Fmax,
                                                       int G4, int tile0) {
    constexpr int FT = NAT_MIX_FT;
    extern __shared__ float sw[];
    float* mid = sw;                       // [Lmax rounded up to 4]
    float* wn = sw + (Lmax + 3) / 4 * 4;   // [Lmax][FT]
    const int b = blockIdx.z, f0 = (blockIdx.x + tile0) * FT, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int len = lengths[b], nf = nframes[b];
    if (f0 >= nf) return;  // rows at or past the sentence's frames are never used (the step kernel masks them)
    const int chunks = G4 / 1024, layer = blockIdx.y / chunks, col = (blockIdx.y % chunks) * 1024 + 4 * t;
    if (t == 0) {  // jnp.cumsum: sequential fp32 prefix sum
        float end = 0.0f;
        for (int k = 0; k < len; ++k) {
            const float d = dur[(size_t)b * Lmax + k];
            end += d;
            mid[k] = end - d / 2.0f;
        }
    }
    __syncthreads();
    // a wave takes FT / 4 of the tile's frames, one after the other: max and sum over the tokens in a fixed butterfly order
    for (int q = 0; q < FT / 4; ++q) {
        const int fi = wave * (FT / 4) + q;
        const float ff = (float)(f0 + fi);
        float mx = -INFINITY;
#pragma clang loop vectorize(disable)  // the loop vectoriser would pair these into v_pk_*_f32 (build.py: no packed-f32 VALU code)
        for (int k = lane; k < len; k += 64) {
            const float z = mid[k] - ff;
            mx = fmaxf(mx, -(z * z) / 10.0f);
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.0f;
#pragma clang loop vectorize(disable)  // the loop vectoriser would pair these into v_pk_*_f32 (build.py: no packed-f32 VALU code)
        for (int k = lane; k < len; k += 64) {
            const float z = mid[k] - ff;
            const float e = expf(-(z * z) / 10.0f - mx);
            wn[k * FT + fi] = e;
            sum += e;
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
#pragma clang loop vectorize(disable)
        for (int k = lane; k < len; k += 64) wn[k * FT + fi] = wn[k * FT + fi] / sum;
    }
    __syncthreads();
    const float* __restrict__ eg = (layer ? eg2 : eg1) + (size_t)b * Lmax * G4 + col;
    const float4 bv = *reinterpret_cast<const float4*>((layer ? bias2 : bias1) + col);
    float4 acc[FT];
#pragma unroll
    for (int i = 0; i < FT; ++i) acc[i] = bv;
#pragma unroll 4
    for (int k = 0; k < len; ++k) {
        const float4 e = *reinterpret_cast<const float4*>(eg + (size_t)k * G4);
        const float4* __restrict__ w4 = reinterpret_cast<const float4*>(wn + k * FT);
#pragma unroll
        for (int q = 0; q < FT / 4; ++q) {
            const float4 w = w4[q];
            const float ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4& a = acc[4 * q + i];
                a.x = fmaf(ws[i], e.x, a.x);
                a.y = fmaf(ws[i], e.y, a.y);
                a.z = fmaf(ws[i], e.z, a.z);
                a.w = fmaf(ws[i], e.w, a.w);
            }
        }
    }
    float* __restrict__ G = (layer ? G2 : G1) + ((size_t)b * Fmax + f0) * G4 + col;
#pragma unroll
    for (int i = 0; i < FT; ++i)
        if (f0 + i < nf) *reinterpret_cast<float4*>(G + (size_t)i * G4) = acc[i];
}

// AcousticModel.inference's scan body (model.py:134-141) for ALL sentences of the batch at once, one launch per layer per
// frame (the recurrence is sequential in frames; sentences are independent):
//   p = dropout(relu(dropout(relu(prev @ f1)) @ f2))           prenet, no bias, rate 0.5, ALWAYS on (model.py:95-100);
//                                                               keep[b][f][0|1][PN] bytes (1 = keep, value * 2), nullptr = none
//   x = [cond_f ; p];  h1 = LSTM1([x ; h1]);  h2 = LSTM2([[x ; h1] ; h2])      hk.deep_rnn_with_skip_connections
//       (dm-haiku recurrent.py, _DeepRNN.__call__: current_inputs = tree_map(concat, inputs, current_inputs), i.e. the
//        NETWORK INPUT first, then the previous layer's output; hk.LSTM then appends its own hidden state)
//   mel_f = [h1 ; h2] @ wp + bp;  prev = mel_f
// Decoder state in HBM, k-major in groups of four rows with the sentences contiguous (Bp = B rounded up to 64 columns),
// ping-pong by frame parity:
//   Z[parity][row / 4][Bp][row % 4]  (one 16-byte load per lane = 4 consecutive rows of its sentence: the LSTM step is bound
//   by the number of vector-memory instructions a CU can issue, and dword loads of the state were 8 of its 9 per iteration),
//   rows [ p (PN) | h1 (H) | h2 (H) ]:  LSTM1 reads p of the current parity then h1 of the previous one; LSTM2 reads
//   [p ; h1] of the current parity then h2 of the previous one.  Haiku's matrices are [cond ; p ; h1] and
//   [cond ; p ; h1 ; h2] (x = [cond ; p] first): the state order is Haiku's order minus the cond rows.
//   **The conditioning's share of the gates is hoisted out of the frame loop (round 4):** cond_f is known for every frame
//   before the loop starts, so G_l[b][f][:] = b_l + cond[b][f] @ W_l[0:E] is ONE fp32-MFMA GEMM per layer ahead of the loop
//   (nat_conv_mfma_k with one tap, output columns in the step kernel's accumulator order) and a step starts its sums from
//   G_l[b][f] instead of from the bias: per-frame K 1280 -> 768 (layer 1) and 1792 -> 1280 (layer 2), a third less of the
//   L2 -> CU weight stream that bounds the step.  Only the first 64 frames' G is computed in front of the loop; the rest runs on
//   a side stream beside the (latency-bound) first 64 steps.  Cell states c1, c2 as [H][Bp].
//
// nat_dec_lstm_k: gates[32 sentences x (8 units x 4 gates)] per wave on the fp32 matrix cores (v_mfma_f32_32x32x2_f32:
// M = the slice's 32 gate columns ordered 4*unit + gate, N = 32 sentences, K = 2 per instruction).  With that row order a
// lane's 16 accumulators are the i, g, f, o pre-activations of 4 (unit, sentence) pairs: the LSTM cell update
// (hk.LSTM: gates i, g, f, o; forget bias +1) happens in registers, no exchange.  Weights host-packed per slice so that one
// 16-byte load per lane feeds 4 MFMAs ([slice][K/8][lane][4]: element i = W[8*kb + 4*(lane/32) + i][col(lane%32)]);
// activations straight from Z (128-byte rows, L2-resident); both PD iterations (8 k each) ahead in registers.
// Every output element depends on its own sentence's column only: rows are bit-identical alone or batched.
__device__ __forceinline__ size_t nat_zidx(int row, int b, int Bp) { return ((size_t)(row >> 2) * Bp + b) * 4 + (row & 3); }
#ifndef VTTS_NAT_PD
#define VTTS_NAT_PD 4
#endif
constexpr int NAT_DEC_PD = VTTS_NAT_PD;   // iterations (8 k each) a wave keeps in flight (7 measured no faster: the step is L2-bandwidth-bound)

// NT = 32-sentence tiles per wave (one weight fragment feeds NT MFMAs: L2 traffic for the weights / NT), KW = waves per
// workgroup, each with a contiguous share of K; their partial sums meet in LDS in a fixed tree order.
// One LSTM's operands; blockIdx.z picks one of two sets (the token encoder steps its forward and backward LSTMs in one launch).
struct NatLstmOps {
    const float* inA;    // KA state rows, then
    const float* inB;    // KB state rows (the LSTM's own previous hidden state)
    const float4* wpk;   // [slice][K/8][lane][4]
    const float* bias;   // [4H], gates i, g, f, o
    float* cst;          // cell state [H][Bp]
    float* hout;         // new hidden state, state layout
    const float* gin;    // optional: this step's gate pre-activations computed ahead of the loop (bias + the contribution of inputs known in
                         // advance), [sentence][gpitch floats] with the step's 4H values in the order ((slice * 2 + lane / 32) * 4 + unit pair) * 4 + gate:
                         // a lane's 16 accumulators are 64 contiguous bytes.  nullptr = start from the bias.
    size_t gpitch;
};
template <int NT, int KW, int SL = 1>
__global__ __launch_bounds__(64 * KW) void nat_dec_lstm_k(NatLstmOps ops0, NatLstmOps ops1, int KA, int KB, const int* __restrict__ nframes, int f, int B,
                                                          int Bp, int H) {
    // SL = slices (8 units each) per workgroup: a wave's state fragments feed SL weight fragments, so the state's share of the L2 -> CU stream
    // (2/3 of it at SL = 1, NT = 2: every slice's workgroup reads the whole state of its sentences) falls by SL.  Each output element's sum is
    // the same chain in the same order whatever SL is.  MEASURED (round 4, 256 sentences): SL = 2 makes the step 19.3 -> 28.9 us — the step is not
    // bound by that stream but by what ONE workgroup has to do (its fp32 MFMAs: 2 waves per SIMD x K/64 iterations x 8 x 64 cycles = 6-10 us, and
    // a cold L2 at every launch: 17.5 MB of misses per step, PMC passes in profiles/r04_e_nat_decoder_findings.md); the launches use SL = 1.
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    static_assert(KW == 1 || KW == 2 || KW == 4 || KW == 8, "tree reduction");
    __shared__ float red[KW][SL][NT][16][64];  // every wave's share of the gate sums
    const NatLstmOps& ops = blockIdx.z ? ops1 : ops0;
    const float* __restrict__ inA = ops.inA;
    const float* __restrict__ inB = ops.inB;
    const float4* __restrict__ wpk = ops.wpk;
    const float* __restrict__ bias = ops.bias;
    float* __restrict__ cst = ops.cst;
    float* __restrict__ hout = ops.hout;
    const int lane = threadIdx.x & 63, kw = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int slice0 = blockIdx.x * SL, b0 = blockIdx.y * 32 * NT;
    const int NIT = (KA + KB) / 8, NWMAX = (NIT + KW - 1) / KW, it_lo = kw * NWMAX;
    const int NW = it_lo >= NIT ? 0 : (NIT - it_lo < NWMAX ? NIT - it_lo : NWMAX);  // this wave's iterations [it_lo, it_lo + NW)
    // this wave's cell-update blocks ((slice, sentence tile, unit pair) kw, kw + KW, ...): their cell states are requested now, a kernel's length
    // before they are needed (round 4: loaded after the reduction they were ~1 us of every step)
    constexpr int NBLK = (SL * NT * 4 + KW - 1) / KW;
    float cold[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q) {
        const int blk = kw + q * KW, sl = blk / (NT * 4), nt = (blk / 4) % NT, rq = blk % 4;
        cold[q] = blk < SL * NT * 4 ? cst[(size_t)(8 * (slice0 + sl) + 2 * rq + lh) * Bp + b0 + 32 * nt + l31] : 0.0f;
    }
    f32x16 acc[SL][NT][2];
    if (ops.gin != nullptr && kw == 0) {
        // the sum starts from the hoisted part (bias + the inputs known ahead of the loop, themselves an MFMA chain in k order)
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int b = b0 + 32 * nt + l31;
                const float4* __restrict__ gp =
                    reinterpret_cast<const float4*>(ops.gin + (size_t)(b < B ? b : B - 1) * ops.gpitch + (size_t)(2 * (slice0 + sl) + lh) * 16);
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const float4 g4 = gp[rq];
                    acc[sl][nt][0][4 * rq + 0] = g4.x;
                    acc[sl][nt][0][4 * rq + 1] = g4.y;
                    acc[sl][nt][0][4 * rq + 2] = g4.z;
                    acc[sl][nt][0][4 * rq + 3] = g4.w;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[sl][nt][1][4 * rq + i] = 0.0f;
                }
            }
    } else {
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float bv = (kw == 0 && ops.gin == nullptr) ? bias[i * H + 8 * (slice0 + sl) + 2 * rq + lh] : 0.0f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[sl][nt][0][4 * rq + i] = bv;
                        acc[sl][nt][1][4 * rq + i] = 0.0f;
                    }
                }
    }
    float4 wv[NAT_DEC_PD][SL];
    float4 xv[NAT_DEC_PD][NT];
    const float4* __restrict__ wsl = wpk + (size_t)slice0 * NIT * 64 + lane;
    auto load_it = [&](int it, int slot) {
        if (it >= NIT) it = NIT - 1;  // tail: an in-bounds re-read, never used
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) wv[slot][sl] = wsl[((size_t)sl * NIT + it) * 64];
        const int k0 = it * 8;
        // rows k0 + 4*lh .. + 3 of this lane's sentences: MFMA j of the iteration takes k = k0 + 4*(lane/32) + j on both operands
        const float* __restrict__ xr = (k0 < KA ? inA + (size_t)k0 * Bp : inB + (size_t)(k0 - KA) * Bp) + ((size_t)lh * Bp + b0 + l31) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xv[slot][nt] = *reinterpret_cast<const float4*>(xr + (size_t)(32 * nt) * 4);
    };
#pragma unroll
    for (int j = 0; j < NAT_DEC_PD; ++j) load_it(it_lo + j, j);
    // which sentences still decode is looked up only now, behind the first operands' loads (the early exit of a finished tile waits for an L2
    // round trip; in front of everything it was that much of EVERY step: 0.18 ms of the bf16x3 acoustic model)
    bool live[NT];
    bool any = false;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int b = b0 + 32 * nt + l31;
        live[nt] = b < B && f < nframes[b < B ? b : B - 1];
        any = any || live[nt];
    }
    if (__ballot(any) == 0ull) return;  // every sentence of these tiles has all its frames (same for all waves)
#pragma nounroll
    for (int it0 = 0; it0 < NW; it0 += NAT_DEC_PD) {
#pragma unroll
        for (int j = 0; j < NAT_DEC_PD; ++j) {
            if (it0 + j >= NW) break;  // wave-uniform
#pragma unroll
            for (int sl = 0; sl < SL; ++sl)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[sl][nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][sl].x, xv[j][nt].x, acc[sl][nt][0], 0, 0, 0);
#pragma unroll
            for (int sl = 0; sl < SL; ++sl)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[sl][nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][sl].y, xv[j][nt].y, acc[sl][nt][1], 0, 0, 0);
#pragma unroll
            for (int sl = 0; sl < SL; ++sl)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[sl][nt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][sl].z, xv[j][nt].z, acc[sl][nt][0], 0, 0, 0);
#pragma unroll
            for (int sl = 0; sl < SL; ++sl)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[sl][nt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][sl].w, xv[j][nt].w, acc[sl][nt][1], 0, 0, 0);
            const int nx = it0 + j + NAT_DEC_PD;
            load_it(nx < NW ? it_lo + nx : NIT, j);
        }
    }
#pragma unroll
    for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sl][nt][0][r] += acc[sl][nt][1][r];
    // every wave leaves its share of the sums in LDS; after ONE barrier a wave adds the KW shares of its own cell-update blocks in wave order
    // (round 4: a three-level tree with a barrier per level cost ~0.5 us of every step)
    if constexpr (KW > 1) {
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[kw][sl][nt][r][lane] = acc[sl][nt][0][r];
        __syncthreads();
    }
    // The cell update (3 sigmoids + 2 tanh per (unit, sentence): ~1000 VALU instructions per lane for a wave's 8 pairs) is shared out: wave w takes
    // the (slice, sentence tile, unit pair) blocks w, w + KW, ... (round 2 left it to wave 0 alone while the other seven idled: ~2.5 us of a step).
    auto cell_update = [&](int sl, int nt, int rq, float c, float gi, float gg, float gf, float go) {
        if (!live[nt]) return;
        const int b = b0 + 32 * nt + l31;
        const int u = 8 * (slice0 + sl) + 2 * rq + lh;
        c = sigmoidf_(gf + 1.0f) * c + sigmoidf_(gi) * tanhf(gg);
        cst[(size_t)u * Bp + b] = c;
        hout[nat_zidx(u, b, Bp)] = sigmoidf_(go) * tanhf(c);
    };
    if constexpr (KW == 1) {
#pragma unroll
        for (int blk = 0; blk < SL * NT * 4; ++blk) {
            const int sl = blk / (NT * 4), nt = (blk / 4) % NT, rq = blk % 4;
            cell_update(sl, nt, rq, cold[blk], acc[sl][nt][0][4 * rq + 0], acc[sl][nt][0][4 * rq + 1], acc[sl][nt][0][4 * rq + 2], acc[sl][nt][0][4 * rq + 3]);
        }
    } else {
#pragma unroll
        for (int blk = 0; blk < SL * NT * 4; ++blk) {
            if (blk % KW != kw) continue;  // wave-uniform
            const int sl = blk / (NT * 4), nt = (blk / 4) % NT, rq = blk % 4;
            float gs[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = red[0][sl][nt][4 * rq + i][lane];
#pragma unroll
                for (int w = 1; w < KW; ++w) v += red[w][sl][nt][4 * rq + i][lane];
                gs[i] = v;
            }
            cell_update(sl, nt, rq, cold[blk / KW], gs[0], gs[1], gs[2], gs[3]);
        }
    }
}

// (Round 4 built the decoder as ONE resident kernel per run of frames — bit-identical, 26.7 ms against 22.4 with the per-frame launches; the L2s of the 8 XCDs
//  are not coherent, so every barrier costs the workgroups their cached state — and did not ship it: profiles/r04_e_nat_decoder_findings.md.  The kernel lives in
//  tools/kbench/experiments/nat_persist_*.inc and is compiled only into experiment builds: --define VTTS_NAT_PERSIST=1.)
#ifdef VTTS_NAT_PERSIST
#include "../../tools/kbench/experiments/nat_persist_kernel.inc"
#endif

// ---- the decoder step with the option "bf16x3": the gate sums as three bf16 x bf16 terms per product on the bf16 matrix pipe ----------------
// The state lives in HBM already split: Zx[parity][hi | lo][row / 8][Bp][8] bf16 (rows [p | h1 | h2]; h = hi + lo to 16 mantissa bits), so that a
// lane's B fragment of v_mfma_f32_32x32x16_bf16 (sentence lane % 32, rows 16 step + 8 (lane / 32) .. + 7) is one 16-byte load per plane; the
// weights are split at pack time ("…#x3": [slice][K / 16][hi | lo][lane][8] bf16, the fp32 fragments' bytes).  Per 16 rows and 32-sentence tile
// three matrix instructions of 32 cycles instead of eight of 64; everything around them (the hoisted gates G in fp32, the tree over the K
// shares, the cell update in fp32 registers, c in fp32) is the fp32 step's.  Every output element still depends on its own sentence only.
__device__ __forceinline__ size_t nat_zxidx(int row, int b, int Bp) { return ((size_t)(row >> 3) * Bp + b) * 8 + (row & 7); }
struct NatLstmX3Ops {
    const unsigned short* zc;  // this frame's parity (rows [0, KA) are read from it), plane 0; plane 1 at + plane
    const unsigned short* zp;  // the previous frame's (rows [KA, K))
    size_t plane;              // bf16 elements between the hi and the lo plane
    const uint4* wpk;          // [slice][K / 16][2][64] x 16 bytes
    const float* gin;          // this step's hoisted gate pre-activations (NatLstmOps::gin)
    size_t gpitch;
    float* cst;                // [H][Bp] fp32
    unsigned short* hout;      // zc, plane 0: the new hidden state goes to rows out_row0 + unit
    int out_row0;
};
template <int NT, int KW>
__global__ __launch_bounds__(64 * KW) void nat_dec_lstm_x3_k(NatLstmX3Ops ops, int KA, int K, const int* __restrict__ nframes, int f, int B, int Bp, int H) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    using vtts::bf16x8;
    constexpr int PD = 3;
    __shared__ float red[KW][NT][16][64];
    const int lane = threadIdx.x & 63, kw = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const int slice = blockIdx.x, b0 = blockIdx.y * 32 * NT;
    const int NST = K / 16, NWMAX = (NST + KW - 1) / KW, st_lo = kw * NWMAX;
    const int NW = st_lo >= NST ? 0 : (NST - st_lo < NWMAX ? NST - st_lo : NWMAX);
    f32x16 acc[NT];
    if (kw == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int b = b0 + 32 * nt + l31;
            const float4* __restrict__ gp = reinterpret_cast<const float4*>(ops.gin + (size_t)(b < B ? b : B - 1) * ops.gpitch + (size_t)(2 * slice + lh) * 16);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float4 g4 = gp[rq];
                acc[nt][4 * rq + 0] = g4.x;
                acc[nt][4 * rq + 1] = g4.y;
                acc[nt][4 * rq + 2] = g4.z;
                acc[nt][4 * rq + 3] = g4.w;
            }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    }
    // this wave's cell-update block(s): the cell state is requested now, a kernel's length before it is needed
    constexpr int NBLK = (NT * 4 + KW - 1) / KW;
    float cold[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q) {
        const int blk = kw + q * KW, nt = blk / 4, rq = blk % 4;
        cold[q] = blk < NT * 4 ? ops.cst[(size_t)(8 * slice + 2 * rq + lh) * Bp + b0 + 32 * nt + l31] : 0.0f;
    }
    uint4 wv[PD][2], xv[PD][NT][2];
    const uint4* __restrict__ wsl = ops.wpk + (size_t)slice * NST * 2 * 64 + lane;
    auto load_st = [&](int st, int slot) {
        if (st >= NST) st = NST - 1;  // tail: an in-bounds re-read, never used
        wv[slot][0] = wsl[(size_t)(2 * st) * 64];
        wv[slot][1] = wsl[(size_t)(2 * st + 1) * 64];
        const unsigned short* __restrict__ zs = (16 * st < KA ? ops.zc : ops.zp) + ((size_t)(2 * st + lh) * Bp + b0 + l31) * 8;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            xv[slot][nt][0] = *reinterpret_cast<const uint4*>(zs + (size_t)(32 * nt) * 8);
            xv[slot][nt][1] = *reinterpret_cast<const uint4*>(zs + ops.plane + (size_t)(32 * nt) * 8);
        }
    };
#pragma unroll
    for (int j = 0; j < PD; ++j) load_st(st_lo + j, j);
    // (the frame counts behind the first operands' loads, as in nat_dec_lstm_k)
    bool live[NT];
    bool any = false;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int b = b0 + 32 * nt + l31;
        live[nt] = b < B && f < nframes[b < B ? b : B - 1];
        any = any || live[nt];
    }
    if (__ballot(any) == 0ull) return;
#pragma nounroll
    for (int i0 = 0; i0 < NW; i0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            if (i0 + j >= NW) break;  // wave-uniform
            const bf16x8 whi = __builtin_bit_cast(bf16x8, wv[j][0]), wlo = __builtin_bit_cast(bf16x8, wv[j][1]);
            // the small terms first
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, __builtin_bit_cast(bf16x8, xv[j][nt][0]), acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, __builtin_bit_cast(bf16x8, xv[j][nt][1]), acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, __builtin_bit_cast(bf16x8, xv[j][nt][0]), acc[nt], 0, 0, 0);
            const int nx = i0 + j + PD;
            load_st(nx < NW ? st_lo + nx : NST, j);
        }
    }
    // every wave leaves its share of the sums in LDS; after ONE barrier a wave adds the KW shares of its own cell-update block in wave order
    // (round 4: a three-level tree with a barrier per level, and the cell state loaded only after it, cost 1.1 us of every step)
    {
        float* redf = &red[0][0][0][0];  // [KW][NT][16][64] (the kernel's LDS is sized for it under this switch)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) redf[((kw * NT + nt) * 16 + r) * 64 + lane] = acc[nt][r];
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < NT * 4; ++blk) {
        if (blk % KW != kw) continue;  // wave-uniform
        const int nt = blk / 4, rq = blk % 4;
        if (!live[nt]) continue;
        float gs[4];
        {
            const float* redf = &red[0][0][0][0];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = redf[((0 * NT + nt) * 16 + 4 * rq + i) * 64 + lane];
#pragma unroll
                for (int w = 1; w < KW; ++w) v += redf[((w * NT + nt) * 16 + 4 * rq + i) * 64 + lane];
                gs[i] = v;
            }
        }
        const float gi = gs[0], gg = gs[1], gf = gs[2], go = gs[3];
        const int b = b0 + 32 * nt + l31, u = 8 * slice + 2 * rq + lh;
        float c = cold[blk / KW];
        c = sigmoidf_(gf + 1.0f) * c + sigmoidf_(gi) * tanhf(gg);
        ops.cst[(size_t)u * Bp + b] = c;
        const float hv = sigmoidf_(go) * tanhf(c);
        const unsigned hp = vtts::pack_bf16x2(hv, 0.0f);
        const unsigned lp = vtts::pack_bf16x2(hv - vtts::bf16_lo(hp), 0.0f);
        const size_t o = nat_zxidx(ops.out_row0 + u, b, Bp);
        ops.hout[o] = (unsigned short)(hp & 0xffffu);
        ops.hout[o + ops.plane] = (unsigned short)(lp & 0xffffu);
    }
}

// TokenEncoder's two LSTMs (model.py:39-46) on the same batched step kernel: hk.LSTM over [x_t ; h] is the decoder step with
// KA = D input rows and KB = D hidden rows, and all sentences advance together (one launch per token position steps BOTH
// directions: blockIdx.z).  The step kernel wants its operands k-major with the sentences contiguous, so
//   nat_enc_scatter_k : x [B][Lmax][D] -> XT[dir][s][D/4][Bp][4]; step s of the backward LSTM reads token len-1-s, which IS
//                       hk.dynamic_unroll over the length-reversed sequence (jnp.flip with the ResetCore reset falling on the
//                       padding steps, where the state still is the initial state);
//   step s            : reads XT[dir][s] and HS[dir][s] (slab 0 = zeros = the initial state), writes HS[dir][s + 1] for the
//                       sentences with s < len (the others' columns are never read again);
//   nat_enc_gather_k  : enc[b][t][dir * D + j] = HS[dir][1 + (dir ? len-1-t : t)][j][b], zero at and beyond the length.
__global__ __launch_bounds__(256) void nat_enc_scatter_k(const float* __restrict__ x, const int* __restrict__ lengths, float* __restrict__ xt, int B, int Bp,
                                                         int Lmax, int D) {
    const int s = blockIdx.x, dir = blockIdx.z, D4 = D / 4;
    const size_t slab = (size_t)D * Bp;
    float4* __restrict__ dst = reinterpret_cast<float4*>(xt + ((size_t)dir * Lmax + s) * slab);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < D4 * Bp; i += gridDim.y * 256) {
        const int k4 = i / Bp, b = i - k4 * Bp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B) {
            const int len = lengths[b];
            if (s < len) v = *reinterpret_cast<const float4*>(x + ((size_t)b * Lmax + (dir ? len - 1 - s : s)) * D + 4 * k4);
        }
        dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void nat_enc_gather_k(const float* __restrict__ hs, const int* __restrict__ lengths, float* __restrict__ enc, int Bp,
                                                        int Lmax, int D) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int len = lengths[b];
    const size_t slab = (size_t)D * Bp;
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int dir = c >= D, j = c - dir * D;
        float v = 0.0f;
        if (t < len) v = hs[((size_t)dir * (Lmax + 1) + 1 + (dir ? len - 1 - t : t)) * slab + nat_zidx(j, b, Bp)];
        enc[((size_t)b * Lmax + t) * 2 * D + c] = v;
    }
}

// Keep masks for the prenet's dropout drawn on the device: Threefry-2x32 with 20 rounds (Salmon et al., SC'11 — the
// block cipher jax.random is built on), key = the sentence's 64-bit seed, counter = (2 * frame + layer, 64-column block);
// the 64 output bits are the keep flags of 64 consecutive prenet columns (P(keep) = 1/2 = 1 - rate, model.py:97,99).
// This is a stream of our own (one seed per sentence, for batches of unrelated sentences); the reference's schedule from the
// checkpoint's rng is nat_keep_masks_haiku_k below.
__device__ __forceinline__ void threefry2x32_20(unsigned k0, unsigned k1, unsigned& x0, unsigned& x1) {
    const unsigned ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
    const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
    x0 += ks[0];
    x1 += ks[1];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rot = R[(g & 1) * 4 + r];
            x0 += x1;
            x1 = (x1 << rot) | (x1 >> (32 - rot));
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (unsigned)(g + 1);
    }
}
__global__ void nat_keep_masks_k(const unsigned long long* __restrict__ seeds, unsigned char* __restrict__ keep, int B, int Fmax, int PN) {
    const int nblk = (PN + 63) / 64;
    const size_t n = (size_t)B * Fmax * 2 * nblk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int blk = (int)(i % nblk);
        const size_t fl = i / nblk;  // (b * Fmax + f) * 2 + layer
        const int b = (int)(fl / (2 * (size_t)Fmax));
        const unsigned ctr0 = (unsigned)(fl % (2 * (size_t)Fmax));
        const unsigned long long sd = seeds[b];
        unsigned x0 = ctr0, x1 = (unsigned)blk;
        threefry2x32_20((unsigned)sd, (unsigned)(sd >> 32), x0, x1);
        unsigned char* dst = keep + fl * PN + (size_t)blk * 64;
        for (int j = 0; j < 64 && blk * 64 + j < PN; ++j) dst[j] = (unsigned char)(((j < 32 ? x0 >> j : x1 >> (j - 32)) & 1u));
    }
}

// The REFERENCE's mask stream (jax.random's classic threefry layout under dm-haiku's PRNGSequence; restated in
// oracle/nat_oracle.py::haiku_prenet_keep_masks, which carries the derivation): K_0 = the checkpoint's rng,
// (K_n, S_n) = jax.random.split(K_{n-1}) = the cipher on counters (0, 2) and (1, 3), frame f takes S_{2f+1} / S_{2f+2}, and a mask of
// PN columns is uniform(S, (1, PN)) < 0.5: column c < half takes word x0 of counter (c, c + half), column half + c word x1
// (half = ceil(PN / 2)); keep <=> the word's top bit is clear.  Every sentence of a batch gets the same masks (the reference runs
// every sentence from the same checkpoint key).  One block per frame: its threads walk the key chain to frame f together (wave-
// uniform, <= 4 * (f + 1) ciphers), then thread (layer, c) draws its word and writes it to all B sentences.
// mode 1 = the layout of jax_threefry_partitionable=True (JAX >= 0.5's default; oracle/nat_oracle.py::jax_partitionable_*, restated from
// recollection of jax/_src/prng.py and NOT pinned by any known answer): split(key, 2)[i] = cipher(key, (0, i)) as the pair
// (y0, y1); a 32-bit draw of element c = y0 ^ y1 of cipher(S, (0, c)).
// Only the block's first wave walks the key chain (2 f + 2 splits, wave-uniform) and hands the frame's two subkeys to the others
// through LDS (round 2: every thread of the block walked it).
__global__ void nat_keep_masks_haiku_k(unsigned k0, unsigned k1, int mode, unsigned char* __restrict__ keep, int B, int Fmax, int PN) {
    __shared__ unsigned sub[4];  // s0[0], s1[0], s0[1], s1[1]
    const int f = blockIdx.x;
    if (threadIdx.x < 64) {
        unsigned ka = k0, kb = k1;
        for (int n = 0; n < 2 * f + 2; ++n) {
            unsigned a0, b0, a1, b1;
            if (mode == 0) {
                a0 = 0u, b0 = 2u, a1 = 1u, b1 = 3u;  // counts iota(4) in halves: pairs (0, 2), (1, 3)
                threefry2x32_20(ka, kb, a0, b0);
                threefry2x32_20(ka, kb, a1, b1);
                if (n >= 2 * f && threadIdx.x == 0) {  // [1] = (y1[0], y1[1]) is handed out
                    sub[2 * (n - 2 * f)] = b0;
                    sub[2 * (n - 2 * f) + 1] = b1;
                }
                ka = a0;  // split(key, 2)[0] = (y0[0], y0[1]) stays the sequence's key
                kb = a1;
            } else {
                a0 = 0u, b0 = 0u, a1 = 0u, b1 = 1u;  // subkey i = cipher(key, (0, i))
                threefry2x32_20(ka, kb, a0, b0);
                threefry2x32_20(ka, kb, a1, b1);
                if (n >= 2 * f && threadIdx.x == 0) {
                    sub[2 * (n - 2 * f)] = a1;
                    sub[2 * (n - 2 * f) + 1] = b1;
                }
                ka = a0;
                kb = b0;
            }
        }
    }
    __syncthreads();
    const int half = (PN + 1) / 2;
    for (int u = threadIdx.x; u < 2 * PN; u += blockDim.x) {
        const int layer = u / PN, c = u % PN;
        unsigned word;
        if (mode == 0) {
            const int i = c < half ? c : c - half;
            unsigned x0 = (unsigned)i, x1 = i + half < PN ? (unsigned)(i + half) : 0u;  // counters iota(PN) in two halves; an odd count is padded with one 0
            threefry2x32_20(sub[2 * layer], sub[2 * layer + 1], x0, x1);
            word = c < half ? x0 : x1;
        } else {
            unsigned x0 = 0u, x1 = (unsigned)c;
            threefry2x32_20(sub[2 * layer], sub[2 * layer + 1], x0, x1);
            word = x0 ^ x1;
        }
        const unsigned char kp = (word >> 31) ? 0 : 1;  // uniform = (word >> 9 | 1.0f) - 1 < 0.5  <=>  top bit clear
        for (int b = 0; b < B; ++b) keep[(((size_t)b * Fmax + f) * 2 + layer) * PN + c] = kp;
    }
}

// (Round 4 measured two restructurings of this kernel and kept neither: every weight fragment loaded ahead of its use, 16.3-18.7 -> 19.4 us, and
// 4-5 adjacent columns per thread to cut the LDS operand reads by four, 16.0 -> 20.0 us — the step is bound by one CU streaming the three
// matrices' 670 KB through its vector-memory path, ~5 us, plus three dependent phases: profiles/r04_e_nat_decoder_findings.md.)
// mel_f = [h1 ; h2] @ wp + bp, then the prenet of frame f + 1 into the other parity's state.  One
// 1024-thread workgroup per 4 sentences (weights read once per k for the four).  Every product is split over k into
// 1024 / width partial sums that are added in chunk order: a frame step is latency-bound, short dependent chains matter.
template <bool X3>  // X3: the state is the bf16x3 step's (two bf16 planes, nat_zxidx; `plane` elements apart); h = hi + lo exactly, p is split on its way out
__global__ __launch_bounds__(1024) void nat_dec_proj_prenet_k(const float* __restrict__ zcur, float* __restrict__ znext,
                                                              const int* __restrict__ nframes, const float4* __restrict__ f1, const float4* __restrict__ f2,
                                                              const float4* __restrict__ wp, const float* __restrict__ bp,
                                                              const unsigned char* __restrict__ keep, float* __restrict__ mel, int f, int B, int Bp,
                                                              int Fmax, int PN, int H, int MEL, size_t plane) {
    extern __shared__ float4 sq[];
    float4* hs = sq;              // [2H]   h1 ; h2 of the 4 sentences
    float4* part = hs + 2 * H;    // [1024] partial sums of the product in flight
    float4* prev = part + 1024;   // [MEL]
    float4* p1 = prev + MEL;      // [PN]
    const int g = threadIdx.x, b0 = blockIdx.x * 4;
    int nf[4];
    bool any = false;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        nf[s] = b0 + s < B ? nframes[b0 + s] : 0;
        any = any || f < nf[s];
    }
    if (!any) return;
    // the projection's bias and the keep bytes of frame f + 1, requested before anything else (each was a cold load behind a barrier: 0.7 us of a step)
    const float bb_h = g < MEL ? bp[g] : 0.0f;
    unsigned char kp[2][4];
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            kp[which][s] = (keep && g < PN && b0 + s < B && f + 1 < Fmax) ? keep[(((size_t)(b0 + s) * Fmax + f + 1) * 2 + which) * PN + g] : (unsigned char)1;
#pragma clang loop vectorize(disable)  // (it would pair the hi + lo additions of two rows into v_pk_add_f32: build.py)
    for (int k = g; k < 2 * H; k += 1024) {
        if constexpr (X3) {
            const unsigned short* __restrict__ zr = reinterpret_cast<const unsigned short*>(zcur) + nat_zxidx(PN + k, b0, Bp);  // sentences 8 elements apart
            float v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = __builtin_bit_cast(float, (unsigned)zr[8 * s] << 16) + __builtin_bit_cast(float, (unsigned)zr[plane + 8 * s] << 16);
            hs[k] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            const float* __restrict__ zr = zcur + nat_zidx(PN + k, b0, Bp);  // state rows [p | h1 | h2]; the 4 sentences of a row are 16 bytes apart
            hs[k] = make_float4(zr[0], zr[4], zr[8], zr[12]);
        }
    }
    __syncthreads();
    // out[col] (4 sentences) = sum over chunk `ch` of rows [ch*per, (ch+1)*per) of src[row] * w[row][col]
    // weights in [row / 4][col][4] order (pack-time copies "…#k4"): one 16-byte load per lane = 4 consecutive rows of its
    // column, a wave's loads 1 KiB contiguous — dword loads made the step wait on the number of vector-memory instructions
    // a CU can issue (2 700 per workgroup and frame)
    auto partial = [&](const float4* __restrict__ src, const float4* __restrict__ w4, int rows, int width, int col, int ch, int per, int phase_bit) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k1 = (ch + 1) * per < rows ? (ch + 1) * per : rows;  // per and rows are multiples of 4
#ifdef VTTS_NAT_PKFMA  // bisect builds only (tools/experiments/r05/pkfma_build.sh): the packed-f32 variants of this loop, wrong by construction beside a bf16 MFMA stream
#include "../../tools/kbench/experiments/nat_pkfma_partial.inc"
#endif
#pragma unroll 4
        for (int k = ch * per; k < k1; k += 4) {
            const float4 wv = w4[(size_t)(k >> 2) * width + col];
            const float4 x0 = src[k], x1 = src[k + 1], x2 = src[k + 2], x3 = src[k + 3];
            a.x = fmaf(x0.x, wv.x, a.x); a.y = fmaf(x0.y, wv.x, a.y); a.z = fmaf(x0.z, wv.x, a.z); a.w = fmaf(x0.w, wv.x, a.w);
            a.x = fmaf(x1.x, wv.y, a.x); a.y = fmaf(x1.y, wv.y, a.y); a.z = fmaf(x1.z, wv.y, a.z); a.w = fmaf(x1.w, wv.y, a.w);
            a.x = fmaf(x2.x, wv.z, a.x); a.y = fmaf(x2.y, wv.z, a.y); a.z = fmaf(x2.z, wv.z, a.z); a.w = fmaf(x2.w, wv.z, a.w);
            a.x = fmaf(x3.x, wv.w, a.x); a.y = fmaf(x3.y, wv.w, a.y); a.z = fmaf(x3.z, wv.w, a.z); a.w = fmaf(x3.w, wv.w, a.w);
        }
        return a;
    };
    auto gather = [&](float4 a, int width, int col, int nch) {
        for (int ch = 0; ch < nch; ++ch) {
            const float4 q = part[ch * width + col];
            a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
        }
        return a;
    };
    const int nchP = 1024 / MEL, perP = ((2 * H + nchP - 1) / nchP + 3) / 4 * 4;
    if (g < nchP * MEL) part[g] = partial(hs, wp, 2 * H, MEL, g % MEL, g / MEL, perP, 1);
    __syncthreads();
    if (g < MEL) {
        const float bb = bb_h;
        const float4 a = gather(make_float4(bb, bb, bb, bb), MEL, g, nchP);
        prev[g] = a;
        const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (f < nf[s]) mel[((size_t)(b0 + s) * Fmax + f) * MEL + g] = v[s];
    }
    __syncthreads();
    if (f + 1 >= Fmax) return;
    auto masked = [&](float4 a, int which, int col) {  // relu, then hk.dropout(rate 0.5) with the given keep bytes of frame f + 1
        float v[4] = {fmaxf(a.x, 0.0f), fmaxf(a.y, 0.0f), fmaxf(a.z, 0.0f), fmaxf(a.w, 0.0f)};
        if (keep) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (b0 + s < B) v[s] = kp[which][s] ? v[s] * 2.0f : 0.0f;  // (col == g for both callers)
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    const int nchN = 1024 / PN;
    if (g < nchN * PN) part[g] = partial(prev, f1, MEL, PN, g % PN, g / PN, ((MEL + nchN - 1) / nchN + 3) / 4 * 4, 2);
    __syncthreads();
    if (g < PN) p1[g] = masked(gather(make_float4(0.f, 0.f, 0.f, 0.f), PN, g, nchN), 0, g);
    __syncthreads();
    if (g < nchN * PN) part[g] = partial(p1, f2, PN, PN, g % PN, g / PN, ((PN + nchN - 1) / nchN + 3) / 4 * 4, 4);
    __syncthreads();
    if (g < PN) {
        const float4 r = masked(gather(make_float4(0.f, 0.f, 0.f, 0.f), PN, g, nchN), 1, g);
        if constexpr (X3) {
            unsigned short* __restrict__ zw = reinterpret_cast<unsigned short*>(znext) + nat_zxidx(g, b0, Bp);
            const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const unsigned hp = vtts::pack_bf16x2(v[s], 0.0f);
                const unsigned lp = vtts::pack_bf16x2(v[s] - vtts::bf16_lo(hp), 0.0f);
                zw[8 * s] = (unsigned short)(hp & 0xffffu);
                zw[plane + 8 * s] = (unsigned short)(lp & 0xffffu);
            }
        } else {
            float* __restrict__ zw = znext + nat_zidx(g, b0, Bp);
            zw[0] = r.x; zw[4] = r.y; zw[8] = r.z; zw[12] = r.w;
        }
    }
}

// (Round 6 cut the projection + prenet step along its WEIGHTS — a workgroup = a 32-sentence tile x a slice of a matrix, two launches per frame over 64-workgroup
//  grids, every load requested before the first wait — and measured it SLOWER: 6.7 + 24.0 us per frame against this kernel's 15.2, the acoustic model 17.0 ms against
//  12.5 (profiles/r06_c_nat_proj_prenet_findings.md).  The kernels live in tools/kbench/experiments/nat_pp_split_*.inc: --define VTTS_NAT_PP_EXP=1, option "pp_split".)
#ifdef VTTS_NAT_PP_EXP
#include "../../tools/kbench/experiments/nat_pp_split_kernels.inc"
#endif
// ---- shared host-side sequence: TokenEncoder of `m` under module prefix `te` -> enc [B][Lmax][2D] ------------------
// scratch of the two encoder LSTMs: XT[2][Lmax], HS[2][Lmax + 1] slabs of [D][Bp] and the two cell states
size_t nat_enc_lstm_floats(int D, int B, int Lmax) {
    const size_t Bp = (size_t)(B + 63) / 64 * 64;
    return ((size_t)2 * Lmax + 2 * ((size_t)Lmax + 1) + 2) * D * Bp;
}
int run_token_encoder(const NatModel& m, const std::string& te, int V, int D, const int32_t* tokens, const int32_t* lengths, int B, int Lmax,
                      float* bufA, float* bufB, float* lstm_ws, float* enc, hipStream_t s) {
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
    {  // forward + backward hk.LSTM (model.py:39-46), every sentence and both directions per launch
        const int Bp = (B + 63) / 64 * 64;
        const size_t slab = (size_t)D * Bp;
        float* xt = lstm_ws;                                   // [2][Lmax] slabs
        float* hs = xt + 2 * (size_t)Lmax * slab;              // [2][Lmax + 1] slabs
        float* cs = hs + 2 * ((size_t)Lmax + 1) * slab;        // [2] slabs
        HIP_TRYN(hipMemsetAsync(hs, 0, slab * 4, s));
        HIP_TRYN(hipMemsetAsync(hs + ((size_t)Lmax + 1) * slab, 0, slab * 4, s));
        HIP_TRYN(hipMemsetAsync(cs, 0, 2 * slab * 4, s));
        const int gy = (int)std::min<size_t>((slab / 4 + 255) / 256, 64);
        hipLaunchKernelGGL(nat_enc_scatter_k, dim3(Lmax, gy, 2), dim3(256), 0, s, cur, lengths, xt, B, Bp, Lmax, D);
        const bool wide = B > 32;  // two 32-sentence tiles per wave once there are that many sentences
        const dim3 lgrid(D / 8, wide ? Bp / 64 : 1, 2);
        NatLstmOps o[2];
        for (int dir = 0; dir < 2; ++dir) {
            const std::string mod = te + (dir ? "lstm_1/linear" : "lstm/linear");
            o[dir].wpk = reinterpret_cast<const float4*>(m.extra(mod + "#mfma"));
            o[dir].bias = m.dev(mod, "b");
            o[dir].cst = cs + dir * slab;
            o[dir].gin = nullptr;
            o[dir].gpitch = 0;
        }
        for (int st = 0; st < Lmax; ++st) {
            for (int dir = 0; dir < 2; ++dir) {
                o[dir].inA = xt + ((size_t)dir * Lmax + st) * slab;
                o[dir].inB = hs + ((size_t)dir * (Lmax + 1) + st) * slab;
                o[dir].hout = hs + ((size_t)dir * (Lmax + 1) + st + 1) * slab;
            }
            if (wide) hipLaunchKernelGGL((nat_dec_lstm_k<2, 8>), lgrid, dim3(512), 0, s, o[0], o[1], D, D, lengths, st, B, Bp, D);
            else hipLaunchKernelGGL((nat_dec_lstm_k<1, 8>), lgrid, dim3(512), 0, s, o[0], o[1], D, D, lengths, st, B, Bp, D);
        }
        hipLaunchKernelGGL(nat_enc_gather_k, dim3(Lmax, B), dim3(256), 0, s, hs, lengths, enc, Bp, Lmax, D);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "token encoder launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}

int check_encoder_dims(const char* what, int D, int V) {
    if (D < 64 || D > 256 || D % 64 != 0 || V < 1)
        return failf(VTTS_ERR_INVALID, "%s: encoder width must be 64, 128, 192 or 256 (one thread per channel in the convolutions) and vocab_size >= 1 (got %d, %d)",
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
    // two ping-pong [B][Lmax][D] buffers + the encoder output [B][Lmax][2D] + the LSTMs' scratch
    *bytes = 2 * align_up((size_t)B * Lmax * D * 4, 256) + align_up((size_t)B * Lmax * 2 * D * 4, 256) + align_up(nat_enc_lstm_floats((int)D, B, Lmax) * 4, 256);
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
    float* lstm_ws = reinterpret_cast<float*>(static_cast<char*>(workspace) + 2 * per + align_up((size_t)B * Lmax * 2 * D * 4, 256));
    rc = run_token_encoder(*h, "token_encoder/~/", V, D, tokens_dev, lengths_dev, B, Lmax, bufA, bufB, lstm_ws, enc, s);
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
    if (H < 32 || H > 1024 || H % 32 != 0 || PN < 32 || PN % 32 != 0 || MEL < 4 || MEL > 128 || MEL % 4 != 0 || PD < 4 || PD > 1024 || PD % 4 != 0 || 2 * D + PN > 1024 || (2 * D) % 32 != 0)
        return failf(VTTS_ERR_INVALID, "acoustic model: decoder_dim and prenet_dim must be multiples of 32 (matrix-core k-steps), decoder_dim <= 1024, "
                                       "2 * encoder_dim + prenet_dim <= 1024, mel_dim <= 128 and postnet_dim <= 1024 multiples of 4");
    auto* h = new (std::nothrow) vtts_nat_acoustic();
    if (!h) return failf(VTTS_ERR_NOMEM, "host allocation failed");
    h->what = "acoustic model";
    h->cfg = *cfg;
    h->device = device;
    const int X = 2 * D + PN;
    h->add_token_encoder("token_encoder/~/", V, D);
    h->add("lstm/linear", "w", {X + H, 4 * H});          // decoder layer 1: [x ; h1]
    h->add("lstm/linear", "b", {4 * H});
    h->add("lstm_1/linear", "w", {X + H + H, 4 * H});    // decoder layer 2: [[x ; h1] ; h2]   (skip connection: input first)
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
    // postnet convolution weights in MFMA A-fragment order (nat_conv_mfma_k): [mblk][32-channel step][tap][lane][16]
    for (int i = 0; i < 5; ++i) {
        const std::string mod = "conv1_d" + (i ? "_" + std::to_string(i) : std::string());
        const int cin = i == 0 ? MEL : PD, cout = i == 4 ? MEL : PD, MB = (cout + 31) / 32, NCS = (cin + 31) / 32;
        h->add_extra(mod + "#mfma", (size_t)MB * NCS * 5 * 64 * 16 * sizeof(float), [mod, cin, cout, MB, NCS](const NatModel& m, float* out) {
            const std::vector<float>& W = m.arrs[m.find(mod, "w")].host;
            for (int mb = 0; mb < MB; ++mb)
                for (int cs = 0; cs < NCS; ++cs)
                    for (int j = 0; j < 5; ++j)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 16; ++e) {
                                const int c = 32 * cs + 16 * (lane >> 5) + e, co = 32 * mb + (lane & 31);
                                out[((((size_t)mb * NCS + cs) * 5 + j) * 64 + lane) * 16 + e] = (c < cin && co < cout) ? W[((size_t)j * cin + c) * cout + co] : 0.0f;
                            }
        });
    }
    // ... and split into two bf16 terms for nat_conv_x3_k: [mblk][step][tap][16-channel half][hi | lo][lane][8] bf16
    for (int i = 0; i < 5; ++i) {
        const std::string mod = "conv1_d" + (i ? "_" + std::to_string(i) : std::string());
        const int cin = i == 0 ? MEL : PD, cout = i == 4 ? MEL : PD, MB = (cout + 31) / 32, NCS = (cin + 31) / 32;
        h->add_extra(mod + "#x3", (size_t)MB * NCS * 5 * 4 * 64 * 8 * sizeof(unsigned short), [mod, cin, cout, MB, NCS](const NatModel& m, float* outf) {
            unsigned short* out = reinterpret_cast<unsigned short*>(outf);
            const std::vector<float>& W = m.arrs[m.find(mod, "w")].host;
            for (int mb = 0; mb < MB; ++mb)
                for (int cs = 0; cs < NCS; ++cs)
                    for (int j = 0; j < 5; ++j)
                        for (int ks = 0; ks < 2; ++ks)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 8; ++e) {
                                    const int c = 32 * cs + 16 * ks + 8 * (lane >> 5) + e, co = 32 * mb + (lane & 31);
                                    const float w = (c < cin && co < cout) ? W[((size_t)j * cin + c) * cout + co] : 0.0f;
                                    const unsigned short hi = nat_bf16_rne(w), lo = nat_bf16_rne(w - nat_bf16_to_float(hi));
                                    const size_t base = ((((size_t)mb * NCS + cs) * 5 + j) * 4 + ks * 2) * 64;
                                    out[(base + lane) * 8 + e] = hi;
                                    out[(base + 64 + lane) * 8 + e] = lo;
                                }
        });
    }
    // projection and prenet matrices in [row / 4][col][4] order (nat_dec_proj_prenet_k: one 16-byte load = 4 rows of a column)
    for (const char* l : {"linear", "linear_1", "linear_2"}) {
        const std::string mod = l;
        const int rows = mod == "linear" ? 2 * H : (mod == "linear_1" ? MEL : PN), cols = mod == "linear" ? MEL : PN;
        h->add_extra(mod + "#k4", (size_t)rows * cols * sizeof(float), [mod, rows, cols](const NatModel& m, float* out) {
            const std::vector<float>& W = m.arrs[m.find(mod, "w")].host;
            for (int k = 0; k < rows; ++k)
                for (int c = 0; c < cols; ++c) out[((size_t)(k >> 2) * cols + c) * 4 + (k & 3)] = W[(size_t)k * cols + c];
        });
    }
    // decoder LSTM weights in the step kernel's order (add_lstm_mfma).  Haiku's matrices are [cond ; p ; h1] (layer 1) and
    // [cond ; p ; h1 ; h2] (layer 2: hk.deep_rnn_with_skip_connections puts the network input first); the per-frame step multiplies
    // the state rows [p ; h1] / [p ; h1 ; h2] = Haiku rows E + r, and the cond rows [0, E) go into the GEMM ahead of the loop:
    // "#cond" = those rows as a one-tap convolution for nat_conv_mfma_k with the output columns in the step kernel's accumulator order
    // c' = ((slice * 2 + lane / 32) * 4 + unit pair) * 4 + gate  <->  Haiku column gate * H + 8 * slice + 2 * (unit pair) + lane / 32,
    // "#condb" = the bias in that order.
    const int E = 2 * D;
    h->add_lstm_mfma("lstm/linear", PN + H, H, [E](int zr) { return E + zr; });
    h->add_lstm_mfma("lstm_1/linear", PN + H + H, H, [E](int zr) { return E + zr; });
    // ... and split into two bf16 terms for nat_dec_lstm_x3_k: [slice][K / 16][hi | lo][lane][8] bf16, element i of lane = the bf16 term of
    // W[E + 16*step + 8*(lane/32) + i][gate*H + 8*slice + unit], (unit, gate) = ((lane%32)/4, (lane%32)%4)
    for (int l = 0; l < 2; ++l) {
        const std::string mod = l ? "lstm_1/linear" : "lstm/linear";
        const int K = l ? PN + 2 * H : PN + H;
        h->add_extra(mod + "#x3", (size_t)K * 4 * H * sizeof(float), [mod, K, H, E](const NatModel& m, float* outf) {
            unsigned short* out = reinterpret_cast<unsigned short*>(outf);
            const std::vector<float>& W = m.arrs[m.find(mod, "w")].host;
            const int NST = K / 16;
            for (int sl = 0; sl < H / 8; ++sl)
                for (int st = 0; st < NST; ++st)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int mrow = lane & 31, lh = lane >> 5, col = (mrow & 3) * H + 8 * sl + (mrow >> 2);
                        for (int i = 0; i < 8; ++i) {
                            const float w = W[(size_t)(E + 16 * st + 8 * lh + i) * 4 * H + col];
                            const unsigned short hi = nat_bf16_rne(w), lo = nat_bf16_rne(w - nat_bf16_to_float(hi));
                            const size_t base = ((size_t)sl * NST + st) * 2 * 64;
                            out[(base + lane) * 8 + i] = hi;
                            out[(base + 64 + lane) * 8 + i] = lo;
                        }
                    }
        });
    }
    for (const char* l : {"lstm/linear", "lstm_1/linear"}) {
        const std::string mod = l;
        const int G4 = 4 * H, MB = G4 / 32, NCS = E / 32;
        auto hcol = [H](int cp) {  // accumulator-order column -> Haiku column
            const int gate = cp & 3, rq = (cp >> 2) & 3, lh = (cp >> 4) & 1, slice = cp >> 5;
            return gate * H + 8 * slice + 2 * rq + lh;
        };
        h->add_extra(mod + "#cond", (size_t)MB * NCS * 64 * 16 * sizeof(float), [mod, E, G4, MB, NCS, hcol](const NatModel& m, float* out) {
            const std::vector<float>& W = m.arrs[m.find(mod, "w")].host;
            for (int mb = 0; mb < MB; ++mb)
                for (int cs = 0; cs < NCS; ++cs)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 16; ++e) {
                            const int c = 32 * cs + 16 * (lane >> 5) + e, cp = 32 * mb + (lane & 31);
                            out[(((size_t)mb * NCS + cs) * 64 + lane) * 16 + e] = W[(size_t)c * G4 + hcol(cp)];
                        }
        });
        h->add_extra(mod + "#condb", (size_t)G4 * sizeof(float), [mod, G4, hcol](const NatModel& m, float* out) {
            const std::vector<float>& bv = m.arrs[m.find(mod, "b")].host;
            for (int cp = 0; cp < G4; ++cp) out[cp] = bv[hcol(cp)];
        });
        h->add_extra(mod + "#cond#x3", (size_t)MB * NCS * 4 * 64 * 8 * sizeof(unsigned short), [mod, E, G4, MB, NCS, hcol](const NatModel& m, float* outf) {
            unsigned short* out = reinterpret_cast<unsigned short*>(outf);  // the same rows for nat_conv_x3_k: [mblk][step][16-channel half][hi | lo][lane][8] bf16
            const std::vector<float>& W = m.arrs[m.find(mod, "w")].host;
            for (int mb = 0; mb < MB; ++mb)
                for (int cs = 0; cs < NCS; ++cs)
                    for (int ks = 0; ks < 2; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int c = 32 * cs + 16 * ks + 8 * (lane >> 5) + e, cp = 32 * mb + (lane & 31);
                                const float w = W[(size_t)c * G4 + hcol(cp)];
                                const unsigned short hi = nat_bf16_rne(w), lo = nat_bf16_rne(w - nat_bf16_to_float(hi));
                                const size_t base = (((size_t)mb * NCS + cs) * 4 + ks * 2) * 64;
                                out[(base + lane) * 8 + e] = hi;
                                out[(base + 64 + lane) * 8 + e] = lo;
                            }
        });
        h->add_extra(mod + "#zerob", (size_t)G4 * sizeof(float), [G4](const NatModel&, float* out) {  // the token-rows GEMM adds no bias (the mix does)
            for (int cp = 0; cp < G4; ++cp) out[cp] = 0.0f;
        });
    }
    h->layout();
    *out = h;
    return VTTS_OK;
}
VTTS_API void vtts_nat_acoustic_destroy(vtts_nat_acoustic* h) { delete h; }
VTTS_API int vtts_nat_acoustic_set_option(vtts_nat_acoustic* h, const char* key, int value) {
    if (!h || !key) return failf(VTTS_ERR_INVALID, "null argument");
    if (!strcmp(key, "bf16x3")) {
        if (value != 0 && value != 1) return failf(VTTS_ERR_INVALID, "bf16x3 must be 0 or 1 (got %d)", value);
        h->x3 = value;
        return VTTS_OK;
    }
#ifdef VTTS_NAT_PP_EXP
    if (!strcmp(key, "pp_split")) {
        if (value != 0 && value != 1) return failf(VTTS_ERR_INVALID, "pp_split must be 0 or 1 (got %d)", value);
        h->pp_split = value;
        return VTTS_OK;
    }
#endif
    return failf(VTTS_ERR_INVALID, "unknown option '%s' (known: bf16x3)", key);
}
VTTS_API int vtts_nat_acoustic_get_option(const vtts_nat_acoustic* h, const char* key, int* value) {
    if (!h || !key || !value) return failf(VTTS_ERR_INVALID, "null argument");
    if (!strcmp(key, "bf16x3")) {
        *value = h->x3;
        return VTTS_OK;
    }
#ifdef VTTS_NAT_PP_EXP
    if (!strcmp(key, "pp_split")) {
        *value = h->pp_split;
        return VTTS_OK;
    }
#endif
    return failf(VTTS_ERR_INVALID, "unknown option '%s' (known: bf16x3)", key);
}
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
static size_t nat_dec_state_floats(const vtts_nat_acoustic_cfg& c, int B) {
    const size_t Bp = (size_t)(B + 63) / 64 * 64, H = c.decoder_dim, ZW = 2 * H + c.prenet_dim;  // rows [p | h1 | h2]
    return (2 * ZW + 2 * H) * Bp;
}
VTTS_API int vtts_nat_acoustic_workspace_bytes(const vtts_nat_acoustic* h, int B, int Lmax, int Fmax, size_t* bytes) {
    if (!h || !bytes) return failf(VTTS_ERR_INVALID, "null argument");
    if (B <= 0 || Lmax <= 0 || Fmax <= 0) return failf(VTTS_ERR_INVALID, "B, Lmax and Fmax must be positive (got %d, %d, %d)", B, Lmax, Fmax);
    const size_t D = h->cfg.encoder_dim, PD = h->cfg.postnet_dim, MEL = h->cfg.mel_dim;
    *bytes = 2 * align_up((size_t)B * Lmax * D * 4, 256) + align_up((size_t)B * Lmax * 2 * D * 4, 256)  // encoder ping-pong + output
             + 2 * align_up((size_t)B * Lmax * 4 * h->cfg.decoder_dim * 4, 256)                           // EG1, EG2: enc @ W_l[0:E] per token
             + align_up((size_t)B * Fmax * MEL * 4, 256)                                                   // decoder mel
             + 2 * align_up((size_t)B * Fmax * PD * 4, 256)                                                // postnet ping-pong
             + align_up(nat_dec_state_floats(h->cfg, B) * 4, 256)                                          // decoder state Z[2], c1, c2
             + 2 * align_up((size_t)B * Fmax * 4 * h->cfg.decoder_dim * 4, 256)                           // hoisted gate pre-activations G1, G2
             + align_up(nat_enc_lstm_floats((int)D, B, Lmax) * 4, 256)                                     // encoder LSTMs' scratch
#ifdef VTTS_NAT_PP_EXP
             + align_up(nat_pp_part_floats((int)MEL, B) * 4, 256)                                          // the projection's partial sums of a frame
#endif
             + 4096;                                                                                       // resident decoder kernel: barrier counters, failure flag
    return VTTS_OK;
}
VTTS_API int vtts_nat_acoustic_keep_masks(const vtts_nat_acoustic* h, const uint64_t* seeds_dev, int B, int Fmax, uint8_t* keep_dev, void* stream) {
    if (!h || !seeds_dev || !keep_dev) return failf(VTTS_ERR_INVALID, "null argument");
    if (B <= 0 || Fmax <= 0) return failf(VTTS_ERR_INVALID, "B and Fmax must be positive (got %d, %d)", B, Fmax);
    const int PN = h->cfg.prenet_dim;
    const size_t n = (size_t)B * Fmax * 2 * ((PN + 63) / 64);
    const int blocks = (int)((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535);
    hipLaunchKernelGGL(nat_keep_masks_k, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const unsigned long long*>(seeds_dev),
                       keep_dev, B, Fmax, PN);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "keep-mask launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}
VTTS_API int vtts_nat_acoustic_keep_masks_haiku_mode(const vtts_nat_acoustic* h, uint32_t rng_key0, uint32_t rng_key1, int threefry_partitionable, int B, int Fmax,
                                                     uint8_t* keep_dev, void* stream) {
    if (!h || !keep_dev) return failf(VTTS_ERR_INVALID, "null argument");
    if (B <= 0 || Fmax <= 0) return failf(VTTS_ERR_INVALID, "B and Fmax must be positive (got %d, %d)", B, Fmax);
    if (threefry_partitionable != 0 && threefry_partitionable != 1) return failf(VTTS_ERR_INVALID, "threefry_partitionable must be 0 (classic layout) or 1");
    const int PN = h->cfg.prenet_dim;
    const int threads = 2 * PN < 1024 ? (2 * PN < 64 ? 64 : 2 * PN) : 1024;
    hipLaunchKernelGGL(nat_keep_masks_haiku_k, dim3(Fmax), dim3(threads), 0, static_cast<hipStream_t>(stream), rng_key0, rng_key1, threefry_partitionable,
                       keep_dev, B, Fmax, PN);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "keep-mask launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}
VTTS_API int vtts_nat_acoustic_keep_masks_haiku(const vtts_nat_acoustic* h, uint32_t rng_key0, uint32_t rng_key1, int B, int Fmax, uint8_t* keep_dev,
                                                void* stream) {
    return vtts_nat_acoustic_keep_masks_haiku_mode(h, rng_key0, rng_key1, 0, B, Fmax, keep_dev, stream);
}
#ifdef VTTS_NAT_PERSIST
#include "../../tools/kbench/experiments/nat_persist_host.inc"
#endif
// forward() and forward_groups(): ngroups = 0 is the plain call (the postnet on the caller's stream after the last frame)
// enc_pre: the token encoder's output [B][Lmax][2D] computed ahead by vtts_nat_acoustic_encode() (tokens_dev is not read then), or nullptr
static int nat_acoustic_run(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev, const float* durations_dev,
                            const int32_t* nframes_dev, int B, int Lmax, int Fmax, const uint8_t* keep_dev, float* mel_dev, void* workspace,
                            size_t workspace_bytes, void* stream, int ngroups, const int32_t* group_row0, const int32_t* group_frames,
                            const float* enc_pre = nullptr) {
    if (!h || (!tokens_dev && !enc_pre) || !lengths_dev || !durations_dev || !nframes_dev || !mel_dev) return failf(VTTS_ERR_INVALID, "null argument");
    if (!h->blob) return failf(VTTS_ERR_STATE, "forward() before pack()/bind_packed()");
    size_t need = 0;
    int rc = vtts_nat_acoustic_workspace_bytes(h, B, Lmax, Fmax, &need);
    if (rc) return rc;
    if (!workspace || workspace_bytes < need) return failf(VTTS_ERR_NOMEM, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    if (Lmax > 2048) return failf(VTTS_ERR_INVALID, "at most 2048 tokens per sentence (upsampling weights live in LDS)");
    h->groups_valid = 0;
    if (ngroups > 0) {
        if (!group_row0 || !group_frames) return failf(VTTS_ERR_INVALID, "null argument");
        if (ngroups > 64) return failf(VTTS_ERR_INVALID, "at most 64 groups (got %d)", ngroups);
        if (group_row0[0] != 0 || group_row0[ngroups] != B) return failf(VTTS_ERR_INVALID, "the groups must cover rows [0, %d)", B);
        for (int g = 0; g < ngroups; ++g)
            if (group_row0[g + 1] <= group_row0[g] || group_frames[g] < 1 || group_frames[g] > Fmax)
                return failf(VTTS_ERR_INVALID, "group %d: rows [%d, %d), %d frames (every group needs at least one row and 1 <= frames <= Fmax = %d)", g,
                             group_row0[g], group_row0[g + 1], group_frames[g], Fmax);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int D = h->cfg.encoder_dim, V = h->cfg.vocab_size, H = h->cfg.decoder_dim, PN = h->cfg.prenet_dim, MEL = h->cfg.mel_dim, PD = h->cfg.postnet_dim;
    const int E = 2 * D, G4 = 4 * H;
    // the side stream and the events of the hand-over (created once per handle, on the device the caller made current)
    if (!h->side) {
        HIP_TRYN(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        HIP_TRYN(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRYN(hipEventCreateWithFlags(&h->ev_gates, hipEventDisableTiming));
    }
    while ((int)h->ev_dec.size() < ngroups) {
        hipEvent_t a, b;
        HIP_TRYN(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIP_TRYN(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        h->ev_dec.push_back(a);
        h->ev_done.push_back(b);
    }
    char* p = static_cast<char*>(workspace);
    auto take = [&](size_t bytes) {
        float* r = reinterpret_cast<float*>(p);
        p += align_up(bytes, 256);
        return r;
    };
    float* bufA = take((size_t)B * Lmax * D * 4);
    float* bufB = take((size_t)B * Lmax * D * 4);
    float* enc = take((size_t)B * Lmax * E * 4);
    float* EG1 = take((size_t)B * Lmax * G4 * 4);
    float* EG2 = take((size_t)B * Lmax * G4 * 4);
    float* mel0 = take((size_t)B * Fmax * MEL * 4);
    float* pA = take((size_t)B * Fmax * PD * 4);
    float* pB = take((size_t)B * Fmax * PD * 4);
    float* dstate = take(nat_dec_state_floats(h->cfg, B) * 4);
    float* G1 = take((size_t)B * Fmax * G4 * 4);
    float* G2 = take((size_t)B * Fmax * G4 * 4);
    float* lstm_ws = take(nat_enc_lstm_floats(D, B, Lmax) * 4);
#ifdef VTTS_NAT_PP_EXP
    float4* ppart = reinterpret_cast<float4*>(take(nat_pp_part_floats(MEL, B) * 4));
#endif
    unsigned* ctl = reinterpret_cast<unsigned*>(take(4096));
    if (enc_pre) {
        enc = const_cast<float*>(enc_pre);  // read only from here on
    } else {
        rc = run_token_encoder(*h, "token_encoder/~/", V, D, tokens_dev, lengths_dev, B, Lmax, bufA, bufB, lstm_ws, enc, s);  // model.py:131
        if (rc) return rc;
    }
    // :132 (upsample) lives inside the gates below: cond is never materialised (nat_gates_mix_k)
    HIP_TRYN(hipMemsetAsync(mel_dev, 0, (size_t)B * Fmax * MEL * 4, s));  // rows past a sentence's last frame
    // postnet (:113-121) + residual (:151) of rows [r0, r1) over their first `frames` frames: 4 x (Conv1D(PD, 5) + BatchNorm + tanh),
    // Conv1D(MEL, 5), mel + .   A row's result does not depend on the launch it is part of.
    auto postnet = [&](int r0, int r1, int frames, hipStream_t ps) {
        const float* cur = mel0 + (size_t)r0 * Fmax * MEL;
        float* bufs[2] = {pA + (size_t)r0 * Fmax * PD, pB + (size_t)r0 * Fmax * PD};
        for (int i = 0; i < 5; ++i) {
            const std::string sfx = i ? "_" + std::to_string(i) : "";
            const std::string cv = "conv1_d" + sfx, bn = "batch_norm" + sfx;
            const int cin = i == 0 ? MEL : PD, cout = i == 4 ? MEL : PD, MB = (cout + 31) / 32;
            float* dst = i == 4 ? mel_dev + (size_t)r0 * Fmax * MEL : bufs[i & 1];
            const float4* wpk = reinterpret_cast<const float4*>(h->extra(cv + "#mfma"));
            const float *iv = i < 4 ? h->inv(bn) : nullptr, *mv = i < 4 ? h->dev(bn + "/~/mean_ema", "average") : nullptr, *ov = i < 4 ? h->dev(bn, "offset") : nullptr;
            const int act = i < 4 ? (int)NAT_ACT_TANH : (int)NAT_ACT_NONE;
            const float* res = i == 4 ? mel0 + (size_t)r0 * Fmax * MEL : nullptr;
            const uint4* wx3 = reinterpret_cast<const uint4*>(h->extra(cv + "#x3"));
            if (h->x3 && MB >= 8)
                hipLaunchKernelGGL((nat_conv_x3_k<5, 2>), dim3((frames + 63) / 64, (MB + 7) / 8, r1 - r0), dim3(256), 0, ps, cur, nframes_dev + r0, wx3,
                                   h->dev(cv, "b"), iv, mv, ov, res, dst, Fmax, cin, cout, act, 0);
            else if (h->x3)
                hipLaunchKernelGGL((nat_conv_x3_k<5, 1>), dim3((frames + 63) / 64, (MB + 3) / 4, r1 - r0), dim3(256), 0, ps, cur, nframes_dev + r0, wx3,
                                   h->dev(cv, "b"), iv, mv, ov, res, dst, Fmax, cin, cout, act, 0);
            else if (MB >= 8)
                hipLaunchKernelGGL((nat_conv_mfma_k<5, 2>), dim3((frames + 63) / 64, (MB + 7) / 8, r1 - r0), dim3(256), 0, ps, cur, nframes_dev + r0, wpk,
                                   h->dev(cv, "b"), iv, mv, ov, res, dst, Fmax, cin, cout, act, 0);
            else
                hipLaunchKernelGGL((nat_conv_mfma_k<5, 1>), dim3((frames + 63) / 64, (MB + 3) / 4, r1 - r0), dim3(256), 0, ps, cur, nframes_dev + r0, wpk,
                                   h->dev(cv, "b"), iv, mv, ov, res, dst, Fmax, cin, cout, act, 0);
            cur = dst;
        }
    };
    {  // autoregressive decoder (:134-150): per frame LSTM1, LSTM2, projection + next frame's prenet, all sentences at once
        const int Bp = (B + 63) / 64 * 64, ZW = PN + 2 * H;
        float* Z[2] = {dstate, dstate + (size_t)ZW * Bp};
        float* c1 = dstate + 2 * (size_t)ZW * Bp;
        float* c2 = c1 + (size_t)H * Bp;
        HIP_TRYN(hipMemsetAsync(dstate, 0, nat_dec_state_floats(h->cfg, B) * 4, s));  // frame 0: h1 = h2 = 0, c = 0, prenet(0) = 0 (no biases)
        HIP_TRYN(hipMemsetAsync(mel0, 0, (size_t)B * Fmax * MEL * 4, s));  // rows past a sentence's last frame stay zero
        // the conditioning's share of both layers' gates for every frame (nat_gates_mix_k): the GEMM over the tokens' rows, then the mix for
        // frames [0, 64) here and for the rest beside the first 64 steps
        const int MBG = G4 / 32;
        if (G4 % 1024 != 0) return failf(VTTS_ERR_INVALID, "decoder_dim %d: the gate mix wants 4 * decoder_dim in multiples of 1024", H);
        for (int l = 0; l < 2; ++l) {
            const std::string mod = l ? "lstm_1/linear" : "lstm/linear";
            if (h->x3)
                hipLaunchKernelGGL((nat_conv_x3_k<1, 2>), dim3((Lmax + 63) / 64, (MBG + 7) / 8, B), dim3(256), 0, s, enc, lengths_dev,
                                   reinterpret_cast<const uint4*>(h->extra(mod + "#cond#x3")), h->extra(mod + "#zerob"), nullptr, nullptr, nullptr, nullptr,
                                   l ? EG2 : EG1, Lmax, E, G4, (int)NAT_ACT_NONE, 0);
            else
                hipLaunchKernelGGL((nat_conv_mfma_k<1, 2>), dim3((Lmax + 63) / 64, (MBG + 7) / 8, B), dim3(256), 0, s, enc, lengths_dev,
                                   reinterpret_cast<const float4*>(h->extra(mod + "#cond")), h->extra(mod + "#zerob"), nullptr, nullptr, nullptr, nullptr,
                                   l ? EG2 : EG1, Lmax, E, G4, (int)NAT_ACT_NONE, 0);
        }
        const size_t mlds = ((size_t)(Lmax + 3) / 4 * 4 + (size_t)Lmax * NAT_MIX_FT) * sizeof(float);
        if (mlds > 48 * 1024)
            HIP_TRYN(hipFuncSetAttribute(reinterpret_cast<const void*>(&nat_gates_mix_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
        const int mtiles = (Fmax + NAT_MIX_FT - 1) / NAT_MIX_FT, mfirst = 64 / NAT_MIX_FT < mtiles ? 64 / NAT_MIX_FT : mtiles;
        auto gates = [&](int tile0, int ntiles, hipStream_t gs) {
            hipLaunchKernelGGL(nat_gates_mix_k, dim3(ntiles, 2 * (G4 / 1024), B), dim3(256), mlds, gs, EG1, EG2, h->extra("lstm/linear#condb"),
                               h->extra("lstm_1/linear#condb"), lengths_dev, durations_dev, nframes_dev, G1, G2, Lmax, Fmax, G4, tile0);
        };
        gates(0, mfirst, s);
        if (mtiles > mfirst) {
            HIP_TRYN(hipEventRecord(h->ev_fork, s));
            HIP_TRYN(hipStreamWaitEvent(h->side, h->ev_fork, 0));
            gates(mfirst, mtiles - mfirst, h->side);
            HIP_TRYN(hipEventRecord(h->ev_gates, h->side));
        }
        const float4* w1 = reinterpret_cast<const float4*>(h->extra("lstm/linear#mfma"));
        const float4* w2 = reinterpret_cast<const float4*>(h->extra("lstm_1/linear#mfma"));
        const float4* f1 = reinterpret_cast<const float4*>(h->extra("linear_1#k4"));
        const float4* f2 = reinterpret_cast<const float4*>(h->extra("linear_2#k4"));
        const float4* wp = reinterpret_cast<const float4*>(h->extra("linear#k4"));
        const float* bp = h->dev("linear", "b");
        const bool wide = B > 32;  // two 32-sentence tiles per wave once there are that many sentences
        const dim3 lgrid(H / 8, wide ? Bp / 64 : 1);
        auto lstm = [&](const float* inA, int KA, const float* inB, const float4* w, const float* gin, float* cst, float* hout, int f) {
            const NatLstmOps o{inA, inB, w, nullptr, cst, hout, gin + (size_t)f * G4, (size_t)Fmax * G4};
            if (wide) hipLaunchKernelGGL((nat_dec_lstm_k<2, 8>), lgrid, dim3(512), 0, s, o, o, KA, H, nframes_dev, f, B, Bp, H);
            else hipLaunchKernelGGL((nat_dec_lstm_k<1, 8>), lgrid, dim3(512), 0, s, o, o, KA, H, nframes_dev, f, B, Bp, H);
        };
        const size_t plds = ((size_t)2 * H + 1024 + MEL + PN) * sizeof(float4);
#ifdef VTTS_NAT_PP_EXP  // the projection + prenet step cut along its weights (round 6 experiment; option "pp_split")
        const bool pp_split = h->pp_split && nat_pp_split_ok(PN, H, MEL);
        const dim3 pgridA(NAT_PP_NSL, (B + NAT_PP_TS - 1) / NAT_PP_TS), pgridB(PN / 32, (B + NAT_PP_TS - 1) / NAT_PP_TS);
        const int pthreadsA = 640;
        const size_t pldsA = (size_t)128 * 8 * sizeof(float4), pldsB = ((size_t)80 * 8 + (size_t)256 * 8 + 2 * 8 * 32) * sizeof(float4);
#endif
        auto group_handover = [&](int frames_done) -> int {  // a group whose last frame was frames_done - 1: its postnet starts now, on the side stream
            for (int g = 0; g < ngroups; ++g) {
                if (group_frames[g] != frames_done) continue;
                HIP_TRYN(hipEventRecord(h->ev_dec[g], s));
                HIP_TRYN(hipStreamWaitEvent(h->side, h->ev_dec[g], 0));
                postnet(group_row0[g], group_row0[g + 1], group_frames[g], h->side);
                HIP_TRYN(hipEventRecord(h->ev_done[g], h->side));
            }
            return VTTS_OK;
        };
        bool persist = false;
#ifdef VTTS_NAT_PERSIST
#include "../../tools/kbench/experiments/nat_persist_launch.inc"
#endif
        // option "bf16x3": the split-state step (nat_dec_lstm_x3_k) where its 16-row steps divide the row blocks; the state's two parities hold
        // two bf16 planes each (the same bytes as the fp32 rows)
        const bool dx3 = h->x3 && PN % 16 == 0 && H % 16 == 0 && ((PN + H) / 16) % 8 == 0 && ((PN + 2 * H) / 16) % 8 == 0;
        const size_t zplane = (size_t)ZW * Bp;  // bf16 elements per plane
        const uint4* w1x = reinterpret_cast<const uint4*>(h->extra("lstm/linear#x3"));
        const uint4* w2x = reinterpret_cast<const uint4*>(h->extra("lstm_1/linear#x3"));
        auto lstm_x3 = [&](unsigned short* zcx, const unsigned short* zpx, int KA, int K, const uint4* w, const float* gin, float* cst, int out_row0, int f) {
            const NatLstmX3Ops o{zcx, zpx, zplane, w, gin + (size_t)f * G4, (size_t)Fmax * G4, cst, zcx, out_row0};
            if (wide) hipLaunchKernelGGL((nat_dec_lstm_x3_k<2, 8>), lgrid, dim3(512), 0, s, o, KA, K, nframes_dev, f, B, Bp, H);
            else hipLaunchKernelGGL((nat_dec_lstm_x3_k<1, 8>), lgrid, dim3(512), 0, s, o, KA, K, nframes_dev, f, B, Bp, H);
        };
        for (int f = 0; f < Fmax && !persist; ++f) {
            if (f == 64) HIP_TRYN(hipStreamWaitEvent(s, h->ev_gates, 0));
            float* zc = Z[f & 1];
            float* zp = Z[(f + 1) & 1];
            if (dx3) {
                unsigned short* zcx = reinterpret_cast<unsigned short*>(zc);
                const unsigned short* zpx = reinterpret_cast<const unsigned short*>(zp);
                lstm_x3(zcx, zpx, PN, PN + H, w1x, G1, c1, PN, f);
                lstm_x3(zcx, zpx, PN + H, PN + 2 * H, w2x, G2, c2, PN + H, f);
#ifdef VTTS_NAT_PP_EXP
                if (pp_split) {
                    hipLaunchKernelGGL(nat_dec_proj_part_k<true>, pgridA, dim3(pthreadsA), pldsA, s, zc, nframes_dev, wp, ppart, f, B, Bp, PN, H, MEL, zplane);
                    hipLaunchKernelGGL(nat_dec_prenet_k<true>, pgridB, dim3(512), pldsB, s, zp, nframes_dev, f1, f2, ppart, bp, keep_dev, mel0, f, B, Bp, Fmax, PN, MEL,
                                       zplane);
                } else
#endif
                    hipLaunchKernelGGL(nat_dec_proj_prenet_k<true>, dim3((B + 3) / 4), dim3(1024), plds, s, zc, zp, nframes_dev, f1, f2, wp, bp, keep_dev, mel0, f,
                                       B, Bp, Fmax, PN, H, MEL, zplane);
                rc = group_handover(f + 1);
                if (rc) return rc;
                continue;
            }
            lstm(zc, PN, zp + (size_t)PN * Bp, w1, G1, c1, zc + (size_t)PN * Bp, f);
            lstm(zc, PN + H, zp + (size_t)(PN + H) * Bp, w2, G2, c2, zc + (size_t)(PN + H) * Bp, f);
#ifdef VTTS_NAT_PP_EXP
            if (pp_split) {
                hipLaunchKernelGGL(nat_dec_proj_part_k<false>, pgridA, dim3(pthreadsA), pldsA, s, zc, nframes_dev, wp, ppart, f, B, Bp, PN, H, MEL, (size_t)0);
                hipLaunchKernelGGL(nat_dec_prenet_k<false>, pgridB, dim3(512), pldsB, s, zp, nframes_dev, f1, f2, ppart, bp, keep_dev, mel0, f, B, Bp, Fmax, PN, MEL,
                                   (size_t)0);
            } else
#endif
                hipLaunchKernelGGL(nat_dec_proj_prenet_k<false>, dim3((B + 3) / 4), dim3(1024), plds, s, zc, zp, nframes_dev, f1, f2, wp, bp, keep_dev, mel0, f, B,
                                   Bp, Fmax, PN, H, MEL, (size_t)0);
            rc = group_handover(f + 1);  // under the remaining decoder steps
            if (rc) return rc;
        }
    }
    if (ngroups > 0) {
        for (int g = 0; g < ngroups; ++g) HIP_TRYN(hipStreamWaitEvent(s, h->ev_done[g], 0));  // stream order for the caller: mel_dev is complete after this call on `s`
        h->groups_valid = ngroups;
    } else {
        postnet(0, B, Fmax, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "acoustic model launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}

VTTS_API int vtts_nat_acoustic_forward(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev, const float* durations_dev,
                                       const int32_t* nframes_dev, int B, int Lmax, int Fmax, const uint8_t* keep_dev, float* mel_dev,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    return nat_acoustic_run(h, tokens_dev, lengths_dev, durations_dev, nframes_dev, B, Lmax, Fmax, keep_dev, mel_dev, workspace, workspace_bytes, stream, 0,
                            nullptr, nullptr);
}

VTTS_API int vtts_nat_acoustic_forward_groups(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev, const float* durations_dev,
                                              const int32_t* nframes_dev, int B, int Lmax, int Fmax, const uint8_t* keep_dev, float* mel_dev,
                                              void* workspace, size_t workspace_bytes, void* stream, int ngroups, const int32_t* group_row0,
                                              const int32_t* group_frames) {
    if (ngroups < 1) return failf(VTTS_ERR_INVALID, "forward_groups() needs at least one group (got %d)", ngroups);
    return nat_acoustic_run(h, tokens_dev, lengths_dev, durations_dev, nframes_dev, B, Lmax, Fmax, keep_dev, mel_dev, workspace, workspace_bytes, stream,
                            ngroups, group_row0, group_frames);
}

// The token encoder alone, ahead of forward_from_encoder(): it needs the tokens only, so a pipeline can run it while the host still turns the
// duration model's output into frame counts.  A row's output does not depend on its batch: rows may be re-ordered / dropped before
// forward_from_encoder() (whose B and row order are its own; Lmax must be this call's).  Workspace: workspace_bytes(h, B, Lmax, 1) suffice.
VTTS_API int vtts_nat_acoustic_encode(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev, int B, int Lmax, float* enc_dev,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !tokens_dev || !lengths_dev || !enc_dev) return failf(VTTS_ERR_INVALID, "null argument");
    if (!h->blob) return failf(VTTS_ERR_STATE, "encode() before pack()/bind_packed()");
    size_t need = 0;
    int rc = vtts_nat_acoustic_workspace_bytes(h, B, Lmax, 1, &need);
    if (rc) return rc;
    if (!workspace || workspace_bytes < need) return failf(VTTS_ERR_NOMEM, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    const int D = h->cfg.encoder_dim, V = h->cfg.vocab_size;
    char* p = static_cast<char*>(workspace);
    auto take = [&](size_t bytes) {
        float* r = reinterpret_cast<float*>(p);
        p += align_up(bytes, 256);
        return r;
    };
    float* bufA = take((size_t)B * Lmax * D * 4);
    float* bufB = take((size_t)B * Lmax * D * 4);
    float* lstm_ws = take(nat_enc_lstm_floats(D, B, Lmax) * 4);  // (fits: the full layout holds the same three buffers and more)
    rc = run_token_encoder(*h, "token_encoder/~/", V, D, tokens_dev, lengths_dev, B, Lmax, bufA, bufB, lstm_ws, enc_dev, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(VTTS_ERR_HIP, "token encoder launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}
// forward() / forward_groups() (ngroups = 0: no hand-over) from an encoder output computed by vtts_nat_acoustic_encode(): the same mel, bit for bit
VTTS_API int vtts_nat_acoustic_forward_from_encoder(vtts_nat_acoustic* h, const float* enc_dev, const int32_t* lengths_dev, const float* durations_dev,
                                                    const int32_t* nframes_dev, int B, int Lmax, int Fmax, const uint8_t* keep_dev, float* mel_dev,
                                                    void* workspace, size_t workspace_bytes, void* stream, int ngroups, const int32_t* group_row0,
                                                    const int32_t* group_frames) {
    if (!enc_dev) return failf(VTTS_ERR_INVALID, "null argument");
    if (ngroups < 0) return failf(VTTS_ERR_INVALID, "ngroups must be >= 0 (got %d)", ngroups);
    return nat_acoustic_run(h, nullptr, lengths_dev, durations_dev, nframes_dev, B, Lmax, Fmax, keep_dev, mel_dev, workspace, workspace_bytes, stream, ngroups,
                            group_row0, group_frames, enc_dev);
}

VTTS_API int vtts_nat_acoustic_wait_group(vtts_nat_acoustic* h, int group, void* stream) {
    if (!h) return failf(VTTS_ERR_INVALID, "null argument");
    if (group < 0 || group >= h->groups_valid) return failf(VTTS_ERR_STATE, "wait_group(%d): the last forward_groups() call had %d groups", group, h->groups_valid);
    HIP_TRYN(hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_done[group], 0));
    return VTTS_OK;
}
