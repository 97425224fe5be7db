// Host-side engine behind the C ABI of include/vtts_hifigan.h.
//
// Holds the execution plan of the HiFi-GAN generator (vietTTS/hifigan/model.py:78-125): the 78
// convolution modules in execution order, where each one's weights live in the packed device blob,
// and the schedule that fuses LeakyReLU / bias / residual / MRF mean / tanh into the convolution
// kernels.  All device memory is caller-owned (blob, workspace, mel, wav).
#include "../../include/vtts_hifigan.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vtts_internal.h"

using namespace vtts;

#define VTTS_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace

// shared with nat.hip: record the calling thread's last error, return the status code
int vtts::set_error(int code, const char* msg) {
    g_last_error = msg;
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(VTTS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

enum Kind { KIND_CONV = 0, KIND_CONVT = 1 };

struct Layer {
    std::string key;
    int kind = KIND_CONV;
    int cin = 0, cout = 0, k = 0, dil = 1, stride = 1;
    int pad = 0;    // conv: symmetric pad, get_padding (model.py:8-10)
    int pad_a = 0;  // convT: left pad of the zero-stuffed input (lax "SAME")
    // host copies (Haiku layout)
    std::vector<float> w, b;
    bool have_w = false, have_b = false;
    // packed blob offsets in bytes
    size_t off_w = 0, off_b = 0, off_wp = 0;
    bool has_wp = false;
    size_t wp_floats = 0;
    // bf16 path: kernel class, packed bf16 weights, bias expanded to the GEMM's row count
    int bcls = BCLS_NONE;
    size_t off_wb = 0, wb_bytes = 0;
    int coutp = 0;  // GEMM rows: cout, or stride*cout for a transposed convolution run as Conv1d(k=3)
    // fused pair (this layer = convs1_z, next layer = convs2_z): [c1 slabs][c2 slabs] + [b1][b2]
    bool has_pair = false;
    size_t off_pw = 0, pw_bytes = 0, off_pb = 0;
    // whole ResBlock (this layer = convs1_0 of a ResBlock with a fused kernel): six convolutions' fragments + six biases
    bool has_rb = false;
    size_t off_rw = 0, rw_bytes = 0, off_rb = 0;
    // transposed convolution on the register-streamed kernel (kernels_bf16_up.hip): a second packing of the 3-tap form
    bool has_ug = false;
    size_t off_ug = 0, ug_bytes = 0;
    // VTTS_BF16X3: this ResBlock convolution's split weights, [hi fragments][lo fragments] (kernels_x3.hip: pair_x3_pack)
    bool has_x3 = false;
    size_t off_x3 = 0;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct vtts_hifigan {
    vtts_hifigan_cfg cfg;
    int device = 0;
    int dtype = VTTS_F32;
    bool x3 = false;                 // VTTS_BF16X3: dtype stays VTTS_F32 (the fp32 engine's layouts and schedule), the ResBlock pairs run on kernels_x3.hip
    int hop = 1;
    std::vector<Layer> layers;       // execution order
    int idx_pre = -1, idx_post = -1;
    std::vector<int> idx_ups;        // per stage
    std::vector<int> idx_res;        // [stage][kernel][z][convs1|convs2] flattened: base index of each resblock
    size_t blob_bytes = 0;
    char* blob = nullptr;            // bound device blob (caller-owned)
    // options
    int64_t opt_kernels = 0;         // 0 auto, 1 generic only
    int64_t opt_microbatch = 0;      // 0 auto
    int64_t opt_profile = 0;
    int cur_b0 = 0;                 // first utterance of the micro-batch being launched
    const int* cur_lens = nullptr;  // ragged forward in flight: valid mel frames per utterance (device) ...
    int cur_T = 0;                  // ... of the T allocated
    int64_t opt_tiles = 0;           // 0 auto, 1 wide, 2 narrow
    int64_t opt_fuse = 2;            // bf16: 0 one kernel per convolution, 1 fused pairs, 2 fused pairs + whole ResBlocks at C = 32
    int64_t opt_streams = 0;         // micro-batches in flight on separate HIP streams (1..4; 0 = the engines' default: 2 — auto_streams)
    int64_t opt_graph = 1;           // small launches: replay a captured hipGraph once the same buffers were seen twice (0 = always eager)
    struct GraphEntry {
        const void* mel = nullptr;
        void* wav = nullptr;
        void* ws = nullptr;
        int B = 0, T = 0, seen = 0;
        uint64_t epoch = 0, last_use = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
    };
    std::vector<GraphEntry> graphs;  // at most GRAPH_SLOTS, least recently used evicted
    hipStream_t cap_stream = nullptr; // launches are recorded on this stream, graphs are launched on the caller's
    uint64_t epoch = 0, use_clock = 0;  // epoch: bumped by everything a captured launch sequence bakes in (options, the weight blob)
    int64_t opt_zigzag = 1;          // consecutive launches walk the batch in alternating directions (see next_zrev)
    unsigned zrev_count = 0;
    int64_t opt_tail = 1;            // bf16: conv_post + tanh inside the generator's last pair launch (0 = the separate streaming kernel)
    int64_t opt_stage = 0;           // bf16: 1 = the whole last stage (three ResBlocks, MRF mean, conv_post, tanh) in ONE launch (kernels_bf16_stage.hip: bit-identical, measured
                                     // 8.0-8.5 ms against 6.2 ms for the launches it replaces at 64 x 1024 frames: profiles/r06_a_stage_kernel_findings.md); 0 (default) = a launch per ResBlock / pair
    int64_t opt_chains = 1;          // small launches: the MRF's ResBlocks of a stage on parallel streams (0 = one after the other, 2 = always)
    hipEvent_t ev_chain[3] = {nullptr, nullptr, nullptr};  // bf16: ResBlock j's output is in the shared accumulator (orders the accumulating epilogues)
    hipStream_t side_streams[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // profiling of the dominant kernel class
    int prof_C = 0, prof_K = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    double prof_flops = 0.0;
    std::string prof_name, prof_name_pair;
};

namespace {

int conv_same_pad_a(int k, int s) {
    // lax.conv_transpose padding="SAME": pad_len = k + s - 2; pad_a = k-1 if s > k-1 else ceil(pad_len/2)
    const int pad_len = k + s - 2;
    return (s > k - 1) ? (k - 1) : (pad_len + 1) / 2;
}

int classify_bf16(const vtts_hifigan* h, const Layer& l, bool is_pre, bool is_post) {
    if (is_post) return (l.cin == 32 && l.cout == 1 && l.k == 7) ? BCLS_NONE - 1 : BCLS_NONE;  // -2 = streaming conv_post
    if (is_pre) return (l.cin <= 128 && l.cin % 8 == 0 && l.cout == 512 && l.k == 7) ? BCLS_PRE : BCLS_NONE;
    if (l.kind == KIND_CONVT) {
        if (!convT1d_f32_mfma_supported(l.cin, l.cout, l.k, l.stride, l.pad_a, 4)) return BCLS_NONE;  // same polyphase condition
        if (l.cin == 512) return BCLS_UP0;
        if (l.cin == 256) return BCLS_UP1;
        if (l.cin == 128) return BCLS_UP2;
        return BCLS_UP3;
    }
    if (l.cin != l.cout || !(l.k == 3 || l.k == 7 || l.k == 11) || l.dil > 5) return BCLS_NONE;
    switch (l.cin) {
        case 256: return BCLS_RES256;
        case 128: return BCLS_RES128;
        case 64: return BCLS_RES64;
        case 32: return BCLS_RES32;
    }
    return BCLS_NONE;
}

int build_layers_bf16(vtts_hifigan* h) {
    size_t off = 0;
    double best_flops = -1.0;
    for (auto& l : h->layers) {
        const bool is_pre = (&l == &h->layers[h->idx_pre]), is_post = (&l == &h->layers[h->idx_post]);
        l.bcls = classify_bf16(h, l, is_pre, is_post);
        if (l.bcls == BCLS_NONE)
            return fail(VTTS_ERR_INVALID, "dtype bf16: no kernel for module %s (%d->%d, k=%d): the bf16 path covers the HiFi-GAN V1 shapes",
                        l.key.c_str(), l.cin, l.cout, l.k);
        l.coutp = (l.kind == KIND_CONVT) ? l.cout * l.stride : l.cout;
        l.off_b = off;
        off = align_up(off + (size_t)l.coutp * sizeof(float), 256);
        if (is_post) {
            l.off_w = off;
            off = align_up(off + (size_t)l.k * l.cin * l.cout * sizeof(float), 256);
        } else {
            const BPackGeom g = bf16_pack_geom(l.bcls, l.kind == KIND_CONVT ? 3 : l.k);
            l.wb_bytes = bf16_packed_bytes(g);
            l.off_wb = off;
            off = align_up(off + l.wb_bytes, 256);
            if (l.kind == KIND_CONVT || (is_pre && l.bcls == BCLS_PRE && l.cin == 80)) {  // the register-streamed kernel's packing (kernels_bf16_up.hip; its conv_pre tile is the 80-mel one)
                l.has_ug = true;
                l.ug_bytes = bf16_packed_bytes(convt_g_pack_geom(l.bcls));
                l.off_ug = off;
                off = align_up(off + l.ug_bytes, 256);
            }
            if (l.kind == KIND_CONV && l.cin == l.cout) {
                double fl = 0.0;
                long len2 = 1;
                for (auto& m : h->layers) {
                    if (m.kind == KIND_CONVT) len2 *= m.stride;
                    if (m.kind == KIND_CONV && m.cin == l.cin && m.k == l.k && m.cin == m.cout) fl += 2.0 * len2 * m.cin * m.cout * m.k;
                }
                if (fl > best_flops) {
                    best_flops = fl;
                    h->prof_C = l.cin;
                    h->prof_K = l.k;
                }
            }
        }
    }
    // (ResBlock2 generators, model.py:54-74, have no fused packings: their two convolutions per block run on the per-convolution kernel)
    for (size_t r = 0; r < h->idx_res.size() && h->cfg.resblock != 2; ++r) {
        for (int z = 0; z < 3; ++z) {
            Layer& c1 = h->layers[h->idx_res[r] + 2 * z];
            const Layer& c2 = h->layers[h->idx_res[r] + 2 * z + 1];
            if (!pair_bf16_supported(c1.cin, c1.k, c1.dil) || c2.dil != 1 || c2.k != c1.k) continue;
            c1.has_pair = true;
            c1.pw_bytes = 2 * bf16_packed_bytes(pair_pack_geom(c1.cin, c1.k));
            c1.off_pw = off;
            off = align_up(off + c1.pw_bytes, 256);
            c1.off_pb = off;
            off = align_up(off + (size_t)2 * c1.cin * sizeof(float), 256);
        }
    }
    for (size_t r = 0; r < h->idx_res.size() && h->cfg.resblock != 2; ++r) {
        Layer& c0 = h->layers[h->idx_res[r]];
        const int dils[3] = {h->layers[h->idx_res[r] + 0].dil, h->layers[h->idx_res[r] + 2].dil, h->layers[h->idx_res[r] + 4].dil};
        bool ok = resblock_bf16_supported(c0.cin, c0.k, dils);
        for (int q = 0; q < 6 && ok; ++q) {
            const Layer& l = h->layers[h->idx_res[r] + q];
            ok = l.cin == c0.cin && l.cout == c0.cin && l.k == c0.k && ((q & 1) ? l.dil == 1 : true);
        }
        if (!ok) continue;
        c0.has_rb = true;
        c0.rw_bytes = 6 * bf16_packed_bytes(pair_g_pack_geom(c0.cin, c0.k));
        c0.off_rw = off;
        off = align_up(off + c0.rw_bytes, 256);
        c0.off_rb = off;
        off = align_up(off + (size_t)6 * c0.cin * sizeof(float), 256);
    }
    h->blob_bytes = off;
    for (auto& l : h->layers)
        if (l.kind == KIND_CONV && l.cin == h->prof_C && l.cout == h->prof_C && l.k == h->prof_K) h->prof_name = bf16_kernel_name(l.bcls, l.k);
    h->prof_name_pair = pair_kernel_name(h->prof_C, h->prof_K);
    return VTTS_OK;
}

int build_layers(vtts_hifigan* h) {
    const vtts_hifigan_cfg& c = h->cfg;
    auto add = [&](const std::string& key, int kind, int cin, int cout, int k, int dil, int stride) {
        Layer l;
        l.key = key;
        l.kind = kind;
        l.cin = cin;
        l.cout = cout;
        l.k = k;
        l.dil = dil;
        l.stride = stride;
        if (kind == KIND_CONV) l.pad = (k * dil - dil) / 2;
        else l.pad_a = conv_same_pad_a(k, stride);
        h->layers.push_back(l);
        return (int)h->layers.size() - 1;
    };
    const int c0 = c.upsample_initial_channel;
    h->idx_pre = add("generator/~/conv1_d", KIND_CONV, c.num_mels, c0, 7, 1, 1);
    int n = 0;
    h->hop = 1;
    for (int i = 0; i < c.num_upsamples; ++i) {
        const int cin = c0 >> i, cout = c0 >> (i + 1);
        h->idx_ups.push_back(add("generator/~/ups_" + std::to_string(i), KIND_CONVT, cin, cout,
                                 c.upsample_kernel_sizes[i], 1, c.upsample_rates[i]));
        h->hop *= c.upsample_rates[i];
        for (int j = 0; j < c.num_kernels; ++j) {
            const std::string base = "generator/~/res_block1_" + std::to_string(n) + "/~/";
            int first = -1;
            if (c.resblock == 2) {
                // ResBlock2 (model.py:54-74): two convolutions, default hk.Conv1D names inside the module named res_block1_N (model.py:105)
                for (int z = 0; z < 2; ++z) {
                    int i1 = add(base + (z == 0 ? std::string("conv1_d") : "conv1_d_" + std::to_string(z)), KIND_CONV, cout, cout,
                                 c.resblock_kernel_sizes[j], c.resblock_dilation_sizes[j][z], 1);
                    if (z == 0) first = i1;
                }
                h->idx_res.push_back(first);  // layers first, first + 1
                ++n;
                continue;
            }
            for (int z = 0; z < 3; ++z) {
                int i1 = add(base + "convs1_" + std::to_string(z), KIND_CONV, cout, cout, c.resblock_kernel_sizes[j],
                             c.resblock_dilation_sizes[j][z], 1);
                int i2 = add(base + "convs2_" + std::to_string(z), KIND_CONV, cout, cout, c.resblock_kernel_sizes[j], 1, 1);
                (void)i2;
                if (z == 0) first = i1;
            }
            h->idx_res.push_back(first);  // layers first..first+5 = c1_0, c2_0, c1_1, c2_1, c1_2, c2_2
            ++n;
        }
    }
    h->idx_post = add("generator/~/conv1_d_1", KIND_CONV, c0 >> c.num_upsamples, 1, 7, 1, 1);

    if (h->dtype == VTTS_BF16) return build_layers_bf16(h);

    // blob layout: per layer plain weights, bias, optional MFMA-packed weights; 256-B aligned
    size_t off = 0;
    double best_flops = -1.0;
    for (auto& l : h->layers) {
        l.off_w = off;
        off = align_up(off + (size_t)l.k * l.cin * l.cout * sizeof(float), 256);
        l.off_b = off;
        off = align_up(off + (size_t)l.cout * sizeof(float), 256);
        const bool is_pre = (&l == &h->layers[h->idx_pre]);
        if (h->dtype == VTTS_F32 && l.kind == KIND_CONV && conv1d_f32_mfma_supported(l.cin, l.cout, l.k, l.dil, 4, is_pre)) {
            l.has_wp = true;
            l.wp_floats = conv1d_f32_mfma_packed_floats(l.cin, l.cout, l.k);
            l.off_wp = off;
            off = align_up(off + l.wp_floats * sizeof(float), 256);
            if (!is_pre) {
                // dominant kernel class = the ResBlock (C, K) with the most FLOPs per mel frame
                double fl = 0.0;
                long len2 = 1;
                for (auto& m : h->layers) {
                    if (m.kind == KIND_CONVT) len2 *= m.stride;
                    if (m.kind == KIND_CONV && m.cin == l.cin && m.k == l.k && m.cin == m.cout) fl += 2.0 * len2 * m.cin * m.cout * m.k;
                }
                if (fl > best_flops) {
                    best_flops = fl;
                    h->prof_C = l.cin;
                    h->prof_K = l.k;
                }
            }
        } else if (h->dtype == VTTS_F32 && l.kind == KIND_CONVT &&
                   convT1d_f32_mfma_supported(l.cin, l.cout, l.k, l.stride, l.pad_a, 4)) {
            l.has_wp = true;
            l.wp_floats = convT1d_f32_mfma_packed_floats(l.cin, l.cout, l.k);
            l.off_wp = off;
            off = align_up(off + l.wp_floats * sizeof(float), 256);
        }
    }
    if (h->x3) {
        Layer& pre = h->layers[h->idx_pre];
        if (conv_pre_x3_supported(pre.cin, pre.cout, pre.k, pre.dil)) {  // conv_pre on the split route too (round 5)
            pre.has_x3 = true;
            pre.off_x3 = off;
            off = align_up(off + conv_pre_x3_bytes(), 256);
        }
        for (int i : h->idx_ups) {
            Layer& l = h->layers[i];
            if (!convt_x3_supported(l.cin, l.cout, l.k, l.stride, l.pad_a, 4)) continue;
            l.has_x3 = true;
            l.off_x3 = off;
            off = align_up(off + convt_x3_bytes(l.cin, l.cout, l.stride), 256);
        }
    }
    if (h->x3 && h->cfg.resblock != 2) {
        for (size_t r = 0; r < h->idx_res.size(); ++r)
            for (int q = 0; q < 6; ++q) {
                Layer& l = h->layers[h->idx_res[r] + q];
                if (l.cin != l.cout || !pair_x3_supported(l.cin, l.k, l.dil, 4)) continue;
                l.has_x3 = true;
                l.off_x3 = off;
                off = align_up(off + pair_x3_conv_bytes(l.cin, l.k), 256);
            }
    }
    h->blob_bytes = off;
    if (h->prof_C) h->prof_name = conv1d_f32_mfma_kernel_name(h->prof_C, h->prof_K);
    if (h->prof_C && h->x3) {
        char buf[96];
        snprintf(buf, sizeof(buf), "resblock_pair_x3_k<XTile<%d, %d,", h->prof_C, h->prof_K);
        h->prof_name = buf;
    }
    return VTTS_OK;
}

Layer* find_layer(vtts_hifigan* h, const char* key) {
    for (auto& l : h->layers)
        if (l.key == key) return &l;
    return nullptr;
}

// ---- one convolution launch -------------------------------------------------------------------
struct Act {  // channel-major activation view
    const float* p;
    long sb, sc, st;
};

// A launch's output (up to 1.07 GB at B = 64 x T = 1024) is the next launch's input, and the last 256 MB written are still in the Infinity
// Cache when that launch starts: walking the batch (blockIdx.z) in the opposite direction makes the consumer start with them instead of
// with the utterances written first, which have long been evicted (bf16 pass 40.00 -> 39.87 ms, three interleaved repetitions,
// gpurun_out/r03_exp40; the fp32 MFMA kernels take the same flag).  The samples do not depend on the order: utterances are independent.
int next_zrev(vtts_hifigan* h) { return h->opt_zigzag ? (int)(h->zrev_count++ & 1) : 0; }

// ragged batches on the fp32 / bf16x3 engines (vtts_hifigan_forward_ragged): a layer whose input has L columns per utterance slot learns each
// utterance's valid columns = frames * (columns per frame) (device_common.h: valid_len)
void set_ragged(const vtts_hifigan* h, ConvArgs& a, int L) {
    a.lens = h->cur_lens ? h->cur_lens + h->cur_b0 : nullptr;
    a.len_mul = h->cur_lens ? L / h->cur_T : 1;
}

int run_layer(vtts_hifigan* h, const Layer& l, Act x, int B, int L, float slope_in, const float* res, float* y,
              int acc_mode, float div, int tanh_out, float* pre_act, hipStream_t s) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x.p;
    a.x_sb = x.sb;
    a.x_sc = x.sc;
    a.x_st = x.st;
    a.w = reinterpret_cast<const float*>(h->blob + l.off_w);
    a.wp = l.has_wp ? (h->blob + l.off_wp) : nullptr;
    a.bias = reinterpret_cast<const float*>(h->blob + l.off_b);
    a.res = res;
    a.y = y;
    a.B = B;
    a.Cin = l.cin;
    a.Cout = l.cout;
    a.K = l.k;
    a.dil = l.dil;
    a.pad = l.pad;
    a.stride = l.stride;
    a.pad_a = l.pad_a;
    a.L = L;
    a.Lout = (l.kind == KIND_CONVT) ? L * l.stride : L;
    a.slope_in = slope_in;
    a.acc_mode = acc_mode;
    a.div = div;
    a.tanh_out = tanh_out;
    a.pre_act = pre_act;
    a.tile_pref = (int)h->opt_tiles;
    a.zrev = next_zrev(h);
    set_ragged(h, a, L);

    hipError_t e;
    const bool ncw = x.st == 1 && x.sc == L && (x.sb % 4) == 0;
    const bool want_mfma = h->opt_kernels == 0 && l.has_wp;
    if (l.kind == KIND_CONVT) {
        if (h->x3 && l.has_x3 && h->opt_kernels == 0 && h->opt_fuse >= 1 && ncw && !res && acc_mode == ACC_STORE) {
            a.wp = h->blob + l.off_x3;  // VTTS_BF16X3: the transposed convolutions on the bf16 matrix pipe with split operands too (kernels_x3.hip)
            e = launch_convt_x3(a, s);
        } else if (want_mfma && ncw && !res && acc_mode == ACC_STORE && convT1d_f32_mfma_supported(l.cin, l.cout, l.k, l.stride, l.pad_a, L))
            e = launch_convT1d_f32_mfma(a, s);
        else
            e = launch_convT1d_generic(a, s);
    } else if (tanh_out) {
        if (h->opt_kernels == 0 && ncw && !res && acc_mode == ACC_STORE && conv_post_fast_supported(l.cin, l.cout, l.k, L))
            e = launch_conv_post_fast(a, s);
        else
            e = launch_conv1d_generic(a, s);
    } else {
        const bool nwc = x.sc == 1 && x.st == l.cin;
        if (h->x3 && l.has_x3 && nwc && h->opt_kernels == 0 && h->opt_fuse >= 1 && !res && acc_mode == ACC_STORE && &l == &h->layers[h->idx_pre] &&
            (reinterpret_cast<uintptr_t>(x.p) & 15) == 0) {  // (float4 row loads: a mel pointer that is not 16-byte aligned takes the fp32 kernel)
            a.wp = h->blob + l.off_x3;  // VTTS_BF16X3: conv_pre with split operands (kernels_x3.hip: conv_pre_x3_k)
            const hipError_t ex = launch_conv_pre_x3(a, s);
            if (ex != hipSuccess) return fail(VTTS_ERR_HIP, "kernel launch for %s failed: %s", l.key.c_str(), hipGetErrorString(ex));
            return VTTS_OK;
        }
        const bool mfma = want_mfma && (ncw || nwc) && conv1d_f32_mfma_supported(l.cin, l.cout, l.k, l.dil, L, nwc);
        if (mfma) {
            const bool prof = h->opt_profile && l.cin == h->prof_C && l.cout == h->prof_C && l.k == h->prof_K;
            if (prof) {
                if (h->prof_used == h->prof_events.size()) {
                    hipEvent_t e0, e1;
                    HIP_TRY(hipEventCreate(&e0));
                    HIP_TRY(hipEventCreate(&e1));
                    h->prof_events.emplace_back(e0, e1);
                }
                HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].first, s));
            }
            e = launch_conv1d_f32_mfma(a, s);
            if (prof) {
                HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].second, s));
                h->prof_used++;
                h->prof_flops += 2.0 * (double)B * L * l.cin * l.cout * l.k;
            }
        } else {
            e = launch_conv1d_generic(a, s);
        }
    }
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "kernel launch for %s failed: %s", l.key.c_str(), hipGetErrorString(e));
    return VTTS_OK;
}


// fp32 fused pair: c1 = convs1_z (rate d), c2 = convs2_z (rate 1): x [B][C][L] -> out, with the accumulate mode of the unfused c2 launch
bool pair_f32_wanted(const vtts_hifigan* h, const Layer& c1, const Layer& c2, int L) {
    if (h->opt_kernels != 0 || h->opt_fuse < 1 || !c1.has_wp || !c2.has_wp) return false;
    if (c1.cin != c1.cout || c2.cin != c1.cin || c2.cout != c1.cin || c2.k != c1.k || c2.dil != 1) return false;
    if (!pair_f32_supported(c1.cin, c1.k, c1.dil, L)) return false;
    // Where the fused pair measured faster than the two launches it replaces (64 x 1024 frames, rocprofv3 per launch, gpurun_out/r04_run3):
    // C = 32: k = 3 / 7 / 11  -16 / -17 / -20 %;  C = 64: -13 / -6 / +-0 %;  C = 128: k = 3 -10 %, k = 7 -1 %, k = 11 +4.5 % (a fused pair
    // recomputes KS - 1 of every 128 columns and holds 78 KB of LDS: at C = 128, k = 11 that costs more than the saved traffic gains).
    // fuse = 1: C <= 64 only; fuse = 2 (default): + C = 128, k = 3; fuse = 3: every pair the kernel covers.
    return c1.cin <= 64 || (h->opt_fuse >= 2 && c1.cin == 128 && c1.k == 3) || h->opt_fuse >= 3;
}

int run_pair_f32(vtts_hifigan* h, const Layer& c1, const Layer& c2, const float* x, int B, int L, float* y, int acc_mode, float div, hipStream_t s) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.x_sb = (long)c1.cin * L;
    a.x_sc = L;
    a.x_st = 1;
    a.wp = h->blob + c1.off_wp;
    a.bias = reinterpret_cast<const float*>(h->blob + c1.off_b);
    a.res = x;  // x = xt + x (model.py:50)
    a.y = y;
    a.B = B;
    a.Cin = c1.cin;
    a.Cout = c1.cout;
    a.K = c1.k;
    a.dil = c1.dil;
    a.pad = c1.pad;
    a.stride = 1;
    a.L = L;
    a.Lout = L;
    a.slope_in = 0.1f;  // LRELU_SLOPE, both activations of the pair (model.py:46,48)
    a.acc_mode = acc_mode;
    a.div = div;
    a.zrev = next_zrev(h);
    set_ragged(h, a, L);
    const bool prof = h->opt_profile && c1.cin == h->prof_C && c1.k == h->prof_K;  // (fuse = 3: the dominant class runs here)
    if (prof) {
        if (h->prof_used == h->prof_events.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            h->prof_events.emplace_back(e0, e1);
        }
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].first, s));
    }
    hipError_t e = launch_pair_f32(a, h->blob + c2.off_wp, reinterpret_cast<const float*>(h->blob + c2.off_b), s);
    if (prof) {
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].second, s));
        h->prof_used++;
        h->prof_flops += 2.0 * 2.0 * (double)B * L * c1.cin * c1.cout * c1.k;  // two convolutions
    }
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "fused fp32 pair launch for %s failed: %s", c1.key.c_str(), hipGetErrorString(e));
    return VTTS_OK;
}

// VTTS_BF16X3: the same pair on the bf16 matrix pipe with split operands (kernels_x3.hip); same buffers, same accumulate modes
bool pair_x3_wanted(const vtts_hifigan* h, const Layer& c1, const Layer& c2, int L) {
    return h->x3 && h->opt_kernels == 0 && h->opt_fuse >= 1 && c1.has_x3 && c2.has_x3 && c2.k == c1.k && c2.dil == 1 && c2.cin == c1.cin &&
           pair_x3_supported(c1.cin, c1.k, c1.dil, L);
}

int run_pair_x3(vtts_hifigan* h, const Layer& c1, const Layer& c2, const float* x, int B, int L, float* y, int acc_mode, float div, hipStream_t s) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.x_sb = (long)c1.cin * L;
    a.x_sc = L;
    a.x_st = 1;
    a.bias = reinterpret_cast<const float*>(h->blob + c1.off_b);
    a.res = x;
    a.y = y;
    a.B = B;
    a.Cin = c1.cin;
    a.Cout = c1.cout;
    a.K = c1.k;
    a.dil = c1.dil;
    a.pad = c1.pad;
    a.stride = 1;
    a.L = L;
    a.Lout = L;
    a.slope_in = 0.1f;
    a.acc_mode = acc_mode;
    a.div = div;
    a.zrev = next_zrev(h);
    set_ragged(h, a, L);
    const bool prof = h->opt_profile && c1.cin == h->prof_C && c1.k == h->prof_K;
    if (prof) {
        if (h->prof_used == h->prof_events.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            h->prof_events.emplace_back(e0, e1);
        }
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].first, s));
    }
    hipError_t e = launch_pair_x3(a, h->blob + c1.off_x3, h->blob + c2.off_x3, reinterpret_cast<const float*>(h->blob + c2.off_b), s);
    if (prof) {
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].second, s));
        h->prof_used++;
        h->prof_flops += 2.0 * 2.0 * (double)B * L * c1.cin * c1.cout * c1.k;  // ALGORITHMIC flops of the two convolutions (the kernel issues 3x as many bf16 MFMA flops)
    }
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "split-operand pair launch for %s failed: %s", c1.key.c_str(), hipGetErrorString(e));
    return VTTS_OK;
}

// VTTS_BF16X3: a whole ResBlock1 (three pairs + the MRF bookkeeping) in one launch where the kernel exists and is the faster choice
// (kernels_x3_rb.hip; fuse = 2: C = 32 and C = 64, k = 3; fuse = 3: wherever it exists); bit-identical to the three pair launches
bool resblock_x3_wanted(const vtts_hifigan* h, const Layer* rb, int L) {
    if (!h->x3 || h->opt_kernels != 0 || h->opt_fuse < 2) return false;
    for (int q = 0; q < 6; ++q)
        if (!rb[q].has_x3 || rb[q].cin != rb[0].cin || rb[q].cout != rb[0].cin || rb[q].k != rb[0].k || ((q & 1) && rb[q].dil != 1)) return false;
    const int dils[3] = {rb[0].dil, rb[2].dil, rb[4].dil};
    if (!resblock_x3_supported(rb[0].cin, rb[0].k, dils, L)) return false;
    return h->opt_fuse >= 3 || resblock_x3_preferred(rb[0].cin, rb[0].k);
}

int run_resblock_x3(vtts_hifigan* h, const Layer* rb, const float* x, int B, int L, float* y, int acc_mode, float div, hipStream_t s) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.x_sb = (long)rb[0].cin * L;
    a.x_sc = L;
    a.x_st = 1;
    a.y = y;
    a.B = B;
    a.Cin = rb[0].cin;
    a.Cout = rb[0].cin;
    a.K = rb[0].k;
    a.stride = 1;
    a.L = L;
    a.Lout = L;
    a.slope_in = 0.1f;  // LRELU_SLOPE (model.py:46,48)
    a.acc_mode = acc_mode;
    a.div = div;
    a.zrev = next_zrev(h);
    set_ragged(h, a, L);
    const int dils[3] = {rb[0].dil, rb[2].dil, rb[4].dil};
    const void* w[6];
    const float* bias[6];
    for (int q = 0; q < 6; ++q) {
        w[q] = h->blob + rb[q].off_x3;
        bias[q] = reinterpret_cast<const float*>(h->blob + rb[q].off_b);
    }
    // option "profile": a class the whole-ResBlock kernel serves is bracketed like the pair launches it replaces (six convolutions per launch)
    const bool prof = h->opt_profile && rb[0].cin == h->prof_C && rb[0].k == h->prof_K;
    if (prof) {
        if (h->prof_used == h->prof_events.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            h->prof_events.emplace_back(e0, e1);
        }
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].first, s));
    }
    hipError_t e = launch_resblock_x3(a, dils, w, bias, s);
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "split-operand ResBlock launch for %s failed: %s", rb[0].key.c_str(), hipGetErrorString(e));
    if (prof) {
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].second, s));
        h->prof_used++;
        h->prof_flops += 6 * 2.0 * (double)B * L * rb[0].cin * rb[0].cin * rb[0].k;
    }
    return VTTS_OK;
}

// ---- bf16 path -------------------------------------------------------------------------------------
// ragged batches (vtts_hifigan_forward_ragged): every layer learns each utterance's valid rows = frames * (rows per frame)
void set_ragged(const vtts_hifigan* h, BConvArgs& a, int L) {
    a.lens = h->cur_lens ? h->cur_lens + h->cur_b0 : nullptr;
    a.len_mul = h->cur_lens ? L / h->cur_T : 1;
}

int run_layer_bf16(vtts_hifigan* h, const Layer& l, const void* x, int x_pitch, int cin_real, int B, int L, float slope_in,
                   float slope_out, const void* res, void* y, int acc_add, float div, hipStream_t s) {
    BConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.wp = h->blob + l.off_wb;
    a.bias = reinterpret_cast<const float*>(h->blob + l.off_b);
    a.res = res;
    a.y = y;
    a.B = B;
    a.L = L;
    set_ragged(h, a, L);
    a.x_pitch = x_pitch;
    a.cin_real = cin_real;
    a.dil = (l.kind == KIND_CONVT) ? 1 : l.dil;
    a.pad = (l.kind == KIND_CONVT) ? 1 : l.pad;
    a.convt_halves = (l.kind == KIND_CONVT) ? 1 : 0;
    a.slope_in = slope_in;
    a.slope_out = slope_out;
    a.acc_add = acc_add;
    a.div = div;
    const int K = (l.kind == KIND_CONVT) ? 3 : l.k;
    const bool prof = h->opt_profile && l.kind == KIND_CONV && l.cin == h->prof_C && l.cout == h->prof_C && l.k == h->prof_K;
    if (prof) {
        if (h->prof_used == h->prof_events.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            h->prof_events.emplace_back(e0, e1);
        }
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].first, s));
    }
    // transposed convolutions: the register-streamed kernel (fuse >= 1), else the first-generation 3-tap convolution.  Per launch at B = 64 x
    // T = 1024 (rocprofv3): ups_0 305 vs 422 us, ups_1 551 vs 985; ups_2 529 vs 688 and ups_3 422 vs 453 since round 3, when the two
    // HBM-bound ones (128 -> 2 x 64, 64 -> 2 x 32) got an LDS-staged epilogue that stores whole rows (round 2: 684 / 597 us with 32-byte
    // stores per frame and instruction, which is why they had stayed on the first-generation kernel; gpurun_out/r03_exp16).
#ifndef VTTS_UG_ALL  // A/B switch: 0 = round 2's policy (ups_2 / ups_3 on the first-generation kernel)
#define VTTS_UG_ALL 1
#endif
    const bool ug_pref = VTTS_UG_ALL || l.bcls == BCLS_UP0 || l.bcls == BCLS_UP1 || h->opt_fuse >= 3;
    // conv_pre (80 -> 512, k = 7, fp32 mel in) runs on the same kernel since round 3 (7 taps per chunk, rows converted while staging)
    const bool ug = (l.kind == KIND_CONVT || l.bcls == BCLS_PRE) && l.has_ug && h->opt_fuse >= 1 && ug_pref && res == nullptr && acc_add == 0 && div == 1.0f;
    if (ug) a.wp = h->blob + l.off_ug;
    if (ug) a.zrev = next_zrev(h);
    hipError_t e = ug ? launch_convt_g_bf16(l.bcls, a, s) : launch_conv_bf16(l.bcls, K, a, s);
    if (prof) {
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].second, s));
        h->prof_used++;
        h->prof_flops += 2.0 * (double)B * L * l.cin * l.cout * l.k;
    }
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "bf16 kernel launch for %s failed: %s", l.key.c_str(), hipGetErrorString(e));
    return VTTS_OK;
}

int run_pair_bf16(vtts_hifigan* h, const Layer& c1, const void* x, int B, int L, float slope_out, void* y, int acc_add, float div,
                  hipStream_t s, float* tail_wav = nullptr) {
    BConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.wp = h->blob + c1.off_pw;
    a.bias = reinterpret_cast<const float*>(h->blob + c1.off_pb);
    a.y = y;
    a.B = B;
    a.L = L;
    set_ragged(h, a, L);
    a.x_pitch = c1.cin;
    a.cin_real = c1.cin;
    a.dil = c1.dil;
    a.pad = c1.pad;
    a.slope_in = 0.1f;
    a.slope_out = slope_out;
    a.acc_add = acc_add;
    a.div = div;
    a.tile_pref = (int)h->opt_tiles;
    a.zrev = next_zrev(h);
    if (tail_wav) {  // the stage-4 tail rides on this launch: conv_post + tanh from the rows this pair produces (kernels_bf16_rbg.hip: GTail)
        const Layer& post = h->layers[h->idx_post];
        a.tail_wav = tail_wav;
        a.tail_wf = reinterpret_cast<const float*>(h->blob + post.off_w);
        a.tail_bias = reinterpret_cast<const float*>(h->blob + post.off_b);
    }
    const bool prof = h->opt_profile && c1.cin == h->prof_C && c1.k == h->prof_K;
    if (prof) {
        if (h->prof_used == h->prof_events.size()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            h->prof_events.emplace_back(e0, e1);
        }
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].first, s));
    }
    hipError_t e = launch_pair_bf16(c1.cin, c1.k, a, s);
    if (prof) {
        HIP_TRY(hipEventRecord(h->prof_events[h->prof_used].second, s));
        h->prof_used++;
        h->prof_flops += 2.0 * 2.0 * (double)B * L * c1.cin * c1.cout * c1.k;  // two convolutions
    }
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "fused pair launch for %s failed: %s", c1.key.c_str(), hipGetErrorString(e));
    return VTTS_OK;
}

int run_resblock_bf16(vtts_hifigan* h, const Layer* rb, const void* x, int B, int L, float slope_out, void* y, int acc_add, float div,
                        hipStream_t s) {
    BConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x;
    a.wp = h->blob + rb[0].off_rw;
    a.bias = reinterpret_cast<const float*>(h->blob + rb[0].off_rb);
    a.y = y;
    a.B = B;
    a.L = L;
    set_ragged(h, a, L);
    a.x_pitch = rb[0].cin;
    a.cin_real = rb[0].cin;
    a.dils[0] = rb[0].dil;
    a.dils[1] = rb[2].dil;
    a.dils[2] = rb[4].dil;
    a.slope_in = 0.1f;
    a.slope_out = slope_out;
    a.acc_add = acc_add;
    a.div = div;
    a.zrev = next_zrev(h);
    hipError_t e = launch_resblock_bf16(rb[0].cin, rb[0].k, a, s);
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "fused ResBlock launch for %s failed: %s", rb[0].key.c_str(), hipGetErrorString(e));
    return VTTS_OK;
}

// The generator's last stage in one launch (kernels_bf16_stage.hip): x = the stage input (ups output, raw), scratch = the stage's accumulator tensor
// (holds the MRF sum across the third ResBlock), wav = this micro-batch's samples.  rbs[j] = first layer of ResBlock j.
bool stage_bf16_wanted(const vtts_hifigan* h, int stage) {
    const vtts_hifigan_cfg& c = h->cfg;
    if (h->dtype != VTTS_BF16 || !h->opt_stage || h->opt_fuse < 2 || c.resblock == 2 || c.num_kernels != 3 || stage + 1 != c.num_upsamples) return false;
    int ks[3], dils[3][3];
    for (int j = 0; j < 3; ++j) {
        const Layer* rb = &h->layers[h->idx_res[stage * 3 + j]];
        if (!rb[0].has_rb) return false;
        ks[j] = rb[0].k;
        for (int z = 0; z < 3; ++z) dils[j][z] = rb[2 * z].dil;
    }
    const Layer& post = h->layers[h->idx_post];
    return stage_bf16_supported(h->layers[h->idx_res[stage * 3]].cin, 3, ks, dils, post.cin, post.cout, post.k);
}
int run_stage_bf16(vtts_hifigan* h, int stage, const void* x, int B, int L, void* scratch, float* wav, float slope_out, hipStream_t s) {
    BStageArgs a;
    memset(&a, 0, sizeof(a));
    int ks[3];
    for (int j = 0; j < 3; ++j) {
        const Layer* rb = &h->layers[h->idx_res[stage * 3 + j]];
        a.wp[j] = h->blob + rb[0].off_rw;
        a.bias[j] = reinterpret_cast<const float*>(h->blob + rb[0].off_rb);
        ks[j] = rb[0].k;
        for (int z = 0; z < 3; ++z) a.dils[j][z] = rb[2 * z].dil;
    }
    const Layer& post = h->layers[h->idx_post];
    a.x = x;
    a.s = scratch;
    a.wav = wav;
    a.post_w = reinterpret_cast<const float*>(h->blob + post.off_w);
    a.post_b = reinterpret_cast<const float*>(h->blob + post.off_b);
    a.B = B;
    a.L = L;
    a.lens = h->cur_lens ? h->cur_lens + h->cur_b0 : nullptr;
    a.len_mul = h->cur_lens ? L / h->cur_T : 1;
    a.zrev = next_zrev(h);
    a.margin = stage_margin(ks, a.dils);
    a.div = 3.0f;
    a.slope_out = slope_out;
    hipError_t e = launch_stage_bf16(a, s);
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "whole-stage launch failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}

struct Taps;
int tap_copy_bf16(const void* src, float* dst, size_t n, hipStream_t s) {
    hipError_t e = launch_bf16_to_f32(src, dst, n, s);
    if (e != hipSuccess) return fail(VTTS_ERR_HIP, "tap conversion failed: %s", hipGetErrorString(e));
    return VTTS_OK;
}

size_t max_act_elems(const vtts_hifigan* h, int T) {
    // largest [C][L] activation per utterance over all stages (8192*T for V1)
    size_t best = (size_t)h->cfg.upsample_initial_channel * T;
    long L = T;
    for (int i = 0; i < h->cfg.num_upsamples; ++i) {
        L *= h->cfg.upsample_rates[i];
        size_t e = (size_t)(h->cfg.upsample_initial_channel >> (i + 1)) * L;
        if (e > best) best = e;
    }
    return best;
}

// Both engines run a large batch as two half-size passes on two streams: one pass's launch tails and memory phases lie under the other's
// MFMAs (fp32: 64 x 1024 frames 366.5 -> 363.9 ms, gpurun_out/r03_exp35; bf16 with the zigzag batch order: 39.95 -> 39.32-39.50 ms,
// profiles/r03_c_narrow_stage_findings.md §4, r03_exp42).  Micro-batches are independent, so the samples do not depend on the schedule
// (bit-identical: tests/test_gpu_bf16.py::test_ragged_batch_across_micro_batches_and_streams, test_gpu_parity.py).  A kernel's duration is its
// own only while nothing else shares the GPU: bench.py therefore times the dominant kernel in a one-stream calibration pass of the same build
// and inputs (option streams = 1), and tools/profile_final.sh profiles that schedule.
int auto_streams(const vtts_hifigan* h) { return h->opt_streams > 0 ? (int)h->opt_streams : 2; }

// mel frames per CALL the Python schedulers build their batches for (option "pass_frames": viettts_amd/longform.py, pipeline.py).  The bf16
// engine takes 65536 frames per call and cuts them into its micro-batches itself; the fp32 engine's callers hand it one micro-batch at a time.
int pass_frames(const vtts_hifigan* h) { return (h->dtype == VTTS_F32 && auto_streams(h) >= 2) ? 32768 : 65536; }

// mel frames per micro-batch (one sequence of launches on one stream)
int microbatch_frames(const vtts_hifigan* h) { return auto_streams(h) >= 2 ? 32768 : 65536; }

int pick_microbatch(const vtts_hifigan* h, int B, int T) {
    if (h->opt_microbatch > 0) return (int)std::min<int64_t>(h->opt_microbatch, B);
    // Enough frames per pass that every launch is many rounds of workgroups on the 256 CUs: with few
    // rounds the last, partly filled one costs 10-20 % (measured: bf16 61.6 ms/step at 4096 frames per
    // pass, 49.9 ms at 65536).  fp32 tiles are 2-4x narrower, so fewer frames reach the same round count.
    // (round 3: the fp32 engine too — 16384 frames per pass measured 388.4 ms per 64 x 1024 batch, 65536 frames 380.7 ms; the 8.6 GB of workspace are 3 % of the HBM)
    const int frames = microbatch_frames(h);
    int mb = (frames + T - 1) / T;
    if (mb < 1) mb = 1;
    if (mb > B) mb = B;
    // equal micro-batches, as many as a multiple of the streams (48 utterances of 1024 frames: 24 + 24, not 32 + 16; 256 sentences padded to
    // 281 frames: 4 x 64, not 3 x 86 on two streams): the streams finish together
    int nmb = (B + mb - 1) / mb;
    const int ns = auto_streams(h);
    if (nmb > 1 && ns > 1) nmb = (nmb + ns - 1) / ns * ns;
    if (nmb > B) nmb = B;
    mb = (B + nmb - 1) / nmb;
    return mb;
}

// The kernels may form one utterance's row * channel offsets (elements or bytes) in 32 bits and index the batch with
// blockIdx.z: a longer utterance goes through the chunk scheduler (viettts_amd/longform.py: 13-frame halo), a larger batch in
// several calls.
int check_pass_size(const vtts_hifigan* h, int B, int T) {
    const size_t es = h->dtype == VTTS_BF16 ? 2 : sizeof(float);
    if (max_act_elems(h, T) * es >= ((size_t)1 << 31))
        return fail(VTTS_ERR_INVALID, "T=%d frames is too long for one pass (%zu activation bytes per utterance, limit 2^31): synthesize it in chunks",
                    T, max_act_elems(h, T) * es);
    if (pick_microbatch(h, B, T) > 65535) return fail(VTTS_ERR_INVALID, "at most 65535 utterances per pass (got %d)", B);
    return VTTS_OK;
}

// x = (rb_0(x) + rb_1(x) [+ rb_2(x)]) / num_kernels with the ResBlocks' outputs in separate buffers (model.py:115-121): the same
// additions in the same order as the ACC_STORE / ACC_ADD / ACC_MEAN epilogues (device_common.h), hence the same bits.
__global__ __launch_bounds__(256) void mrf_mean_k(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ y2,
                                                  float* __restrict__ out, size_t n, float div) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = y0[i] + y1[i];
        if (y2) v = v + y2[i];
        out[i] = v / div;
    }
}

// A single small micro-batch leaves most of the chip idle (B = 1 x T = 512: 256 workgroups of 4 waves in stage 1), and the MRF's
// ResBlocks of a stage are independent given the stage input: the fp32 engine then runs them on parallel streams into separate
// buffers and combines them with mrf_mean_k (3 forwards in flight measured 2.2x the time of one: 1.36x the throughput).
constexpr int PAR_CHAIN_FRAMES = 2048;
bool chains_parallel(const vtts_hifigan* h, int B, int T) {
    const int nk = h->cfg.num_kernels;
    if (!h->opt_chains || ((long)B * T > PAR_CHAIN_FRAMES && h->opt_chains != 2) || pick_microbatch(h, B, T) < B) return false;
    return h->dtype == VTTS_F32 ? (nk == 2 || nk == 3) : (nk == 3 && h->cfg.resblock != 2);
}
// workspace buffers of one pass: [X | S | per ResBlock: T, C (, Y: the fp32 engine's separate output)] or the sequential schedule's X, T, C, S
int pass_buffers(const vtts_hifigan* h, int B, int T) {
    if (!chains_parallel(h, B, T)) return 4;
    return 2 + (h->dtype == VTTS_F32 ? 3 : 2) * h->cfg.num_kernels;
}

struct Taps {
    const char* name = nullptr;
    float* out = nullptr;
    int Bfull = 0;
};



// Micro-batches are independent, so consecutive ones may run on different HIP streams: workgroups of
// one micro-batch in an HBM phase (tile staging / epilogue) then share the CUs with workgroups of
// another in its MFMA phase instead of every workgroup on the chip hitting HBM at the same time.
int num_streams(const vtts_hifigan* h, int B, int T) {
    const int mb = pick_microbatch(h, B, T);
    const int nmb = (B + mb - 1) / mb;
    int n = auto_streams(h);
    if (n > nmb) n = nmb;
    return n < 1 ? 1 : n;
}

int fork_streams(vtts_hifigan* h, int n, hipStream_t s, hipStream_t* out) {
    out[0] = s;
    if (n <= 1) return VTTS_OK;
    if (!h->ev_fork) {
        HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < 3; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&h->side_streams[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
    }
    HIP_TRY(hipEventRecord(h->ev_fork, s));
    for (int i = 1; i < n; ++i) {
        HIP_TRY(hipStreamWaitEvent(h->side_streams[i - 1], h->ev_fork, 0));
        out[i] = h->side_streams[i - 1];
    }
    return VTTS_OK;
}

int join_streams(vtts_hifigan* h, int n, hipStream_t s) {
    for (int i = 1; i < n; ++i) {
        HIP_TRY(hipEventRecord(h->ev_join[i - 1], h->side_streams[i - 1]));
        HIP_TRY(hipStreamWaitEvent(s, h->ev_join[i - 1], 0));
    }
    return VTTS_OK;
}

// bf16 schedule.  Same dataflow as the fp32 one, with two differences that only bf16 needs:
//  * activations are channels-last bf16 [B][L][C] (a transposed convolution's [L][s*Cout] output IS the
//    [s*L][Cout] tensor);
//  * a tensor that is only ever consumed through LeakyReLU is stored already activated by its producer
//    (in fp32, before the bf16 rounding): conv_pre -> ups_0, xt = c1(.) -> c2, MRF mean -> next ups / conv_post.
//    Only the ResBlock's running x is stored raw (it is also the residual) and activated on load by c1.
int forward_bf16(vtts_hifigan* h, const float* mel, int B, int T, float* wav, void* ws, hipStream_t s0, Taps tap) {
    const vtts_hifigan_cfg& c = h->cfg;
    const int mb = pick_microbatch(h, B, T);
    const size_t per = align_up(max_act_elems(h, T) * (size_t)mb * 2, 256);
    const int nstr = num_streams(h, B, T);
    hipStream_t streams[4];
    int rc0 = fork_streams(h, nstr, s0, streams);
    if (rc0) return rc0;
    const int nk = c.num_kernels;
    const long wav_len = (long)h->hop * T;
    const bool par = chains_parallel(h, B, T);
    for (int b0 = 0; b0 < B; b0 += mb) {
        const int si = (b0 / mb) % nstr;
        hipStream_t s = streams[si];
        char* wsb = static_cast<char*>(ws) + (size_t)si * 4 * per;
        char* bufX = wsb + 0 * per;
        char* bufT = wsb + 1 * per;
        char* bufC = wsb + 2 * per;
        char* bufS = wsb + 3 * per;
        if (par) bufS = wsb + 1 * per;  // parallel ResBlocks: [X | S | (T, C) per ResBlock] (one micro-batch, si == 0)
        const int nb = std::min(mb, B - b0);
        h->cur_b0 = b0;
        int rc;
        {
            const Layer& l = h->layers[h->idx_pre];
            rc = run_layer_bf16(h, l, mel + (size_t)b0 * T * c.num_mels, c.num_mels, c.num_mels, nb, T, 1.0f, 0.1f, nullptr, bufS, 0, 1.f, s);
            if (rc) return rc;
            if (tap.name && !strcmp(tap.name, "conv_pre")) {
                rc = tap_copy_bf16(bufS, tap.out + (size_t)b0 * l.cout * T, (size_t)nb * l.cout * T, s);
                if (rc) return rc;
            }
        }
        long L = T;
        bool tail_done = false;  // conv_post + tanh already ran inside the last pair launch
        for (int i = 0; i < c.num_upsamples; ++i) {
            const Layer& up = h->layers[h->idx_ups[i]];
            rc = run_layer_bf16(h, up, bufS, up.cin, up.cin, nb, (int)L, 1.0f, 1.0f, nullptr, bufX, 0, 1.f, s);
            if (rc) return rc;
            L *= up.stride;
            const int C = up.cout;
            const size_t CL = (size_t)C * L;
            if (tap.name && !strncmp(tap.name, "ups_", 4) && atoi(tap.name + 4) == i) {
                rc = tap_copy_bf16(bufX, tap.out + (size_t)b0 * CL, (size_t)nb * CL, s);
                if (rc) return rc;
            }
            const float next_slope = (i + 1 < c.num_upsamples) ? 0.1f : 0.01f;  // model.py:112 / :122
            // the whole last stage in one launch (option "stage", default OFF: it measured slower, profiles/r06_a_stage_kernel_findings.md): ups output -> three
            // ResBlocks -> mean -> LeakyReLU -> conv_post -> tanh from one LDS-resident window per workgroup; nothing but the samples (and the MRF sum once)
            // leaves the CU.  Not when a tap needs a tensor in between.
            if (!tap.name && stage_bf16_wanted(h, i)) {
                rc = run_stage_bf16(h, i, bufX, nb, (int)L, bufS, wav + (size_t)b0 * wav_len, next_slope, s);
                if (rc) return rc;
                tail_done = true;
                continue;
            }
            // the stage-4 tail (option "tail", default on): the generator's last pair launch also runs conv_post + tanh on its own rows, so the stage
            // output is never written and conv_post_bf16_k is not launched.  Where the last ResBlock ends in a pair launch that can carry it (V1: C = 32,
            // k = 11) and nothing needs the stage output itself (no tap)
            float* tail_dst = nullptr;
            if (h->opt_tail && !tap.name && i + 1 == c.num_upsamples && c.resblock != 2 && h->opt_fuse >= 1) {
                const int lb = h->idx_res[i * nk + nk - 1];
                const Layer& lc = h->layers[lb];
                const Layer& post = h->layers[h->idx_post];
                const bool whole_rb = h->opt_fuse >= 2 && lc.has_rb && (h->opt_fuse >= 3 || resblock_bf16_preferred(lc.cin, lc.k));
                if (!whole_rb && lc.has_pair && h->layers[lb + 2].has_pair && h->layers[lb + 4].has_pair &&
                    pair_tail_bf16_supported(lc.cin, lc.k, post.cin, post.cout, post.k))
                    tail_dst = wav + (size_t)b0 * wav_len;
            }
            tail_done = tail_dst != nullptr;
            // one ResBlock of the MRF: X -> (tT, tC scratch) -> the shared accumulator S (store / accumulate / accumulate-and-divide in the
            // LAST kernel's epilogue); `before_last` runs right before that kernel is enqueued (the parallel schedule's ordering point)
            auto run_chain = [&](int j, char* tT, char* tC, hipStream_t cs, auto before_last) -> int {
                const int base = h->idx_res[i * nk + j];
                const char* cur = bufX;
                const bool last_rb = (j == nk - 1);
                int rcc;
                if (c.resblock == 2) {
                    // ResBlock2: x = c_z(leaky_relu(x, 0.1)) + x, z = 0, 1 (model.py:69-74): the running x is stored raw (it is the residual) and
                    // activated on load; the MRF sum / mean and the consumer's LeakyReLU in the second convolution's epilogue.  X -> C -> S
                    if ((rcc = run_layer_bf16(h, h->layers[base], cur, C, C, nb, (int)L, 0.1f, 1.0f, cur, tC, 0, 1.f, cs))) return rcc;
                    if ((rcc = before_last())) return rcc;
                    return run_layer_bf16(h, h->layers[base + 1], tC, C, C, nb, (int)L, 0.1f, last_rb ? next_slope : 1.0f, tC, bufS, j > 0 ? 1 : 0,
                                          last_rb ? (float)nk : 1.0f, cs);
                }
                // the whole-ResBlock kernel where it exists and is the faster choice (fuse = 3: wherever it exists)
                if (h->opt_fuse >= 2 && h->layers[base].has_rb && (h->opt_fuse >= 3 || resblock_bf16_preferred(h->layers[base].cin, h->layers[base].k))) {
                    // the whole ResBlock in one kernel: X -> S (store / accumulate / accumulate-and-divide)
                    if ((rcc = before_last())) return rcc;
                    return run_resblock_bf16(h, &h->layers[base], cur, nb, (int)L, last_rb ? next_slope : 1.0f, bufS, j > 0 ? 1 : 0,
                                             last_rb ? (float)nk : 1.0f, cs);
                }
                if (h->opt_fuse && h->layers[base].has_pair && h->layers[base + 2].has_pair && h->layers[base + 4].has_pair) {
                    // fused pairs cannot run in place (a neighbour tile's halo would see updated rows):
                    // X -> T -> C -> S, with X kept for the other ResBlocks of the stage
                    if ((rcc = run_pair_bf16(h, h->layers[base + 0], cur, nb, (int)L, 1.0f, tT, 0, 1.f, cs))) return rcc;
                    if ((rcc = run_pair_bf16(h, h->layers[base + 2], tT, nb, (int)L, 1.0f, tC, 0, 1.f, cs))) return rcc;
                    if ((rcc = before_last())) return rcc;
                    return run_pair_bf16(h, h->layers[base + 4], tC, nb, (int)L, last_rb ? next_slope : 1.0f, bufS, j > 0 ? 1 : 0,
                                         last_rb ? (float)nk : 1.0f, cs, last_rb ? tail_dst : nullptr);
                }
                for (int z = 0; z < 3; ++z) {
                    const Layer& c1 = h->layers[base + 2 * z];
                    const Layer& c2 = h->layers[base + 2 * z + 1];
                    if ((rcc = run_layer_bf16(h, c1, cur, C, C, nb, (int)L, 0.1f, 0.1f, nullptr, tT, 0, 1.f, cs))) return rcc;
                    if (z < 2) {
                        if ((rcc = run_layer_bf16(h, c2, tT, C, C, nb, (int)L, 1.0f, 1.0f, cur, tC, 0, 1.f, cs))) return rcc;
                        cur = tC;
                    } else {
                        if ((rcc = before_last())) return rcc;
                        if ((rcc = run_layer_bf16(h, c2, tT, C, C, nb, (int)L, 1.0f, last_rb ? next_slope : 1.0f, cur, bufS, j > 0 ? 1 : 0,
                                                  last_rb ? (float)nk : 1.0f, cs))) return rcc;
                    }
                }
                return VTTS_OK;
            };
            if (par) {
                // The ResBlocks side by side on nk streams with scratch of their own.  Only a ResBlock's LAST kernel touches the shared
                // accumulator (bf16, rounded after every addition): those kernels are chained by events in the sequential order
                // rb_0 -> rb_1 -> rb_2, so every sample sees the same additions and roundings as one-after-the-other — the same bits.
                hipStream_t cs[4];
                if ((rc = fork_streams(h, nk, s, cs))) {
                    (void)join_streams(h, nk, s);
                    return rc;
                }
                // every failure inside this block is collected in rc (no early return): the forked chain streams MUST be re-joined into
                // `s` below, also when a launch fails — during graph capture an un-joined fork invalidates the capture
                auto hip_rc = [&](hipError_t e, const char* what) -> int {
                    return e == hipSuccess ? VTTS_OK : fail(VTTS_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
                };
                for (int j = 0; j < nk && !rc; ++j) {
                    char* tT = wsb + (size_t)(2 + 2 * j) * per;
                    char* tC = wsb + (size_t)(3 + 2 * j) * per;
                    rc = run_chain(j, tT, tC, cs[j], [&]() -> int {
                        return j > 0 ? hip_rc(hipStreamWaitEvent(cs[j], h->ev_chain[j - 1], 0), "hipStreamWaitEvent(chain)") : VTTS_OK;
                    });
                    if (!rc && j + 1 < nk) {
                        if (!h->ev_chain[j]) rc = hip_rc(hipEventCreateWithFlags(&h->ev_chain[j], hipEventDisableTiming), "hipEventCreateWithFlags(chain)");
                        if (!rc) rc = hip_rc(hipEventRecord(h->ev_chain[j], cs[j]), "hipEventRecord(chain)");
                    }
                }
                const int rj = join_streams(h, nk, s);
                if (rc) return rc;
                if (rj) return rj;
            } else {
                for (int j = 0; j < nk; ++j)
                    if ((rc = run_chain(j, bufT, bufC, s, []() -> int { return VTTS_OK; }))) return rc;
            }
            if (tap.name && !strncmp(tap.name, "mrf_", 4) && atoi(tap.name + 4) == i) {
                rc = tap_copy_bf16(bufS, tap.out + (size_t)b0 * CL, (size_t)nb * CL, s);
                if (rc) return rc;
            }
        }
        if (!tail_done) {
            const Layer& l = h->layers[h->idx_post];
            BConvArgs a;
            memset(&a, 0, sizeof(a));
            a.x = bufS;
            a.wf = reinterpret_cast<const float*>(h->blob + l.off_w);
            a.bias = reinterpret_cast<const float*>(h->blob + l.off_b);
            a.B = nb;
            a.L = (int)L;
            set_ragged(h, a, (int)L);
            float* pre = (tap.name && !strcmp(tap.name, "pre_tanh")) ? tap.out + (size_t)b0 * wav_len : nullptr;
            hipError_t e = launch_conv_post_bf16(a, wav + (size_t)b0 * wav_len, pre, s);
            if (e != hipSuccess) return fail(VTTS_ERR_HIP, "conv_post launch failed: %s", hipGetErrorString(e));
        }
    }
    return join_streams(h, nstr, s0);
}

int forward_impl(vtts_hifigan* h, const float* mel, int B, int T, float* wav, void* ws, size_t ws_bytes, hipStream_t s,
                 Taps tap) {
    if (!h->blob) return fail(VTTS_ERR_STATE, "forward() before pack()/bind_packed()");
    if (B <= 0 || T <= 0) return fail(VTTS_ERR_INVALID, "B and T must be positive (got B=%d, T=%d)", B, T);
    if (int rc = check_pass_size(h, B, T)) return rc;
    size_t need = 0;
    vtts_hifigan_workspace_bytes(h, B, T, &need);
    if (ws_bytes < need || !ws) return fail(VTTS_ERR_NOMEM, "workspace too small: %zu < %zu bytes", ws_bytes, need);
    if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) return fail(VTTS_ERR_INVALID, "workspace must be 256-B aligned");

    {
        hipError_t e = hipSetDevice(h->device);  // side streams / events are created lazily: on THIS handle's device
        if (e != hipSuccess) return fail(VTTS_ERR_HIP, "hipSetDevice(%d) failed: %s", h->device, hipGetErrorString(e));
    }
    if (h->dtype == VTTS_BF16) {
        const int rc = forward_bf16(h, mel, B, T, wav, ws, s, tap);
        if (rc) (void)join_streams(h, num_streams(h, B, T), s);  // a failed launch must not leave forked side streams un-joined
        return rc;
    }

    const vtts_hifigan_cfg& c = h->cfg;
    const int mb = pick_microbatch(h, B, T);
    const size_t per = align_up(max_act_elems(h, T) * (size_t)mb * sizeof(float), 256);
    const int nstr = num_streams(h, B, T);
    hipStream_t streams[4];
    hipStream_t s0 = s;
    int rc0 = fork_streams(h, nstr, s0, streams);
    if (rc0) return rc0;
    const int nk = c.num_kernels;
    const long wav_len = (long)h->hop * T;
    const bool par = chains_parallel(h, B, T);

    for (int b0 = 0; b0 < B; b0 += mb) {
        const int si = (b0 / mb) % nstr;
        hipStream_t s = streams[si];
        char* wsb = static_cast<char*>(ws) + (size_t)si * 4 * per;
        float* bufX = reinterpret_cast<float*>(wsb + 0 * per);   // stage input x (ups output)
        float* bufT = reinterpret_cast<float*>(wsb + 1 * per);   // xt inside a ResBlock pair
        float* bufC = reinterpret_cast<float*>(wsb + 2 * per);   // running x inside a ResBlock
        float* bufS = reinterpret_cast<float*>(wsb + 3 * per);   // MRF accumulator xs / stage output
        if (par) bufS = reinterpret_cast<float*>(wsb + 1 * per);  // parallel ResBlocks: [X | S | (T, C, Y) per ResBlock] (one micro-batch, si == 0)
        const int nb = std::min(mb, B - b0);
        h->cur_b0 = b0;
        int rc;
        // conv_pre (model.py:110): mel [nb][T][num_mels] NWC -> S [nb][C0][T]
        {
            const Layer& l = h->layers[h->idx_pre];
            Act x{mel + (long)b0 * T * c.num_mels, (long)T * c.num_mels, 1, c.num_mels};
            rc = run_layer(h, l, x, nb, T, 1.0f, nullptr, bufS, ACC_STORE, 1.f, 0, nullptr, s);
            if (rc) { (void)join_streams(h, nstr, s0); return rc; }
            if (tap.name && !strcmp(tap.name, "conv_pre"))
                HIP_TRY(hipMemcpyAsync(tap.out + (size_t)b0 * l.cout * T, bufS, (size_t)nb * l.cout * T * sizeof(float),
                                       hipMemcpyDeviceToDevice, s));
        }
        long L = T;
        for (int i = 0; i < c.num_upsamples; ++i) {
            const Layer& up = h->layers[h->idx_ups[i]];
            // x = ups_i(leaky_relu(x, 0.1))   (model.py:112-114)
            Act xin{bufS, (long)up.cin * L, L, 1};
            rc = run_layer(h, up, xin, nb, (int)L, 0.1f, nullptr, bufX, ACC_STORE, 1.f, 0, nullptr, s);
            if (rc) { (void)join_streams(h, nstr, s0); return rc; }
            L *= up.stride;
            const int C = up.cout;
            const long CL = (long)C * L;
            if (tap.name && !strncmp(tap.name, "ups_", 4) && atoi(tap.name + 4) == i)
                HIP_TRY(hipMemcpyAsync(tap.out + (size_t)b0 * CL, bufX, (size_t)nb * CL * sizeof(float), hipMemcpyDeviceToDevice, s));
            // one ResBlock of the MRF: X -> (T, C scratch) -> out with the given accumulate mode, on stream cs
            auto run_chain = [&](int j, float* tT, float* tC, float* out, int mode, float div, hipStream_t cs) -> int {
                const int base = h->idx_res[i * nk + j];
                const float* cur = bufX;
                int rcc = VTTS_OK;
                if (c.resblock == 2) {
                    // ResBlock2: x = c_z(leaky_relu(x, 0.1)) + x, z = 0, 1 (model.py:69-74); the MRF sum / mean in the last epilogue
                    for (int z = 0; z < 2 && !rcc; ++z) {
                        const Layer& cz = h->layers[base + z];
                        if (z == 0) {
                            rcc = run_layer(h, cz, Act{cur, CL, L, 1}, nb, (int)L, 0.1f, cur, tC, ACC_STORE, 1.f, 0, nullptr, cs);
                            cur = tC;
                        } else {
                            rcc = run_layer(h, cz, Act{cur, CL, L, 1}, nb, (int)L, 0.1f, cur, out, mode, div, 0, nullptr, cs);
                        }
                    }
                    return rcc;
                }
                if (resblock_x3_wanted(h, &h->layers[base], (int)L))  // VTTS_BF16X3, narrow stages: the whole ResBlock X -> out in one launch
                    return run_resblock_x3(h, &h->layers[base], cur, nb, (int)L, out, mode, div, cs);
                if (pair_x3_wanted(h, h->layers[base], h->layers[base + 1], (int)L) && pair_x3_wanted(h, h->layers[base + 2], h->layers[base + 3], (int)L) &&
                    pair_x3_wanted(h, h->layers[base + 4], h->layers[base + 5], (int)L)) {
                    // VTTS_BF16X3: X -> T -> C -> out as the fused fp32 pairs below
                    if ((rcc = run_pair_x3(h, h->layers[base + 0], h->layers[base + 1], cur, nb, (int)L, tT, ACC_STORE, 1.f, cs))) return rcc;
                    if ((rcc = run_pair_x3(h, h->layers[base + 2], h->layers[base + 3], tT, nb, (int)L, tC, ACC_STORE, 1.f, cs))) return rcc;
                    return run_pair_x3(h, h->layers[base + 4], h->layers[base + 5], tC, nb, (int)L, out, mode, div, cs);
                }
                if (pair_f32_wanted(h, h->layers[base], h->layers[base + 1], (int)L) && pair_f32_wanted(h, h->layers[base + 2], h->layers[base + 3], (int)L) &&
                    pair_f32_wanted(h, h->layers[base + 4], h->layers[base + 5], (int)L)) {
                    // fused pairs cannot run in place (a neighbour tile's halo would see updated columns): X -> T -> C -> out
                    if ((rcc = run_pair_f32(h, h->layers[base + 0], h->layers[base + 1], cur, nb, (int)L, tT, ACC_STORE, 1.f, cs))) return rcc;
                    if ((rcc = run_pair_f32(h, h->layers[base + 2], h->layers[base + 3], tT, nb, (int)L, tC, ACC_STORE, 1.f, cs))) return rcc;
                    return run_pair_f32(h, h->layers[base + 4], h->layers[base + 5], tC, nb, (int)L, out, mode, div, cs);
                }
                for (int z = 0; z < 3 && !rcc; ++z) {
                    const Layer& c1 = h->layers[base + 2 * z];
                    const Layer& c2 = h->layers[base + 2 * z + 1];
                    // xt = c1(leaky_relu(x, 0.1))                       (model.py:46-47)
                    rcc = run_layer(h, c1, Act{cur, CL, L, 1}, nb, (int)L, 0.1f, nullptr, tT, ACC_STORE, 1.f, 0, nullptr, cs);
                    if (rcc) break;
                    // x = c2(leaky_relu(xt, 0.1)) + x                  (model.py:48-50)
                    if (z < 2) {
                        rcc = run_layer(h, c2, Act{tT, CL, L, 1}, nb, (int)L, 0.1f, cur, tC, ACC_STORE, 1.f, 0, nullptr, cs);
                        cur = tC;
                    } else {
                        // last pair of the ResBlock: its output (sequential schedule: the MRF sum / mean folded into the epilogue, model.py:115-121)
                        rcc = run_layer(h, c2, Act{tT, CL, L, 1}, nb, (int)L, 0.1f, cur, out, mode, div, 0, nullptr, cs);
                    }
                }
                return rcc;
            };
            if (par) {
                // the ResBlocks side by side on nk streams, each into its own buffers; ((y0 + y1) + y2) / nk afterwards: same additions, same order
                hipStream_t cs[4];
                rc = fork_streams(h, nk, s, cs);
                if (rc) { (void)join_streams(h, nstr, s0); return rc; }
                float* ys[3] = {nullptr, nullptr, nullptr};
                for (int j = 0; j < nk; ++j) {
                    float* cb = reinterpret_cast<float*>(wsb + (size_t)(2 + 3 * j) * per);
                    float* tT = cb;
                    float* tC = reinterpret_cast<float*>(wsb + (size_t)(3 + 3 * j) * per);
                    ys[j] = reinterpret_cast<float*>(wsb + (size_t)(4 + 3 * j) * per);
                    rc = run_chain(j, tT, tC, ys[j], ACC_STORE, 1.f, cs[j]);
                    if (rc) { (void)join_streams(h, nk, s); (void)join_streams(h, nstr, s0); return rc; }
                }
                rc = join_streams(h, nk, s);
                if (rc) { (void)join_streams(h, nstr, s0); return rc; }
                const size_t n = (size_t)nb * CL;
                hipLaunchKernelGGL(mrf_mean_k, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, ys[0], ys[1], ys[2], bufS, n, (float)nk);
                if (hipGetLastError() != hipSuccess) { (void)join_streams(h, nstr, s0); return fail(VTTS_ERR_HIP, "mrf_mean launch failed"); }
            } else {
                for (int j = 0; j < nk; ++j) {
                    const int mode = (j == 0) ? ACC_STORE : (j == nk - 1 ? ACC_MEAN : ACC_ADD);
                    rc = run_chain(j, bufT, bufC, bufS, mode, (float)nk, s);
                    if (rc) { (void)join_streams(h, nstr, s0); return rc; }
                }
            }
            if (nk == 1) {
                // a single-kernel MRF still divides by num_kernels == 1: identity, nothing to do
            }
            if (tap.name && !strncmp(tap.name, "mrf_", 4) && atoi(tap.name + 4) == i)
                HIP_TRY(hipMemcpyAsync(tap.out + (size_t)b0 * CL, bufS, (size_t)nb * CL * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        // tail: tanh(conv_post(leaky_relu(x)))  — slope 0.01, the jax default (model.py:122-124)
        {
            const Layer& l = h->layers[h->idx_post];
            float* pre = (tap.name && !strcmp(tap.name, "pre_tanh")) ? tap.out + (size_t)b0 * wav_len : nullptr;
            rc = run_layer(h, l, Act{bufS, (long)l.cin * L, L, 1}, nb, (int)L, 0.01f, nullptr, wav + (size_t)b0 * wav_len,
                           ACC_STORE, 1.f, 1, pre, s);
            if (rc) { (void)join_streams(h, nstr, s0); return rc; }
        }
    }
    return join_streams(h, nstr, s0);
}

}  // namespace

// ================================ C ABI ==========================================================
VTTS_API int vtts_abi_version(void) { return VTTS_ABI_VERSION; }

VTTS_API const char* vtts_last_error(void) { return g_last_error.c_str(); }

VTTS_API int vtts_hifigan_create(const vtts_hifigan_cfg* cfg, int device, int dtype, vtts_hifigan** out) {
    if (!cfg || !out) return fail(VTTS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (dtype != VTTS_F32 && dtype != VTTS_BF16 && dtype != VTTS_BF16X3) return fail(VTTS_ERR_INVALID, "unknown dtype %d", dtype);
    if (cfg->num_upsamples < 1 || cfg->num_upsamples > VTTS_MAX_UPSAMPLES)
        return fail(VTTS_ERR_INVALID, "num_upsamples %d out of range", cfg->num_upsamples);
    if (cfg->num_kernels < 1 || cfg->num_kernels > VTTS_MAX_KERNELS)
        return fail(VTTS_ERR_INVALID, "num_kernels %d out of range", cfg->num_kernels);
    if (cfg->num_mels < 1 || cfg->upsample_initial_channel < 1)
        return fail(VTTS_ERR_INVALID, "num_mels / upsample_initial_channel must be positive");
    if (cfg->upsample_initial_channel % (1 << cfg->num_upsamples) != 0)
        return fail(VTTS_ERR_INVALID, "upsample_initial_channel %d not divisible by 2^%d", cfg->upsample_initial_channel,
                    cfg->num_upsamples);
    for (int i = 0; i < cfg->num_upsamples; ++i)
        if (cfg->upsample_rates[i] < 1 || cfg->upsample_kernel_sizes[i] < cfg->upsample_rates[i])
            return fail(VTTS_ERR_INVALID, "bad upsample stage %d (rate %d, kernel %d)", i, cfg->upsample_rates[i],
                        cfg->upsample_kernel_sizes[i]);
    for (int j = 0; j < cfg->num_kernels; ++j) {
        if (cfg->resblock_kernel_sizes[j] < 1 || cfg->resblock_kernel_sizes[j] % 2 == 0)
            return fail(VTTS_ERR_INVALID, "resblock kernel size %d must be odd", cfg->resblock_kernel_sizes[j]);
        for (int z = 0; z < (cfg->resblock == 2 ? 2 : 3); ++z)
            if (cfg->resblock_dilation_sizes[j][z] < 1) return fail(VTTS_ERR_INVALID, "dilation must be >= 1");
    }
    if (cfg->resblock != 0 && cfg->resblock != 1 && cfg->resblock != 2)
        return fail(VTTS_ERR_INVALID, "resblock must be 1 (ResBlock1) or 2 (ResBlock2), got %d", cfg->resblock);
    if (device < 0) return fail(VTTS_ERR_INVALID, "device %d out of range", device);
    // the device itself is first touched in pack()/bind_packed(): planning needs no GPU
    auto* h = new (std::nothrow) vtts_hifigan();
    if (!h) return fail(VTTS_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    if (h->cfg.resblock == 0) h->cfg.resblock = 1;
    h->device = device;
    h->dtype = dtype == VTTS_BF16X3 ? VTTS_F32 : dtype;
    h->x3 = dtype == VTTS_BF16X3;
    const int rc = build_layers(h);
    if (rc != VTTS_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return VTTS_OK;
}

static void drop_graphs(vtts_hifigan* h) {
    for (auto& g : h->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    h->cap_stream = nullptr;
}

VTTS_API void vtts_hifigan_destroy(vtts_hifigan* h) {
    if (!h) return;
    for (int i = 0; i < 3; ++i) {
        if (h->side_streams[i]) (void)hipStreamDestroy(h->side_streams[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_chain)
        if (e) (void)hipEventDestroy(e);
    drop_graphs(h);
    for (auto& p : h->prof_events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    delete h;
}

VTTS_API int vtts_hifigan_num_params(const vtts_hifigan* h, int* n) {
    if (!h || !n) return fail(VTTS_ERR_INVALID, "null argument");
    *n = 2 * (int)h->layers.size();
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_param_info(const vtts_hifigan* h, int i, const char** key, const char** which, int64_t shape[3],
                                     int* ndim) {
    if (!h || !key || !which || !shape || !ndim) return fail(VTTS_ERR_INVALID, "null argument");
    if (i < 0 || i >= 2 * (int)h->layers.size()) return fail(VTTS_ERR_INVALID, "parameter index %d out of range", i);
    const Layer& l = h->layers[i / 2];
    *key = l.key.c_str();
    if (i % 2 == 0) {
        *which = "w";
        *ndim = 3;
        shape[0] = l.k;
        shape[1] = (l.kind == KIND_CONV) ? l.cin : l.cout;
        shape[2] = (l.kind == KIND_CONV) ? l.cout : l.cin;
    } else {
        *which = "b";
        *ndim = 1;
        shape[0] = l.cout;
        shape[1] = shape[2] = 0;
    }
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_set_param(vtts_hifigan* h, const char* key, const char* which, const float* host,
                                    const int64_t* shape, int ndim) {
    if (!h || !key || !which || !host || !shape) return fail(VTTS_ERR_INVALID, "null argument");
    Layer* l = find_layer(h, key);
    if (!l) return fail(VTTS_ERR_INVALID, "unknown parameter module '%s'", key);
    if (!strcmp(which, "w")) {
        const int64_t d1 = (l->kind == KIND_CONV) ? l->cin : l->cout;
        const int64_t d2 = (l->kind == KIND_CONV) ? l->cout : l->cin;
        if (ndim != 3 || shape[0] != l->k || shape[1] != d1 || shape[2] != d2)
            return fail(VTTS_ERR_SHAPE, "%s/w: expected [%d,%lld,%lld]", key, l->k, (long long)d1, (long long)d2);
        l->w.assign(host, host + (size_t)l->k * l->cin * l->cout);
        l->have_w = true;
    } else if (!strcmp(which, "b")) {
        if (ndim != 1 || shape[0] != l->cout) return fail(VTTS_ERR_SHAPE, "%s/b: expected [%d]", key, l->cout);
        l->b.assign(host, host + l->cout);
        l->have_b = true;
    } else {
        return fail(VTTS_ERR_INVALID, "parameter name must be \"w\" or \"b\", got \"%s\"", which);
    }
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_packed_bytes(const vtts_hifigan* h, size_t* bytes) {
    if (!h || !bytes) return fail(VTTS_ERR_INVALID, "null argument");
    *bytes = h->blob_bytes;
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_pack(vtts_hifigan* h, void* dev_blob, size_t blob_bytes, vtts_stream stream) {
    if (!h || !dev_blob) return fail(VTTS_ERR_INVALID, "null argument");
    if (blob_bytes < h->blob_bytes) return fail(VTTS_ERR_NOMEM, "blob too small: %zu < %zu", blob_bytes, h->blob_bytes);
    if ((reinterpret_cast<uintptr_t>(dev_blob) & 255) != 0) return fail(VTTS_ERR_INVALID, "blob must be 256-B aligned");
    for (auto& l : h->layers)
        if (!l.have_w || !l.have_b) return fail(VTTS_ERR_MISSING, "parameter %s/%s was never set", l.key.c_str(), l.have_w ? "b" : "w");
    std::vector<char> host(h->blob_bytes, 0);
    if (h->dtype == VTTS_BF16) {
        for (auto& l : h->layers) {
            float* bb = reinterpret_cast<float*>(host.data() + l.off_b);
            for (int i = 0; i < l.coutp; ++i) bb[i] = l.b[i % l.cout];  // transposed conv rows are (phase, co)
            if (l.bcls < BCLS_NONE) {  // conv_post keeps plain fp32 weights
                memcpy(host.data() + l.off_w, l.w.data(), l.w.size() * sizeof(float));
                continue;
            }
            const BPackGeom g = bf16_pack_geom(l.bcls, l.kind == KIND_CONVT ? 3 : l.k);
            if (l.kind == KIND_CONVT) {
                // ConvTranspose1d(k = 2s) == Conv1d(Cin -> s*Cout, k = 3, pad 1) on channels-last data:
                // phase r (group g = r / (s/2)) reads frames q + g - 1 + m (m = 0,1) through tap
                // j = s*(g - 1 + m) + pad_a - r; the third frame of each phase gets zero weights.
                const int sdt = l.stride, coutp = l.coutp;
                std::vector<float> wc((size_t)3 * l.cin * coutp, 0.f);
                for (int r = 0; r < sdt; ++r) {
                    const int gq = r / (sdt / 2);
                    for (int m = 0; m < 2; ++m) {
                        const int f = gq + m, j = sdt * (gq - 1 + m) + l.pad_a - r;
                        for (int co = 0; co < l.cout; ++co)
                            for (int ci = 0; ci < l.cin; ++ci)
                                wc[((size_t)f * l.cin + ci) * coutp + r * l.cout + co] = l.w[((size_t)j * l.cout + co) * l.cin + ci];
                    }
                }
                bf16_pack(wc.data(), l.cin, g, reinterpret_cast<unsigned short*>(host.data() + l.off_wb));
                if (l.has_ug) bf16_pack(wc.data(), l.cin, convt_g_pack_geom(l.bcls), reinterpret_cast<unsigned short*>(host.data() + l.off_ug));
            } else {
                bf16_pack(l.w.data(), l.cin, g, reinterpret_cast<unsigned short*>(host.data() + l.off_wb));
                if (l.has_ug) bf16_pack(l.w.data(), l.cin, convt_g_pack_geom(l.bcls), reinterpret_cast<unsigned short*>(host.data() + l.off_ug));  // conv_pre
            }
        }
        for (size_t i = 0; i + 1 < h->layers.size(); ++i) {
            const Layer& c1 = h->layers[i];
            if (!c1.has_pair) continue;
            const Layer& c2 = h->layers[i + 1];
            const BPackGeom pg = pair_pack_geom(c1.cin, c1.k);
            const size_t half = bf16_packed_bytes(pg);
            bf16_pack(c1.w.data(), c1.cin, pg, reinterpret_cast<unsigned short*>(host.data() + c1.off_pw));
            bf16_pack(c2.w.data(), c2.cin, pg, reinterpret_cast<unsigned short*>(host.data() + c1.off_pw + half));
            float* pb = reinterpret_cast<float*>(host.data() + c1.off_pb);
            memcpy(pb, c1.b.data(), c1.cin * sizeof(float));
            memcpy(pb + c1.cin, c2.b.data(), c1.cin * sizeof(float));
        }
        for (size_t i = 0; i + 5 < h->layers.size(); ++i) {
            const Layer& c0 = h->layers[i];
            if (!c0.has_rb) continue;
            const BPackGeom pg = pair_g_pack_geom(c0.cin, c0.k);
            const size_t one = bf16_packed_bytes(pg);
            float* rb = reinterpret_cast<float*>(host.data() + c0.off_rb);
            for (int q = 0; q < 6; ++q) {
                const Layer& l = h->layers[i + q];
                bf16_pack(l.w.data(), l.cin, pg, reinterpret_cast<unsigned short*>(host.data() + c0.off_rw + q * one));
                memcpy(rb + (size_t)q * c0.cin, l.b.data(), c0.cin * sizeof(float));
            }
        }
    }
    for (auto& l : h->layers) {
        if (h->dtype == VTTS_BF16) break;
        memcpy(host.data() + l.off_w, l.w.data(), l.w.size() * sizeof(float));
        memcpy(host.data() + l.off_b, l.b.data(), l.b.size() * sizeof(float));
        if (l.has_x3 && &l == &h->layers[h->idx_pre]) conv_pre_x3_pack(l.w.data(), reinterpret_cast<unsigned short*>(host.data() + l.off_x3));
        else if (l.has_x3 && l.kind == KIND_CONVT) convt_x3_pack(l.w.data(), l.cin, l.cout, l.k, l.stride, l.pad_a, reinterpret_cast<unsigned short*>(host.data() + l.off_x3));
        else if (l.has_x3) pair_x3_pack(l.w.data(), l.cin, l.k, reinterpret_cast<unsigned short*>(host.data() + l.off_x3));
        if (l.has_wp && l.kind == KIND_CONV)
            conv1d_f32_mfma_pack(l.w.data(), l.cin, l.cout, l.k, reinterpret_cast<float*>(host.data() + l.off_wp));
        else if (l.has_wp)
            convT1d_f32_mfma_pack(l.w.data(), l.cin, l.cout, l.k, l.stride, l.pad_a, reinterpret_cast<float*>(host.data() + l.off_wp));
    }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(dev_blob, host.data(), h->blob_bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));  // `host` dies at return
    h->blob = static_cast<char*>(dev_blob);
    ++h->epoch;  // captured graphs hold the old blob's addresses
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_bind_packed(vtts_hifigan* h, void* dev_blob, size_t blob_bytes) {
    if (!h || !dev_blob) return fail(VTTS_ERR_INVALID, "null argument");
    if (blob_bytes < h->blob_bytes) return fail(VTTS_ERR_NOMEM, "blob too small: %zu < %zu", blob_bytes, h->blob_bytes);
    if ((reinterpret_cast<uintptr_t>(dev_blob) & 255) != 0) return fail(VTTS_ERR_INVALID, "blob must be 256-B aligned");
    h->blob = static_cast<char*>(dev_blob);
    ++h->epoch;  // captured graphs hold the old blob's addresses
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_workspace_bytes(const vtts_hifigan* h, int B, int T, size_t* bytes) {
    if (!h || !bytes) return fail(VTTS_ERR_INVALID, "null argument");
    if (B <= 0 || T <= 0) return fail(VTTS_ERR_INVALID, "B and T must be positive");
    if (int rc = check_pass_size(h, B, T)) return rc;
    const int mb = pick_microbatch(h, B, T);
    const size_t es = h->dtype == VTTS_BF16 ? 2 : sizeof(float);
    const size_t per = align_up(max_act_elems(h, T) * (size_t)mb * es, 256);
    *bytes = chains_parallel(h, B, T) ? (size_t)pass_buffers(h, B, T) * per : (size_t)num_streams(h, B, T) * 4 * per;
    return VTTS_OK;
}

// Small launches are a chain of ~40 short kernels over up to three streams; replaying them as ONE hipGraph removes the host's
// launch / event calls and the cross-stream waits from the critical path (bf16, one 512-frame utterance: 0.81 -> 0.70 ms).  A graph
// bakes in every pointer, so it is keyed by (mel, wav, workspace, B, T) and by an epoch that options and the weight blob bump; it is
// captured the GRAPH_AFTER-th time the same key shows up (the eager runs before it also finish every lazy initialisation: side
// streams, events, function attributes), replayed afterwards, and anything unexpected falls back to the eager path.
constexpr int GRAPH_SLOTS = 8;
constexpr int GRAPH_AFTER = 8;  // capture + instantiate cost about a millisecond, once: only a key that keeps coming back pays it
int forward_maybe_graphed(vtts_hifigan* h, const float* mel, int B, int T, float* wav, void* ws, size_t ws_bytes, hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (!h->opt_graph || h->opt_profile || B <= 0 || T <= 0 || !h->blob || !chains_parallel(h, B, T) ||
        hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone)  // inside the CALLER's capture: just enqueue
        return forward_impl(h, mel, B, T, wav, ws, ws_bytes, s, Taps{});
    if (hipSetDevice(h->device) != hipSuccess) return fail(VTTS_ERR_HIP, "hipSetDevice(%d) failed", h->device);  // graph launches too run on THIS handle's device
    vtts_hifigan::GraphEntry* e = nullptr;
    for (auto& g : h->graphs)
        if (g.mel == mel && g.wav == wav && g.ws == ws && g.B == B && g.T == T) e = &g;
    if (e && e->epoch != h->epoch) {  // stale: start over
        if (e->exec) (void)hipGraphExecDestroy(e->exec);
        if (e->graph) (void)hipGraphDestroy(e->graph);
        e->exec = nullptr;
        e->graph = nullptr;
        e->seen = 0;
        e->epoch = h->epoch;
    }
    if (!e) {
        if ((int)h->graphs.size() < GRAPH_SLOTS) {
            h->graphs.emplace_back();
            e = &h->graphs.back();
        } else {
            e = &h->graphs[0];
            for (auto& g : h->graphs)
                if (g.last_use < e->last_use) e = &g;
            if (e->exec) (void)hipGraphExecDestroy(e->exec);
            if (e->graph) (void)hipGraphDestroy(e->graph);
        }
        *e = vtts_hifigan::GraphEntry{};
        e->mel = mel; e->wav = wav; e->ws = ws; e->B = B; e->T = T; e->epoch = h->epoch;
    }
    e->last_use = ++h->use_clock;
    if (e->exec) {
        hipError_t le = hipGraphLaunch(e->exec, s);
        if (le == hipSuccess) return VTTS_OK;
        (void)hipGetLastError();
        e->seen = -1;  // never again for this key
        (void)hipGraphExecDestroy(e->exec);
        e->exec = nullptr;
        return forward_impl(h, mel, B, T, wav, ws, ws_bytes, s, Taps{});
    }
    if (e->seen < 0 || ++e->seen < GRAPH_AFTER) return forward_impl(h, mel, B, T, wav, ws, ws_bytes, s, Taps{});
    // a key that keeps coming back: record this call's launches into a graph (on a stream of the handle's own: the caller's may be the legacy
    // default stream, which cannot be captured), then launch the graph on the caller's stream
    if (!h->cap_stream && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        h->cap_stream = nullptr;
        e->seen = -1;
        return forward_impl(h, mel, B, T, wav, ws, ws_bytes, s, Taps{});
    }
    if (hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        e->seen = -1;
        return forward_impl(h, mel, B, T, wav, ws, ws_bytes, s, Taps{});
    }
    const int rc = forward_impl(h, mel, B, T, wav, ws, ws_bytes, h->cap_stream, Taps{});
    hipGraph_t g = nullptr;
    hipError_t ce = hipStreamEndCapture(h->cap_stream, &g);
    hipGraphExec_t ex = nullptr;
    if (rc == VTTS_OK && ce == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && hipGraphLaunch(ex, s) == hipSuccess) {
        e->graph = g;
        e->exec = ex;
        return VTTS_OK;
    }
    (void)hipGetLastError();
    if (ex) (void)hipGraphExecDestroy(ex);
    if (g) (void)hipGraphDestroy(g);
    e->seen = -1;
    if (rc) return rc;  // the enqueue itself failed: report that
    return forward_impl(h, mel, B, T, wav, ws, ws_bytes, s, Taps{});  // nothing ran during the capture: run it now
}

VTTS_API int vtts_hifigan_forward(vtts_hifigan* h, const float* mel_dev, int B, int T, float* wav_dev, void* workspace,
                                  size_t workspace_bytes, vtts_stream stream) {
    if (!h || !mel_dev || !wav_dev) return fail(VTTS_ERR_INVALID, "null argument");
    return forward_maybe_graphed(h, mel_dev, B, T, wav_dev, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

VTTS_API int vtts_hifigan_forward_ragged(vtts_hifigan* h, const float* mel_dev, const int32_t* frames_dev, int B, int T, float* wav_dev,
                                         void* workspace, size_t workspace_bytes, vtts_stream stream) {
    if (!h || !mel_dev || !frames_dev || !wav_dev) return fail(VTTS_ERR_INVALID, "null argument");
    if (B <= 0 || T <= 0) return fail(VTTS_ERR_INVALID, "B and T must be positive (got B=%d, T=%d)", B, T);
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(wav_dev, 0, (size_t)B * h->hop * T * sizeof(float), s));  // samples past an utterance's end
    h->cur_lens = frames_dev;
    h->cur_T = T;
    const int rc = forward_impl(h, mel_dev, B, T, wav_dev, workspace, workspace_bytes, s, Taps{});
    h->cur_lens = nullptr;
    h->cur_T = 0;
    return rc;
}

VTTS_API int vtts_hifigan_tap_elems(const vtts_hifigan* h, const char* tap, int B, int T, size_t* elems) {
    if (!h || !tap || !elems) return fail(VTTS_ERR_INVALID, "null argument");
    const vtts_hifigan_cfg& c = h->cfg;
    if (!strcmp(tap, "pre_tanh")) {
        *elems = (size_t)B * h->hop * T;
        return VTTS_OK;
    }
    if (!strcmp(tap, "conv_pre")) {
        *elems = (size_t)B * c.upsample_initial_channel * T;
        return VTTS_OK;
    }
    if (!strncmp(tap, "ups_", 4) || !strncmp(tap, "mrf_", 4)) {
        const int i = atoi(tap + 4);
        if (i < 0 || i >= c.num_upsamples) return fail(VTTS_ERR_INVALID, "tap stage %d out of range", i);
        long L = T;
        for (int q = 0; q <= i; ++q) L *= c.upsample_rates[q];
        *elems = (size_t)B * (c.upsample_initial_channel >> (i + 1)) * L;
        return VTTS_OK;
    }
    return fail(VTTS_ERR_INVALID, "unknown tap '%s'", tap);
}

VTTS_API int vtts_hifigan_forward_tap(vtts_hifigan* h, const float* mel_dev, int B, int T, float* wav_dev, void* workspace,
                                      size_t workspace_bytes, vtts_stream stream, const char* tap, float* tap_dev) {
    if (!h || !mel_dev || !wav_dev || !tap || !tap_dev) return fail(VTTS_ERR_INVALID, "null argument");
    size_t n = 0;
    int rc = vtts_hifigan_tap_elems(h, tap, B, T, &n);
    if (rc) return rc;
    Taps t;
    t.name = tap;
    t.out = tap_dev;
    t.Bfull = B;
    return forward_impl(h, mel_dev, B, T, wav_dev, workspace, workspace_bytes, static_cast<hipStream_t>(stream), t);
}

// device scratch of the test hooks below: released on every return path
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

VTTS_API int vtts_hifigan_run_module(vtts_hifigan* h, const char* key, const float* x_dev, int B, int L, float slope_in,
                                     const float* res_dev, float* y_dev, vtts_stream stream) {
    if (!h || !key || !x_dev || !y_dev) return fail(VTTS_ERR_INVALID, "null argument");
    if (!h->blob) return fail(VTTS_ERR_STATE, "run_module() before pack()/bind_packed()");
    if (B <= 0 || L <= 0) return fail(VTTS_ERR_INVALID, "B and L must be positive");
    Layer* l = find_layer(h, key);
    if (!l) return fail(VTTS_ERR_INVALID, "unknown module '%s'", key);
    const bool is_pre = (l == &h->layers[h->idx_pre]);
    const bool is_post = (l == &h->layers[h->idx_post]);
    if (h->dtype == VTTS_BF16) {
        // the bf16 kernels form LeakyReLU as max(v, slope * v) (bf16_common.h: lrelu_f), valid for slopes in (0, 1]: the model's are 0.1 and 0.01
        if (!(slope_in > 0.0f && slope_in <= 1.0f)) return fail(VTTS_ERR_INVALID, "run_module(): slope_in must lie in (0, 1] on a bf16 handle (got %g)", slope_in);
        // test hook: fp32 channels-last in/out, converted through temporary bf16 buffers
        hipStream_t st = static_cast<hipStream_t>(stream);
        const size_t nx = (size_t)B * L * l->cin, ny = (size_t)B * L * (l->kind == KIND_CONVT ? l->stride : 1) * l->cout;
        DevBuf xbuf, rbuf, ybuf;  // freed on every path (hipFree waits for the device)
        void *&xb = xbuf.p, *&rb = rbuf.p, *&yb = ybuf.p;
        int rc = VTTS_OK;
        if (is_post) {
            HIP_TRY(hipMalloc(&xb, nx * 2));
            if (launch_f32_to_bf16(x_dev, xb, nx, st) != hipSuccess) rc = fail(VTTS_ERR_HIP, "conversion launch failed");
            BConvArgs a;
            memset(&a, 0, sizeof(a));
            a.x = xb;
            a.wf = reinterpret_cast<const float*>(h->blob + l->off_w);
            a.bias = reinterpret_cast<const float*>(h->blob + l->off_b);
            a.B = B;
            a.L = L;
            if (!rc && launch_conv_post_bf16(a, y_dev, nullptr, st) != hipSuccess) rc = fail(VTTS_ERR_HIP, "conv_post launch failed");
        } else {
            HIP_TRY(hipMalloc(&yb, ny * 2));
            if (!is_pre) {
                HIP_TRY(hipMalloc(&xb, nx * 2));
                if (launch_f32_to_bf16(x_dev, xb, nx, st) != hipSuccess) rc = fail(VTTS_ERR_HIP, "conversion launch failed");
            }
            if (res_dev) {
                HIP_TRY(hipMalloc(&rb, ny * 2));
                if (launch_f32_to_bf16(res_dev, rb, ny, st) != hipSuccess) rc = fail(VTTS_ERR_HIP, "conversion launch failed");
            }
            if (!rc) rc = run_layer_bf16(h, *l, is_pre ? static_cast<const void*>(x_dev) : xb, l->cin, l->cin, B, L, slope_in, 1.0f, rb, yb, 0, 1.f, st);
            if (!rc && launch_bf16_to_f32(yb, y_dev, ny, st) != hipSuccess) rc = fail(VTTS_ERR_HIP, "conversion launch failed");
        }
        hipError_t e = hipStreamSynchronize(st);
        if (!rc && e != hipSuccess) rc = fail(VTTS_ERR_HIP, "run_module failed: %s", hipGetErrorString(e));
        return rc;
    }
    Act x = is_pre ? Act{x_dev, (long)L * l->cin, 1, l->cin} : Act{x_dev, (long)l->cin * L, L, 1};
    return run_layer(h, *l, x, B, L, slope_in, res_dev, y_dev, ACC_STORE, 1.f, is_post ? 1 : 0, nullptr,
                     static_cast<hipStream_t>(stream));
}

VTTS_API int vtts_hifigan_run_pair(vtts_hifigan* h, const char* key_c1, const float* x_dev, int B, int L, float* y_dev, vtts_stream stream) {
    if (!h || !key_c1 || !x_dev || !y_dev) return fail(VTTS_ERR_INVALID, "null argument");
    if (!h->blob) return fail(VTTS_ERR_STATE, "run_pair() before pack()/bind_packed()");
    if (B <= 0 || L <= 0) return fail(VTTS_ERR_INVALID, "B and L must be positive");
    Layer* l = find_layer(h, key_c1);
    if (h->x3 && l && l + 1 < h->layers.data() + h->layers.size() && pair_x3_wanted(h, *l, *(l + 1), L))
        return run_pair_x3(h, *l, *(l + 1), x_dev, B, L, y_dev, ACC_STORE, 1.f, static_cast<hipStream_t>(stream));
    if (h->dtype == VTTS_F32) {
        // fp32 handle: x_dev / y_dev are [B, C, L] channel-major (the fp32 engine's layout); asynchronous on `stream`
        if (!l || l + 1 >= h->layers.data() + h->layers.size() || !l->has_wp || !(l + 1)->has_wp || (l + 1)->k != l->k || (l + 1)->dil != 1 ||
            (l + 1)->cin != l->cin || l->cin != l->cout || !pair_f32_supported(l->cin, l->k, l->dil, L))
            return fail(VTTS_ERR_INVALID, "'%s' is not the first convolution of a ResBlock pair the fused fp32 kernel covers (C in {32, 64, 128}, L a multiple of 4)", key_c1);
        return run_pair_f32(h, *l, *(l + 1), x_dev, B, L, y_dev, ACC_STORE, 1.f, static_cast<hipStream_t>(stream));
    }
    if (!l || !l->has_pair) return fail(VTTS_ERR_INVALID, "'%s' is not the first convolution of a fused ResBlock pair", key_c1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)B * L * l->cin;
    DevBuf xbuf, ybuf;
    void *&xb = xbuf.p, *&yb = ybuf.p;
    HIP_TRY(hipMalloc(&xb, n * 2));
    HIP_TRY(hipMalloc(&yb, n * 2));
    int rc = VTTS_OK;
    if (launch_f32_to_bf16(x_dev, xb, n, st) != hipSuccess) rc = fail(VTTS_ERR_HIP, "conversion launch failed");
    if (!rc) rc = run_pair_bf16(h, *l, xb, B, L, 1.0f, yb, 0, 1.f, st);
    if (!rc && launch_bf16_to_f32(yb, y_dev, n, st) != hipSuccess) rc = fail(VTTS_ERR_HIP, "conversion launch failed");
    hipError_t e = hipStreamSynchronize(st);
    if (!rc && e != hipSuccess) rc = fail(VTTS_ERR_HIP, "run_pair failed: %s", hipGetErrorString(e));
    return rc;
}

VTTS_API int vtts_hifigan_set_option(vtts_hifigan* h, const char* name, int64_t value) {
    if (!h || !name) return fail(VTTS_ERR_INVALID, "null argument");
    {  // setting a WRITABLE option to the value it has changes nothing: captured graphs stay valid (read-only and unknown names fall through to their errors)
        static const char* const writable[] = {"kernels", "microbatch", "fuse", "streams", "graph", "zigzag", "chains", "tiles", "tail", "stage"};
        for (const char* w : writable) {
            int64_t cur = 0;
            if (!strcmp(name, w) && vtts_hifigan_get_option(h, name, &cur) == VTTS_OK && cur == value) return VTTS_OK;
        }
    }
    ++h->epoch;  // a captured launch sequence reflects the options it was captured under
    if (!strcmp(name, "kernels")) {
        if (value != 0 && value != 1) return fail(VTTS_ERR_INVALID, "kernels must be 0 (auto) or 1 (generic)");
        h->opt_kernels = value;
    } else if (!strcmp(name, "microbatch")) {
        if (value < 0) return fail(VTTS_ERR_INVALID, "microbatch must be >= 0");
        h->opt_microbatch = value;
    } else if (!strcmp(name, "fuse")) {
        if (value < 0 || value > 3) return fail(VTTS_ERR_INVALID, "fuse must be 0 (per convolution), 1 (pairs), 2 (bf16: pairs + whole ResBlocks where faster; fp32: pairs at C <= 64) or 3 (... wherever supported)");
        h->opt_fuse = value;
    } else if (!strcmp(name, "streams")) {
        if (value < 0 || value > 4) return fail(VTTS_ERR_INVALID, "streams must be 1..4 (0 = the engine's default)");
        h->opt_streams = value;
    } else if (!strcmp(name, "graph")) {
        if (value != 0 && value != 1) return fail(VTTS_ERR_INVALID, "graph must be 0 (always eager) or 1 (small launches replay a captured hipGraph)");
        h->opt_graph = value;
    } else if (!strcmp(name, "zigzag")) {
        if (value != 0 && value != 1) return fail(VTTS_ERR_INVALID, "zigzag must be 0 or 1");
        h->opt_zigzag = value;
    } else if (!strcmp(name, "chains")) {
        if (value < 0 || value > 2) return fail(VTTS_ERR_INVALID, "chains must be 0 (ResBlocks of a stage one after the other), 1 (side by side on small fp32 launches) or 2 (... on every single-micro-batch launch)");
        h->opt_chains = value;
    } else if (!strcmp(name, "tiles")) {
        if (value < 0 || value > 2) return fail(VTTS_ERR_INVALID, "tiles must be 0 (auto), 1 (wide) or 2 (narrow)");
        h->opt_tiles = value;
    } else if (!strcmp(name, "tail")) {
        if (value != 0 && value != 1) return fail(VTTS_ERR_INVALID, "tail must be 0 or 1");
        h->opt_tail = value;
    } else if (!strcmp(name, "stage")) {
        if (value != 0 && value != 1) return fail(VTTS_ERR_INVALID, "stage must be 0 or 1");
        h->opt_stage = value;
    } else if (!strcmp(name, "profile")) {
        h->opt_profile = value ? 1 : 0;
    } else if (!strcmp(name, "hop") || !strcmp(name, "pass_frames") || !strcmp(name, "max_frames_per_pass") || !strcmp(name, "graphs_cached") ||
               !strcmp(name, "profile_C") || !strcmp(name, "profile_K")) {
        return fail(VTTS_ERR_INVALID, "option '%s' is read-only", name);
    } else {
        return fail(VTTS_ERR_INVALID, "unknown option '%s'", name);
    }
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_get_option(const vtts_hifigan* h, const char* name, int64_t* value) {
    if (!h || !name || !value) return fail(VTTS_ERR_INVALID, "null argument");
    if (!strcmp(name, "kernels")) *value = h->opt_kernels;
    else if (!strcmp(name, "microbatch")) *value = h->opt_microbatch;
    else if (!strcmp(name, "profile")) *value = h->opt_profile;
    else if (!strcmp(name, "tiles")) *value = h->opt_tiles;
    else if (!strcmp(name, "streams")) *value = h->opt_streams;
    else if (!strcmp(name, "zigzag")) *value = h->opt_zigzag;
    else if (!strcmp(name, "chains")) *value = h->opt_chains;
    else if (!strcmp(name, "graph")) *value = h->opt_graph;
    else if (!strcmp(name, "tail")) *value = h->opt_tail;
    else if (!strcmp(name, "stage")) *value = h->opt_stage;
    else if (!strcmp(name, "graphs_cached")) {
        *value = 0;
        for (auto& g : h->graphs) *value += g.exec ? 1 : 0;
    }
    else if (!strcmp(name, "fuse")) *value = h->opt_fuse;
    else if (!strcmp(name, "hop")) *value = h->hop;
    else if (!strcmp(name, "max_frames_per_pass")) {
        // the smallest T check_pass_size() refuses (one utterance's largest activation must stay below 2^31 bytes; max_act_elems is linear
        // in T): callers route longer utterances through the chunk scheduler — ONE rule, here
        const size_t per = max_act_elems(h, 1) * (h->dtype == VTTS_BF16 ? 2 : sizeof(float));
        *value = (int64_t)((((size_t)1 << 31) + per - 1) / per);
    }
    else if (!strcmp(name, "pass_frames")) *value = pass_frames(h);
    else if (!strcmp(name, "profile_C")) *value = h->prof_C;
    else if (!strcmp(name, "profile_K")) *value = h->prof_K;
    else return fail(VTTS_ERR_INVALID, "unknown option '%s'", name);
    return VTTS_OK;
}

VTTS_API int vtts_hifigan_profile_read(vtts_hifigan* h, double* resblock_ms, int64_t* launches, double* resblock_flops,
                                       int reset) {
    if (!h) return fail(VTTS_ERR_INVALID, "null argument");
    double ms = 0.0;
    for (size_t i = 0; i < h->prof_used; ++i) {
        float t = 0.f;
        HIP_TRY(hipEventSynchronize(h->prof_events[i].second));
        HIP_TRY(hipEventElapsedTime(&t, h->prof_events[i].first, h->prof_events[i].second));
        ms += t;
    }
    if (resblock_ms) *resblock_ms = ms;
    if (launches) *launches = (int64_t)h->prof_used;
    if (resblock_flops) *resblock_flops = h->prof_flops;
    if (reset) {
        h->prof_used = 0;
        h->prof_flops = 0.0;
    }
    return VTTS_OK;
}

VTTS_API const char* vtts_hifigan_profile_kernel(const vtts_hifigan* h) {
    if (!h) return "";
    return (h->dtype == VTTS_BF16 && h->opt_fuse) ? h->prof_name_pair.c_str() : h->prof_name.c_str();
}
