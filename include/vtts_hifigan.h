/*
 * vtts_hifigan.h — C ABI of the MI355X-native mel->waveform hot path of NTT123/vietTTS.
 *
 * This is the drop-in boundary: the entry points a maintainer of the reference would bind
 * (ctypes stub in INTEGRATION.md) to replace the body of
 *     vietTTS/hifigan/mel2wave.py:20-41   mel2wave(mel)
 * i.e. "load hk_hifi.pickle, build Generator(h), apply it to mel".  Plain pointers and sizes
 * only; no torch / pybind types.  Every function returns 0 on success or a negative
 * vtts_status; the message for the last failure on the calling thread is
 * vtts_last_error().  Nothing throws across this boundary.
 *
 * Threading: a handle is not thread-safe; use one handle per GPU / stream.  forward() is
 * asynchronous on the given hipStream_t.  Device memory (packed weights, workspace, mel, wav)
 * is owned by the CALLER (PyTorch-ROCm tensors in the shipped host layer); the handle owns only
 * small host-side tables.
 */
#ifndef VTTS_HIFIGAN_H
#define VTTS_HIFIGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTTS_ABI_VERSION 2

typedef enum vtts_status {
    VTTS_OK = 0,
    VTTS_ERR_INVALID = -1,     /* bad argument / unsupported configuration            */
    VTTS_ERR_STATE = -2,       /* call order violated (e.g. forward before bind)       */
    VTTS_ERR_MISSING = -3,     /* a parameter array was never supplied                 */
    VTTS_ERR_HIP = -4,         /* a HIP runtime call failed (message has the hipError) */
    VTTS_ERR_NOMEM = -5,       /* workspace too small / host allocation failed         */
    VTTS_ERR_SHAPE = -6        /* array shape does not match the architecture          */
} vtts_status;

typedef enum vtts_dtype {
    VTTS_F32 = 0,  /* fp32 operands, fp32 accumulate: v_mfma_f32_32x32x2_f32 (exact fmaf chain) */
    VTTS_BF16 = 1, /* bf16 operands, fp32 accumulate: v_mfma_f32_32x32x16_bf16                  */
    VTTS_BF16X3 = 2 /* the fp32 engine (fp32 activations in HBM, same layouts, same entry points) with the ResBlock convolutions — 96.8 % of
                       the FLOPs — on the bf16 matrix pipe with SPLIT operands: every product is three bf16 x bf16 terms (x1 w0 + x0 w1 + x0 w0,
                       v = v0 + v1 the two-bf16-term split of an fp32 value), fp32 accumulate.  fp32-grade: whole-generator max-abs 1.5e-5 (measured at 64 x 1024 frames: BENCH_r05.json, bf16x3_path.parity), 3.0x the fp32 engine's throughput against
                       the reference (5x inside BASELINE.json's 1e-4) at 3/16 of the fp32 MFMA time (kernels_x3.hip, profiles/r04_b_split_findings.md) */
} vtts_dtype;

#define VTTS_MAX_UPSAMPLES 8
#define VTTS_MAX_KERNELS 4

/*
 * The architecture-defining fields Generator.__init__ reads from the JSON config
 * (vietTTS/hifigan/model.py:81-106; assets/hifigan/config.json:2,11-15,19).
 * resblock: 1 = ResBlock1 (model.py:13-51; three dilations per kernel size; every engine), 2 = ResBlock2 (model.py:54-74;
 * two dilations per kernel size, the third entry is ignored; every engine — VTTS_BF16 for the V1 channel / kernel-size shapes its
 * kernels cover, as for ResBlock1, on the per-convolution kernel).  0 is read as 1.
 */
typedef struct vtts_hifigan_cfg {
    int32_t num_mels;                                        /* 80  */
    int32_t upsample_initial_channel;                        /* 512 */
    int32_t num_upsamples;                                   /* 4   */
    int32_t upsample_rates[VTTS_MAX_UPSAMPLES];              /* 8,8,2,2     */
    int32_t upsample_kernel_sizes[VTTS_MAX_UPSAMPLES];       /* 16,16,4,4   */
    int32_t num_kernels;                                     /* 3   */
    int32_t resblock_kernel_sizes[VTTS_MAX_KERNELS];         /* 3,7,11      */
    int32_t resblock_dilation_sizes[VTTS_MAX_KERNELS][3];    /* 1,3,5 each  */
    int32_t resblock;                                        /* 1 (config.json "resblock": "1") or 2 */
} vtts_hifigan_cfg;

typedef struct vtts_hifigan vtts_hifigan; /* opaque */
typedef void* vtts_stream;                /* a hipStream_t (0 = default stream) */

/* ABI version of the loaded library (== VTTS_ABI_VERSION of the header it was built from). */
int vtts_abi_version(void);

/* Message describing the last error raised on this thread ("" if none). */
const char* vtts_last_error(void);

/*
 * Build the host-side plan for one generator on HIP device `device`.
 * Replaces: Generator(h) construction, vietTTS/hifigan/model.py:78-107 (via mel2wave.py:28-31).
 */
int vtts_hifigan_create(const vtts_hifigan_cfg* cfg, int device, int dtype, vtts_hifigan** out);
void vtts_hifigan_destroy(vtts_hifigan* h);

/*
 * Supply one parameter array in the layout of hk_hifi.pickle (mel2wave.py:35-36; produced by
 * convert_torch_model_to_haiku.py:50-58): key e.g. "generator/~/res_block1_4/~/convs1_2",
 * which = "w" ([K,Cin,Cout] for convs, [K,Cout,Cin] for ups_*) or "b" ([Cout]).  `host` is
 * fp32 host memory, copied before return.
 */
int vtts_hifigan_set_param(vtts_hifigan* h, const char* key, const char* which, const float* host,
                           const int64_t* shape, int ndim);

/* Number of parameter arrays the architecture needs (156 for V1) and the i-th one's key. */
int vtts_hifigan_num_params(const vtts_hifigan* h, int* n);
int vtts_hifigan_param_info(const vtts_hifigan* h, int i, const char** key, const char** which,
                            int64_t shape[3], int* ndim);

/*
 * Size of the packed (kernel-private layout) weight blob, and the two ways to get one bound:
 *  - pack():  re-lay-out every array supplied via set_param into `dev_blob` (device memory of
 *             packed_bytes() bytes, 256-B aligned, caller-owned) on `stream`, and bind it;
 *  - bind_packed(): bind a blob that already holds packed weights — e.g. one received by an
 *             RCCL broadcast from the rank that called pack().  The packing depends only on
 *             (cfg, dtype, ABI version), so blobs are interchangeable between ranks.
 * Replaces: pickle.load of the parameters on every call, mel2wave.py:35-36.
 */
int vtts_hifigan_packed_bytes(const vtts_hifigan* h, size_t* bytes);
int vtts_hifigan_pack(vtts_hifigan* h, void* dev_blob, size_t blob_bytes, vtts_stream stream);
int vtts_hifigan_bind_packed(vtts_hifigan* h, void* dev_blob, size_t blob_bytes);

/* Scratch bytes forward() needs for a batch of B utterances of T mel frames.  One pass takes utterances whose largest
 * activation is below 2^31 bytes (V1: T < 131072 frames on the bf16 engine = 35 minutes at 16 kHz, T < 65536 on the fp32
 * engine): longer ones return VTTS_ERR_INVALID here and in forward() and go through the chunk scheduler (13-frame halo: the
 * generator's receptive field). */
int vtts_hifigan_workspace_bytes(const vtts_hifigan* h, int B, int T, size_t* bytes);

/*
 * The hot path.  Replaces forward.apply(params, aux, rng, mel) + squeeze,
 * vietTTS/hifigan/mel2wave.py:37-39 == Generator.__call__, model.py:109-125.
 *   mel_dev : [B, T, num_mels] fp32, NWC (the reference's layout), device memory
 *   wav_dev : [B, hop*T] fp32 in (-1,1), device memory (hop = prod(upsample_rates) = 256)
 * Asynchronous on `stream`; the caller synchronises before reading wav_dev.
 */
int vtts_hifigan_forward(vtts_hifigan* h, const float* mel_dev, int B, int T, float* wav_dev,
                         void* workspace, size_t workspace_bytes, vtts_stream stream);

/*
 * forward() over utterances of DIFFERENT lengths in one batch (every dtype; VTTS_F32 / VTTS_BF16X3 since round 5: the sentence
 * pipeline at the reference's 1e-4): utterance b has frames_dev[b] mel
 * frames (1 <= frames <= T) at the start of its [T, num_mels] slot.  Its first hop*frames[b] samples are exactly what
 * forward() returns for that utterance alone (B = 1, T = frames[b]) — every layer treats the rows past the utterance's
 * end as the reference's zero padding (model.py:8-10, "SAME") and skips the tiles beyond it — and the rest of its
 * [hop*T] slot is zero.  (Bit for bit on VTTS_BF16 and VTTS_BF16X3; on VTTS_F32 for frame counts that are multiples of 4 — alone, another
 * count routes ups_0 through the generic kernel, whose fmaf chain runs in another order: ~1e-7.)  The reference has no batching at all (mel2wave.py:20-41 runs one utterance); this is the
 * throughput form of running it once per sentence.
 *   frames_dev : [B] int32, device memory.  The counts are read on the device only (no host round trip); a value outside
 *                [0, T] is CLAMPED into it by every kernel, so a bad count can shorten or lengthen an utterance inside its
 *                own slot but never reads or writes outside the slot.
 * Threading: a handle serves ONE call at a time (the ragged state and the profiling counters live on it); forward() makes
 * the handle's device current (hipSetDevice) and always re-joins its side streams into `stream`, also when a launch fails.
 */
int vtts_hifigan_forward_ragged(vtts_hifigan* h, const float* mel_dev, const int32_t* frames_dev, int B, int T, float* wav_dev,
                                void* workspace, size_t workspace_bytes, vtts_stream stream);

/*
 * forward() that also copies one intermediate out, for parity tests.  `tap` names follow the
 * oracle: "conv_pre", "ups_<i>", "mrf_<i>" (each [B, C, L] fp32, channel-major) and
 * "pre_tanh" ([B, hop*T]).  tap_dev must hold vtts_hifigan_tap_elems() floats.
 * A VTTS_BF16 handle returns taps channels-last ([B, L, C], converted to fp32), and "conv_pre" /
 * "mrf_<i>" hold the tensor as stored, i.e. already through the consumer's LeakyReLU (0.1; 0.01 for
 * the last stage).
 */
int vtts_hifigan_tap_elems(const vtts_hifigan* h, const char* tap, int B, int T, size_t* elems);
int vtts_hifigan_forward_tap(vtts_hifigan* h, const float* mel_dev, int B, int T, float* wav_dev,
                             void* workspace, size_t workspace_bytes, vtts_stream stream,
                             const char* tap, float* tap_dev);

/*
 * Run ONE convolution module of the generator on caller-provided activations (per-layer known
 * answer tests).  x_dev/y_dev/res_dev are [B, C, L] fp32 channel-major (the engine's internal
 * layout); for "generator/~/conv1_d" x_dev is [B, L, num_mels] as at the boundary.
 *   slope_in : LeakyReLU slope applied to the input on load (1.0 = none)
 *   res_dev  : optional residual added to the output (may alias y_dev), or NULL
 * The output length is L for convolutions and stride*L for ups_*.
 * For a VTTS_BF16 handle the arrays are fp32 but CHANNELS-LAST ([B, L, C], the bf16 path's internal
 * layout), are rounded to bf16 on the way in, and the call synchronises the stream.
 */
int vtts_hifigan_run_module(vtts_hifigan* h, const char* key, const float* x_dev, int B, int L,
                            float slope_in, const float* res_dev, float* y_dev, vtts_stream stream);

/*
 * VTTS_BF16 handles only: run ONE fused ResBlock1 pair  x' = convs2_z(lrelu(convs1_z(lrelu(x)))) + x
 * (model.py:45-50) named by its first convolution's key.  x_dev / y_dev are fp32 [B, L, C]
 * channels-last, rounded to bf16 on the way in; the call synchronises the stream.
 */
int vtts_hifigan_run_pair(vtts_hifigan* h, const char* key_c1, const float* x_dev, int B, int L,
                          float* y_dev, vtts_stream stream);

/*
 * Engine options (tests / benchmarks):
 *   "kernels"   0 = auto (MFMA kernels where the shape allows, generic otherwise), 1 = generic only
 *   "microbatch" utterances processed per pass through the network (0 = auto)
 *   "fuse"      bf16: 2 = fused ResBlock pairs + the whole-ResBlock kernel of the C = 32 stage where it is the faster
 *               one (default), 3 = ... wherever it is supported, 1 = fused pairs only, 0 = one kernel per convolution;
 *               bf16x3: 2 (default) = split-operand pairs + the whole-ResBlock kernel where it is faster (C = 32; C = 64 at k <= 7; C = 128 at k = 3),
 *               3 = ... wherever it exists, 1 = pairs only, 0 = the fp32 engine's kernels; fp32: 1 = fused pairs at C <= 64, 2 (default) = + C = 128 at
 *               k = 3, 3 = every pair the kernel covers
 *   "streams"   1..4: consecutive micro-batches run on separate HIP streams (forked from / joined to
 *               the caller's stream with events) so HBM phases of one overlap MFMA phases of another;
 *               0 (default) = the engine's choice: two streams, a large batch (more than 32768 frames) as equal micro-batches of at
 *               most 32768 frames (same samples as one pass: micro-batches are independent); 1 = everything on the caller's stream
 *               (what bench.py's roofline calibration pass and the rocprofv3 profiles use: a kernel's duration is its own)
 *   "chains"    1 (default) = on a small single-micro-batch launch (B * T <= 2048 frames) the ResBlocks of a stage run side by side on
 *               parallel streams with scratch of their own: the fp32 engine combines their outputs afterwards, the bf16 engine chains the
 *               accumulating epilogues by events in the sequential order — the same additions (and bf16 roundings) in the same
 *               order, bit-identical samples; 0 = one after the other; 2 = side by side on every single-micro-batch launch.
 *               workspace_bytes() depends on it.
 *   "graph"     1 (default) = a small launch (as above) whose (mel, wav, workspace, B, T) came back 8 times is captured into a hipGraph
 *               and replayed from then on (about a millisecond once, then no host launch / event calls in the latency path: one 512-frame
 *               utterance 0.81 -> 0.69 ms bf16, 4.57 -> 3.65 ms fp32 together with "chains"); dropped when an option or the weight
 *               blob changes; inside a caller's own stream capture the launches are simply enqueued.  0 = always eager.
 *   "tail"      VTTS_BF16: 1 (default) = the generator's last pair launch (stage 4, C = 32, k = 11) also runs conv_post + tanh on the rows it
 *               produces — the stage output is never written and the streaming conv_post kernel is not launched; bit-identical samples;
 *               0 = the separate kernel.  (forward_tap always takes the separate kernel: a tap wants the stage output.)
 *   "stage"     VTTS_BF16: 1 = the generator's whole LAST stage (ups_3's output -> three ResBlocks -> MRF mean -> LeakyReLU(0.01) -> conv_post -> tanh) is one
 *               launch over LDS-resident windows (kernels_bf16_stage.hip); bit-identical samples.  0 (default): it measured 30 % slower than the launches it
 *               replaces (profiles/r06_a_stage_kernel_findings.md).  forward_tap always takes the other path.
 *   "tiles"     MFMA time-tile width: 0 = by problem size, 1 = wide, 2 = narrow
 *   "zigzag"    1 (default) = consecutive launches walk the batch in alternating directions, so that a launch starts with the
 *               utterances its producer wrote last (still in the 256 MB Infinity Cache); same samples either way.  0 = always ascending.
 *   "profile"   1 = bracket the dominant kernel class with hipEvents (see profile_read); set "streams" = 1 with it — under the default two
 *               streams a bracketed kernel shares the chip with the other micro-batch's and its duration is not its own
 * Read-only (get_option): "hop" (samples per mel frame), "max_frames_per_pass" (the SMALLEST T the engine refuses: forward() takes
 * T < max_frames_per_pass, an utterance of that many frames or more goes through the chunk scheduler), "pass_frames" (mel frames per call
 * the launches are sized for: schedulers that build batches aim at it), "graphs_cached".
 * Setting an option to the value it already has is a no-op (captured graphs stay valid).
 */
int vtts_hifigan_set_option(vtts_hifigan* h, const char* name, int64_t value);
int vtts_hifigan_get_option(const vtts_hifigan* h, const char* name, int64_t* value);

/*
 * Kernel-level timing hook for bench.py: when enabled ("profile" option = 1), forward()
 * brackets the dominant kernel class (ResBlock convolutions) with hipEvents on `stream`;
 * after a stream sync, this returns the accumulated milliseconds and launch count since the
 * last reset, and the algorithmic FLOPs those launches performed.
 */
int vtts_hifigan_profile_read(vtts_hifigan* h, double* resblock_ms, int64_t* launches,
                              double* resblock_flops, int reset);
/* Name prefix of the kernel those events bracket (as rocprofv3 --kernel-trace prints it). */
const char* vtts_hifigan_profile_kernel(const vtts_hifigan* h);

#ifdef __cplusplus
}
#endif
#endif /* VTTS_HIFIGAN_H */
