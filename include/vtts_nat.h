/*
 * vtts_nat.h — C ABI of the MI355X-native NAT duration and acoustic models of NTT123/vietTTS (the caller-side rows
 * next to the mel->waveform hot path: they decide how many mel frames mel2wave() will see, and produce them).
 *
 * Entry points a maintainer of the reference would bind (ctypes stub in INTEGRATION.md) to replace the body of
 *     vietTTS/nat/text2mel.py:22-34   predict_duration(tokens)
 * i.e. "load duration_latest_ckpt.pickle, build DurationModel(is_training=False), apply it to the token ids"
 * (vietTTS/nat/model.py:9-70).  Same conventions as vtts_hifigan.h: plain pointers and sizes, 0 / negative
 * vtts_status (message via vtts_last_error()), device memory owned by the caller, asynchronous on the given stream.
 */
#ifndef VTTS_NAT_H
#define VTTS_NAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vietTTS/nat/config.py:11-13: the fields DurationModel.__init__ / TokenEncoder.__init__ read. */
typedef struct vtts_nat_duration_cfg {
    int32_t vocab_size;        /* 256 */
    int32_t lstm_dim;          /* 256: embedding width = conv channels = LSTM units */
} vtts_nat_duration_cfg;

typedef struct vtts_nat_duration vtts_nat_duration; /* opaque */

/* Replaces: DurationModel(is_training=False) construction, vietTTS/nat/model.py:56-66 (via text2mel.py:23-26). */
int vtts_nat_duration_create(const vtts_nat_duration_cfg* cfg, int device, vtts_nat_duration** out);
void vtts_nat_duration_destroy(vtts_nat_duration* h);

/*
 * Supply one array of the checkpoint (text2mel.py:27-28: dic["params"] and dic["aux"]) by the TAIL of its Haiku
 * module path and its name inside that module, e.g.
 *   ("token_encoder/~/embed", "embeddings") [V,D]      ("token_encoder/~/conv1_d_2", "w") [3,D,D] / "b" [D]
 *   ("token_encoder/~/batch_norm_1", "scale" | "offset") [1,1,D]
 *   ("token_encoder/~/batch_norm_1/~/mean_ema" | ".../~/var_ema", "average") [1,1,D]     (state)
 *   ("token_encoder/~/lstm/linear" | "token_encoder/~/lstm_1/linear", "w") [2D,4D] / "b" [4D]   (forward | backward)
 *   ("linear", "w") [2D,D] / "b" [D]        ("linear_1", "w") [D,1] / "b" [1]
 * fp32 host memory, copied before return.  num_params()/param_info() enumerate what is expected.
 */
int vtts_nat_duration_set_param(vtts_nat_duration* h, const char* module, const char* name, const float* host,
                                const int64_t* shape, int ndim);
int vtts_nat_duration_num_params(const vtts_nat_duration* h, int* n);
int vtts_nat_duration_param_info(const vtts_nat_duration* h, int i, const char** module, const char** name,
                                 int64_t shape[3], int* ndim);

/* Packed device blob (caller-owned, 256-B aligned), as in vtts_hifigan.h. */
int vtts_nat_duration_packed_bytes(const vtts_nat_duration* h, size_t* bytes);
int vtts_nat_duration_pack(vtts_nat_duration* h, void* dev_blob, size_t blob_bytes, void* stream);
int vtts_nat_duration_bind_packed(vtts_nat_duration* h, void* dev_blob, size_t blob_bytes);

/* Scratch bytes forward() needs for B sentences of at most Lmax tokens. */
int vtts_nat_duration_workspace_bytes(const vtts_nat_duration* h, int B, int Lmax, size_t* bytes);

/*
 * Replaces forward_fn(params, aux, rng, DurationInput(tokens[None], [len], None))[0], text2mel.py:29-34 ==
 * DurationModel.__call__, model.py:68-70, one sentence per row (the reference runs batch 1; rows are independent):
 *   tokens_dev    [B, Lmax] int32 token ids (text2tokens, text2mel.py:37-58); entries past a row's length are ignored
 *   lengths_dev   [B] int32, 1 <= length <= Lmax
 *   durations_dev [B, Lmax] fp32 seconds per token; entries past a row's length are set to 0
 */
int vtts_nat_duration_forward(vtts_nat_duration* h, const int32_t* tokens_dev, const int32_t* lengths_dev, int B, int Lmax,
                              float* durations_dev, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Acoustic model: replaces predict_mel()'s network apply, vietTTS/nat/text2mel.py:61-82 ==
 * AcousticModel(is_training=False).inference(tokens, durations, n_frames), vietTTS/nat/model.py:128-151.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct vtts_nat_acoustic_cfg { /* vietTTS/nat/config.py:11-17, :42; model.py:88-89 */
    int32_t vocab_size;   /* 256 */
    int32_t encoder_dim;  /* acoustic_encoder_dim 256 */
    int32_t decoder_dim;  /* acoustic_decoder_dim 512 */
    int32_t prenet_dim;   /* 256 (hk.Linear(256) x 2) */
    int32_t mel_dim;      /* 80  */
    int32_t postnet_dim;  /* 512 */
} vtts_nat_acoustic_cfg;

typedef struct vtts_nat_acoustic vtts_nat_acoustic; /* opaque */

int vtts_nat_acoustic_create(const vtts_nat_acoustic_cfg* cfg, int device, vtts_nat_acoustic** out);
void vtts_nat_acoustic_destroy(vtts_nat_acoustic* h);
/* Arrays by Haiku module tail under "acoustic_model/~/": "token_encoder/~/..." as above; "lstm/linear", "lstm_1/linear"
 * (decoder layers), "linear" (mel projection), "linear_1" / "linear_2" (prenet, "w" only), "conv1_d" .. "conv1_d_4" and
 * "batch_norm" .. "batch_norm_3" (+ "/~/mean_ema", "/~/var_ema" state) of the postnet. */
int vtts_nat_acoustic_set_param(vtts_nat_acoustic* h, const char* module, const char* name, const float* host,
                                const int64_t* shape, int ndim);
int vtts_nat_acoustic_num_params(const vtts_nat_acoustic* h, int* n);
int vtts_nat_acoustic_param_info(const vtts_nat_acoustic* h, int i, const char** module, const char** name,
                                 int64_t shape[3], int* ndim);
int vtts_nat_acoustic_packed_bytes(const vtts_nat_acoustic* h, size_t* bytes);
int vtts_nat_acoustic_pack(vtts_nat_acoustic* h, void* dev_blob, size_t blob_bytes, void* stream);
int vtts_nat_acoustic_bind_packed(vtts_nat_acoustic* h, void* dev_blob, size_t blob_bytes);
int vtts_nat_acoustic_workspace_bytes(const vtts_nat_acoustic* h, int B, int Lmax, int Fmax, size_t* bytes);
/* Options (defaults in brackets):
 *   "bf16x3" [0]  1 = the matrix products of the decoder's LSTM steps, of the gate GEMM and of the postnet as three bf16 x bf16 terms on the
 *                 bf16 matrix pipe (every operand v = v0 + v1, v0 = bf16(v), v1 = bf16(v - v0); x * w ~ x1 w0 + x0 w1 + x0 w0, fp32
 *                 accumulation; the decoder state is kept split, the cell states, the projection and the prenet stay fp32): the mel moves
 *                 by ~1e-5 of its range (tests/test_gpu_nat.py) and the acoustic model runs a third faster.  The default keeps every
 *                 product in fp32 — the mode the parity tests against the reference pin at 5e-5.  For callers whose vocoder is
 *                 bf16-class anyway.  The token encoders and the duration model are fp32 in both modes (integer frame counts).
 * Unknown keys and out-of-range values return VTTS_ERR_INVALID. */
int vtts_nat_acoustic_set_option(vtts_nat_acoustic* h, const char* key, int value);
int vtts_nat_acoustic_get_option(const vtts_nat_acoustic* h, const char* key, int* value);
/*
 *   tokens_dev    [B, Lmax] int32, lengths_dev [B] int32                       as for the duration model
 *   durations_dev [B, Lmax] fp32, in FRAMES (text2mel.py:78: seconds * sample_rate / hop)
 *   nframes_dev   [B] int32: frames to generate per sentence (text2mel.py:79), <= Fmax
 *   keep_dev      [B, Fmax, 2, prenet_dim] bytes: the prenet's two dropout KEEP masks per frame (1 = keep and scale by 2;
 *                 model.py:95-100 — dropout is on at inference), or NULL for no dropout.  The reference draws them from
 *                 JAX's threefry PRNG through Haiku's per-scan-step key splitting; that stream is the caller's business.
 *   mel_dev       [B, Fmax, mel_dim] fp32 log-mel (decoder output + postnet residual); rows past nframes are zero
 */
/*
 * Draws keep masks for forward() on the device: keep_dev [B, Fmax, 2, prenet_dim] bytes, P(keep) = 1/2 (hk.dropout(key,
 * 0.5, x), model.py:97,:99), from seeds_dev [B] (one 64-bit seed per sentence, so a sentence's masks do not depend on
 * the batch it is in) with Threefry-2x32-20: key = seed, counter = (2 * frame + layer, 64-column block), output bit j =
 * column 64 * block + j.  The cipher is the one jax.random uses; the STREAM is this library's own (per-sentence seeds:
 * independent of batching, which a throughput pipeline wants) — the reference's own stream is the next entry point.
 */
int vtts_nat_acoustic_keep_masks(const vtts_nat_acoustic* h, const uint64_t* seeds_dev, int B, int Fmax, uint8_t* keep_dev, void* stream);
/*
 * The same masks as the REFERENCE draws them (text2mel.py:65-73 -> model.py:95-100,134-142): (rng_key0, rng_key1) = the
 * checkpoint's `rng` (a jax.random.PRNGKey, uint32[2]); dm-haiku's PRNGSequence hands out S_n of the chain
 * (K_n, S_n) = jax.random.split(K_{n-1}), frame f takes S_{2f+1} and S_{2f+2}, a mask is
 * jax.random.bernoulli(S, 0.5, (1, prenet_dim)) on jax.random's classic (pre-0.5 default, non-"partitionable") threefry
 * layout.  Every sentence of the batch gets the same masks, as every run of the reference starts from the same key.
 * Restatement: oracle/nat_oracle.py::haiku_prenet_keep_masks (pinned by the known answers JAX's documentation prints for
 * PRNGKey(0); not by a JAX run: none is possible offline).  A checkpoint run under JAX >= 0.5 defaults draws another stream.
 */
int vtts_nat_acoustic_keep_masks_haiku(const vtts_nat_acoustic* h, uint32_t rng_key0, uint32_t rng_key1, int B, int Fmax, uint8_t* keep_dev,
                                       void* stream);
/*
 * ... with the threefry layout as a parameter: threefry_partitionable = 0 is the entry point above; 1 is the layout JAX >= 0.5 uses by
 * default (jax_threefry_partitionable=True: subkey i of a split and element c of a 32-bit draw are the cipher on the 64-bit index
 * itself, a draw's word = y0 ^ y1).  The reference pins no JAX version (setup.py:6-19), so which stream a given checkpoint was
 * sampled with at inference depends on the JAX it ran under.  Mode 1 is restated from recollection of jax/_src/prng.py
 * (oracle/nat_oracle.py::jax_partitionable_*) and is NOT pinned by any known answer: use it knowingly.
 */
int vtts_nat_acoustic_keep_masks_haiku_mode(const vtts_nat_acoustic* h, uint32_t rng_key0, uint32_t rng_key1, int threefry_partitionable, int B,
                                            int Fmax, uint8_t* keep_dev, void* stream);
int vtts_nat_acoustic_forward(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev,
                              const float* durations_dev, const int32_t* nframes_dev, int B, int Lmax, int Fmax,
                              const uint8_t* keep_dev, float* mel_dev, void* workspace, size_t workspace_bytes, void* stream);
/*
 * forward() that hands the mel over in GROUPS of rows as the decoder finishes them (the text -> waveform pipeline, BASELINE configs[3];
 * the reference synthesises one sentence per process, vietTTS/synthesizer.py:33-39, and has nothing to overlap).  The decoder is a chain
 * of Fmax dependent steps that barely loads the chip; a sentence's mel is complete after ITS last frame, long before the batch's.
 *   group_row0   HOST [ngroups + 1]: group g = rows [group_row0[g], group_row0[g + 1]); group_row0[0] = 0, group_row0[ngroups] = B
 *   group_frames HOST [ngroups]: the largest nframes among the group's rows (1 .. Fmax)
 * The arithmetic of every row is that of forward() (same kernels, same order: bit-identical mel).  After decoder frame
 * group_frames[g] - 1 the group's postnet runs on a stream of the handle's own beside the remaining decoder steps, and
 * vtts_nat_acoustic_wait_group(h, g, consumer_stream) makes `consumer_stream` wait for exactly that (rows of group g of mel_dev
 * complete) — e.g. the stream a vocoder runs on.  `stream` itself waits for every group before the call's work on it ends, so code
 * that ignores the groups sees forward()'s semantics.  Sort the rows by descending nframes to make the groups finish one after another.
 */
int vtts_nat_acoustic_forward_groups(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev,
                                     const float* durations_dev, const int32_t* nframes_dev, int B, int Lmax, int Fmax,
                                     const uint8_t* keep_dev, float* mel_dev, void* workspace, size_t workspace_bytes, void* stream,
                                     int ngroups, const int32_t* group_row0, const int32_t* group_frames);
/* Valid for the groups of the handle's LAST forward_groups() call (VTTS_ERR_STATE otherwise). */
int vtts_nat_acoustic_wait_group(vtts_nat_acoustic* h, int group, void* stream);
/*
 * The token encoder alone, then the rest from its output (the text -> waveform pipeline: the encoder needs the tokens only, so it runs while
 * the host still turns the duration model's seconds into frame counts).  enc_dev [B, Lmax, 2 * encoder_dim] fp32.  A row of it does not
 * depend on its batch: rows may be re-ordered or dropped between the two calls (B and the row order of forward_from_encoder() are its own;
 * Lmax must be encode()'s).  encode() needs workspace_bytes(h, B, Lmax, 1).  forward_from_encoder(ngroups = 0) gives forward()'s mel,
 * ngroups >= 1 forward_groups()'s hand-over: the same mel, bit for bit.
 */
int vtts_nat_acoustic_encode(vtts_nat_acoustic* h, const int32_t* tokens_dev, const int32_t* lengths_dev, int B, int Lmax, float* enc_dev,
                             void* workspace, size_t workspace_bytes, void* stream);
int vtts_nat_acoustic_forward_from_encoder(vtts_nat_acoustic* h, const float* enc_dev, const int32_t* lengths_dev, const float* durations_dev,
                                           const int32_t* nframes_dev, int B, int Lmax, int Fmax, const uint8_t* keep_dev, float* mel_dev,
                                           void* workspace, size_t workspace_bytes, void* stream, int ngroups, const int32_t* group_row0,
                                           const int32_t* group_frames);

#ifdef __cplusplus
}
#endif
#endif /* VTTS_NAT_H */
