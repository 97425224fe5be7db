/*
 * vtts_nat.h — C ABI of the MI355X-native NAT duration model of NTT123/vietTTS (the caller-side row next to
 * the mel->waveform hot path: it decides how many mel frames mel2wave() will see).
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

#ifdef __cplusplus
}
#endif
#endif /* VTTS_NAT_H */
