/* C ABI of the token -> mu half of S3Gen's flow stage on MI355X (gfx950); the entry points live in libs3hift.so next to
 * the vocoder's (include/s3hift_hip.h).
 *
 * It replaces, inside the reference's `CausalMaskedDiffWithXvec.inference`
 * (src/chatterbox_vllm/models/s3gen/flow.py:228-278), everything before the flow-matching decoder is called: the speaker
 * projection (:244-245), the token embedding (:248-250), `UpsampleConformerEncoder.forward`
 * (transformer/upsample_encoder.py:256-299: input projection, pre-lookahead convolutions, six conformer blocks with
 * relative-position self-attention at the token rate, 2x upsampling, four more blocks at the mel rate, final norm) and
 * `encoder_proj` (:258).  One utterance per call (the reference asserts batch 1, flow.py:241), `finalize=True`, full
 * attention (the encoder is built without chunking, s3gen.py:58-73).  All arithmetic is fp32, as in the reference.
 * The flow-matching decoder that consumes `mu` is not part of this library.
 *
 * Plain pointers and sizes only.  *_dev pointers are device memory on the handle's GPU; every function returns 0 on
 * success, s3hift_last_error() (s3hift_hip.h) describes a failure.  Calls on one handle must not overlap.  Kernels are
 * enqueued on `stream` (hipStream_t, NULL = the null stream); only create / finalize / destroy synchronise. */
#ifndef S3FLOW_HIP_H
#define S3FLOW_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S3FLOW_D_MODEL 512
#define S3FLOW_MEL 80
#define S3FLOW_SPK 192

typedef struct s3flow_enc s3flow_enc;

/* Workspace for utterances of up to max_tokens speech tokens (prompt + new), i.e. 2 max_tokens mel frames; the attention
 * scores of the mel-rate blocks take 8 heads x 2T x (2T + 4T) fp32 (96 MB at 500 tokens). */
int s3flow_enc_create(int32_t device, int32_t max_tokens, s3flow_enc** out);
int s3flow_enc_destroy(s3flow_enc* h);

/* One tensor of `S3Token2Wav.flow.state_dict()` (names `input_embedding.weight`, `spk_embed_affine_layer.*`, `encoder.*`,
 * `encoder_proj.*`; `decoder.*` entries are not taken), fp32 host memory, PyTorch shapes. */
int s3flow_enc_load_tensor(s3flow_enc* h, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
int s3flow_enc_finalize(s3flow_enc* h);

/* tokens_dev [T] int32 (prompt tokens followed by the new tokens; ids are clamped to [0, 6560] as flow.py:250 does)
 * -> mu_dev [80][2T] fp32: `encoder_proj(encoder(input_embedding(token)))` transposed, the layout in which the reference
 * passes it to its decoder (`mu=h.transpose(1, 2)`, flow.py:269). */
int s3flow_enc_forward(s3flow_enc* h, const int32_t* tokens_dev, int32_t T, float* mu_dev, void* stream);

/* spk_dev [80] = spk_embed_affine_layer(normalize(xvec_dev [192])) (flow.py:244-245). */
int s3flow_enc_speaker(s3flow_enc* h, const float* xvec_dev, float* spk_dev, void* stream);

/* The encoder's output before `encoder_proj`, [512][2T] (channel-major), of the last forward call: a tap for tests. */
int s3flow_enc_last_hidden(s3flow_enc* h, float* out_dev, int32_t T, void* stream);

#ifdef __cplusplus
}
#endif
#endif
