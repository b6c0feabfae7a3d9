/* C ABI of the flow-matching decoder of S3Gen's flow stage (mu -> mel) on MI355X (gfx950); the entry points live in
 * libs3hift.so next to the vocoder's (include/s3hift_hip.h) and the token -> mu encoder's (include/s3flow_hip.h).
 *
 * It replaces the `self.decoder(mu=..., mask=..., spks=..., cond=..., n_timesteps=...)` call of the reference's
 * `CausalMaskedDiffWithXvec.inference` (src/chatterbox_vllm/models/s3gen/flow.py:279-285): `CausalConditionalCFM.forward`
 * (flow_matching.py:203-228), its Euler solver with classifier-free guidance (`ConditionalCFM.solve_euler`, :82-133) and the
 * estimator it calls twice per step, `ConditionalDecoder.forward` (decoder.py:233-317; built by s3gen.py:76-87 with 320 input
 * rows, 256 channels, one down stage, twelve mid stages, one up stage, four transformer blocks of 8 heads x 64 per stage).
 * One utterance per call (the reference asserts batch 1, flow.py:256), all masks ones, full attention
 * (`static_chunk_size` is 0, decoder.py:135); s3cfm_solve_batch runs several such calls in shared launches.  All arithmetic is
 * fp32, as in the reference.
 *
 * Plain pointers and sizes only.  *_dev pointers are device memory on the handle's GPU; every function returns 0 on
 * success, s3hift_last_error() (s3hift_hip.h) describes a failure.  Calls on one handle must not overlap.  Kernels are
 * enqueued on `stream` (hipStream_t, NULL = the null stream); only create / finalize / destroy synchronise. */
#ifndef S3CFM_HIP_H
#define S3CFM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S3CFM_MEL 80
#define S3CFM_MAX_FRAMES 15000
#define S3CFM_MAX_BATCH 8

typedef struct s3cfm s3cfm;

/* Workspace for utterances of up to max_frames mel frames (prompt + new; two per speech token). */
int s3cfm_create(int32_t device, int32_t max_frames, s3cfm** out);
/* The same with room for max_batch utterances per s3cfm_solve_batch call (1 .. S3CFM_MAX_BATCH; about 19 KiB of workspace per
 * frame and utterance). */
int s3cfm_create_batched(int32_t device, int32_t max_frames, int32_t max_batch, s3cfm** out);
int s3cfm_destroy(s3cfm* h);

/* One tensor of `S3Token2Wav.flow.state_dict()` whose name starts with `decoder.estimator.`, fp32 host memory, PyTorch
 * shapes. */
int s3cfm_load_tensor(s3cfm* h, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
int s3cfm_finalize(s3cfm* h);

/* Arithmetic of the matrix launches (no counterpart in the reference, which computes in fp32): on = 0, the default, takes
 * every product on the fp32 matrix instruction; on = 1 holds each fp32 operand of the attention launches as three bf16 values
 * whose sum is the operand and takes six partial products per product on the bf16 matrix instruction with fp32 accumulation
 * (what is left out is below 2^-24 of a product, so results stay as close to fp64 as the fp32 form's; the launch is 1.9x
 * shorter at 1 000 frames); on = 2 does the same in the linear layers and convolutions as well.  Allocates 9 KiB of workspace
 * per frame and utterance on first use, and for on = 2 the weights a second time (6 bytes each); synchronises the device. */
int s3cfm_set_split_bf16(s3cfm* h, int32_t on);

/* One estimator call, `ConditionalDecoder.forward(x, mask, mu, t, spks, cond)` (decoder.py:233-317) on two rows:
 * x_dev, mu_dev, cond_dev [2][80][L], spks_dev [2][80], the same time t for both rows (as solve_euler calls it,
 * flow_matching.py:116) -> out_dev [2][80][L]. */
int s3cfm_estimator(s3cfm* h, const float* x_dev, const float* mu_dev, const float* spks_dev, const float* cond_dev, float t,
                    int32_t L, float* out_dev, void* stream);

/* `CausalConditionalCFM.forward` (flow_matching.py:203-228): mu_dev [80][L], spks_dev [80], cond_dev [80][L] (the prompt's
 * mel frames, then zeros), noise_dev [80][L] (the leading columns of the module's `rand_noise` buffer, already multiplied
 * by the temperature) -> mel_dev [80][L].  t_span is host memory, n_timesteps + 1 values: the time grid of :225-227, which
 * the caller computes (chatterbox_vllm_amd/cfm_weights.py does it with the reference's own torch operations); cfg_rate is
 * `inference_cfg_rate` (0.7, flow_matching.py:26), a double as in the reference: 1 + rate is formed before rounding to fp32. */
int s3cfm_solve(s3cfm* h, const float* mu_dev, const float* spks_dev, const float* cond_dev, const float* noise_dev, int32_t L,
                const float* t_span, int32_t n_timesteps, double cfg_rate, float* mel_dev, void* stream);

/* n_utt independent `CausalConditionalCFM.forward` calls (flow_matching.py:203-228; the reference makes them one after the
 * other, one per utterance: flow.py:256 asserts batch 1) through shared kernel launches, so that short utterances fill the
 * GPU together.  Utterance u has frames[u] mel frames (host memory, 1 .. L); mu_dev, cond_dev, noise_dev, mel_dev are
 * [n_utt][80][L], spks_dev [n_utt][80]; an utterance's columns from frames[u] on must hold finite values on input (zeros) and
 * are unspecified in mel_dev.  Every utterance's result equals its own s3cfm_solve call's up to the rounding of differently
 * ordered fp32 sums (the launches may split their reductions differently at a different size).  s3cfm_solve is this call with
 * n_utt = 1. */
int s3cfm_solve_batch(s3cfm* h, int32_t n_utt, const int32_t* frames, int32_t L, const float* mu_dev, const float* spks_dev,
                      const float* cond_dev, const float* noise_dev, const float* t_span, int32_t n_timesteps, double cfg_rate,
                      float* mel_dev, void* stream);

/* Single operations of the estimator, for tests and timing; both run the launch once, then `iters` more times between two
 * events and report the average in *avg_us (may be NULL).  Tensors hold the two guidance rows: [2][channels][ld] fp32.
 *
 * The linear layers and causal convolutions (decoder.py:94-97 `CausalConv1d`, every `nn.Linear` of the transformer blocks):
 * y[row][co][n] = act(bias[co] + sum_{tap, ci} w[co][ci][tap] x[row][ci][n + tap - pad]) (+ res), w_host [Cout][Cin][K] and
 * bias_host [Cout] in host memory, act = GELU (erf form) when `gelu` is set.  With yt_dev, channels from yt_from on (the last
 * 512) go frame-major to yt_dev [2][ld][576] instead, the layout the attention takes V in, and y_dev holds yt_from channels.
 * With ln_gamma_host / ln_beta_host [Cin] (K = 1) the layer is applied to LayerNorm(x) over the channels of a frame (eps 1e-5,
 * `nn.LayerNorm` as matcha/transformer.py:178, 206 builds norm1 and norm3): the form in which the transformer blocks' norms
 * ride in the q/k/v projection and the first feed-forward layer. */
int s3cfm_op_linear(const float* x_dev, int32_t Cin, int32_t L, int32_t ld, const float* w_host, const float* bias_host, int32_t Cout,
                    int32_t K, int32_t pad, int32_t gelu, const float* res_dev, float* y_dev, float* yt_dev, int32_t yt_from,
                    const float* ln_gamma_host, const float* ln_beta_host, int32_t iters, float* avg_us, void* stream);

/* s3cfm_op_linear with the products taken on split-bf16 operands (the arithmetic of s3cfm_set_split_bf16 mode 2): weights as
 * three bf16 planes made on the GPU, the x values split in registers.  Same arguments and results to the same tolerance. */
int s3cfm_op_linear_split(const float* x_dev, int32_t Cin, int32_t L, int32_t ld, const float* w_host, const float* bias_host, int32_t Cout,
                          int32_t K, int32_t pad, int32_t gelu, const float* res_dev, float* y_dev, float* yt_dev, int32_t yt_from,
                          const float* ln_gamma_host, const float* ln_beta_host, int32_t iters, float* avg_us, void* stream);

/* Self-attention of a transformer block (the diffusers default processor as matcha/transformer.py:267-273 calls it: 8 heads x
 * 64, softmax(q k^T / 8) v, every frame attends to every frame): qk_dev [2][1024][ld] (q in channels 0..511, k in 512..1023,
 * head h at 64 h ..), vt_dev [2][ld][576] frame-major (channels 0..511 used) -> o_dev [2][512][ld].
 * With n_utt > 1 the tensors hold 2 n_utt rows (rows 2 u and 2 u + 1 belong to utterance u) and keys [n_utt] (host memory, may
 * be NULL = L each) gives the number of leading frames of each utterance that are real: queries see only those keys, and
 * the outputs of queries past them are unspecified but finite, as s3cfm_solve_batch uses it. */
int s3cfm_op_attention(const float* qk_dev, const float* vt_dev, int32_t n_utt, const int32_t* keys, int32_t L, int32_t ld, float* o_dev,
                       int32_t iters, float* avg_us, void* stream);

/* The same attention with every fp32 operand held as three bf16 values whose sum is the operand and every product taken
 * as six partial products on the bf16 matrix instruction (fp32 accumulation; what is left out is below 2^-24 of a product):
 * a launch that re-arranges q, k, v into matrix-operand order (timed into split_avg_us), then the attention launch
 * (attn_avg_us).  Same arguments and result layout as s3cfm_op_attention; ld must be a multiple of 32. */
int s3cfm_op_attention_split(const float* qk_dev, const float* vt_dev, int32_t n_utt, const int32_t* keys, int32_t L, int32_t ld,
                             float* o_dev, int32_t iters, float* attn_avg_us, float* split_avg_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif
