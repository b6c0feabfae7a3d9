/* C ABI of libs3hift.so: the HiFT vocoder (mel spectrogram -> 24 kHz waveform) of S3Gen on MI355X (gfx950).
 *
 * It replaces, inside the reference's token -> waveform stage, what `S3Token2Wav.hift_inference`
 * (src/chatterbox_vllm/models/s3gen/s3gen.py:291-294) runs: `HiFTGenerator.inference` (hifigan.py:462-474) =
 * f0 predictor (f0_predictor.py:52-55) -> harmonic-plus-noise source (hifigan.py:201-231, 267-283) -> decode
 * (hifigan.py:412-444: STFT of the source, conv_pre, three upsampling stages with fused source and residual blocks,
 * conv_post, inverse STFT).  All arithmetic is fp32, as in the reference.
 *
 * Plain pointers and sizes only.  Every *_dev pointer is device memory on the handle's GPU; every function returns 0 on
 * success and non-zero on failure, s3hift_last_error() then describes the failure (thread-local).  Calls on one handle
 * must not overlap (they share the handle's workspace).  Kernels are enqueued on `stream` (a hipStream_t, NULL = the
 * null stream); nothing synchronises except s3hift_create / s3hift_finalize / s3hift_destroy.
 *
 * Layouts: mel [B][80][T] fp32 (the reference's speech_feat), f0 [B][T] (Hz), source and waveform [B][480 T]. */
#ifndef S3HIFT_HIP_H
#define S3HIFT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S3HIFT_ABI_VERSION 1
#define S3HIFT_SAMPLES_PER_FRAME 480
#define S3HIFT_HARMONICS 9 /* fundamental + 8 overtones (hifigan.py:295, 322-328) */

typedef struct s3hift s3hift;

int s3hift_abi_version(void);
const char* s3hift_last_error(void);

/* Workspace for up to max_batch utterances of up to max_frames mel frames each (about 0.35 MB per frame and utterance). */
int s3hift_create(int32_t device, int32_t max_batch, int32_t max_frames, s3hift** out);
int s3hift_destroy(s3hift* h);

/* One tensor of `mel2wav.state_dict()` with the weight-norm parametrisation already folded (hifigan.py:120, 331, 339;
 * names as in the state dict with `.parametrizations.weight.original{0,1}` replaced by `.weight`), fp32 host memory,
 * PyTorch shapes ([out, in, k] for Conv1d, [in, out, k] for ConvTranspose1d).  Replaces load_state_dict for
 * `S3Token2Wav.mel2wav` (s3gen.py:238-245). */
int s3hift_load_tensor(s3hift* h, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* Checks that every tensor arrived, rearranges the convolution weights into the kernels' tile order on the device. */
int s3hift_finalize(s3hift* h);

/* f0 = |classifier(condnet(mel))| per frame (f0_predictor.py:52-55). */
int s3hift_f0(s3hift* h, const float* mel_dev, int32_t B, int32_t T, float* f0_dev, void* stream);

/* Harmonic-plus-noise source from f0 (hifigan.py:455-457, 267-283, 201-231): nearest-neighbour upsampling by 480, nine
 * sines with running phase, voiced / unvoiced mix with Gaussian noise, linear merge, tanh.
 * phase_dev [B][9] (initial phases, entry 0 ignored) and noise_dev [B][9][480 T] (standard normal) are the reference's two
 * random draws; pass NULL for either to have it drawn on the device from `seed` (a counter-based generator: same seed,
 * same source; not torch's stream of numbers). */
int s3hift_source(s3hift* h, const float* f0_dev, int32_t B, int32_t T, uint64_t seed, const float* phase_dev,
                  const float* noise_dev, float* source_dev, void* stream);

/* HiFTGenerator.decode (hifigan.py:412-444): waveform [B][480 T] from mel and source, clamped to +-0.99. */
int s3hift_decode(s3hift* h, const float* mel_dev, const float* source_dev, int32_t B, int32_t T, float* wav_dev,
                  void* stream);

/* HiFTGenerator.inference (hifigan.py:462-474): f0 -> source (drawn from `seed`) -> the first n_cache samples of every
 * source row replaced by cache_source_dev [B][n_cache] (may be NULL / 0) -> decode.  Writes the waveform and the source
 * that was used (the reference returns both). */
int s3hift_inference(s3hift* h, const float* mel_dev, int32_t B, int32_t T, uint64_t seed,
                     const float* cache_source_dev, int32_t n_cache, float* wav_dev, float* source_dev, void* stream);

/* Time of the last s3hift_decode in the handle's workspace, by stage, from HIP events recorded on the call's stream when
 * timing is switched on (it then synchronises at the end of the call).  out_ms[0..3] = STFT + conv_pre, stage 0, 1, 2
 * (upsampling + source fusion + residual blocks); out_ms[4] = conv_post + inverse STFT; out_ms[5] = whole call. */
int s3hift_set_timing(s3hift* h, int32_t on);
int s3hift_last_timing(s3hift* h, float* out_ms, int32_t n);

/* One convolution launch on its own, for per-kernel tests against plain fp32 arithmetic: y = conv1d(act(x), w) + bias
 * (+ res), x [B][Cin][Tin], w [Cout][Cin][K] (host memory; tiled by the call), stride 1, zero padding `pad`, dilation
 * `dil`, Tout = Tin + 2 pad - dil (K - 1).  act: 0 none, 1 Snake with alpha_dev [Cin] (hifigan.py:73-84), 2 leaky ReLU with
 * `slope`, 3 ELU.  Cin must be a multiple of 16.  Allocates and frees its weight buffer and synchronises: a test entry
 * point, not a product path. */
int s3hift_op_conv1d(const float* x_dev, const float* w_host, const float* bias_host, const float* alpha_dev,
                     const float* res_dev, float* y_dev, int32_t B, int32_t Cin, int32_t Cout, int32_t K, int32_t Tin,
                     int32_t dil, int32_t pad, int32_t act, float slope, void* stream);

#ifdef __cplusplus
}
#endif
#endif
