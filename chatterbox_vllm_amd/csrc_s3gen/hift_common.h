// Shared declarations of libs3hift.so (HiFT vocoder on gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/s3hift_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_SNAKE = 1, ACT_LRELU = 2, ACT_ELU = 3, ACT_SWISH = 4 };
enum { ACC_NONE = 0, ACC_SET = 1, ACC_ADD = 2, ACC_ADD_DIV3 = 3 };

#define CONV_CK 16          // input channels per LDS tile
#define CONV_MAX_HALO 52    // (K - 1) * dilation of the widest layer is 50

// One convolution launch.  Tensors are [batch][channel][time] fp32 with a row stride (x_ld, y_ld) and a batch stride.
struct ConvArgs {
    const float* x; size_t x_bs; int x_ld; int Tin;
    const float* wt;                    // tile order: ((cb * K + tap) * (Cin / 8) + g) * 256 + lane * 4 + j  =
                                        //   W[cb * 32 + lane % 32][g * 8 + 2 j + lane / 32][tap], zero above Cout
    const float* bias;                  // [Cout]
    const float* alpha;                 // [Cin], Snake only
    float* y; size_t y_bs; int y_ld; int Tout;     // y may be null (accumulate only); res / res2 / acc share its layout
    const float* res; const float* res2;
    float* acc; int acc_mode;
    int Cin, Cout, CoutPad, K, dil, pad, act; float slope;
    int out_stride, out_off;            // output column of position q: q * out_stride + out_off (+ phase, see phase_cout)
    int phase_cout;                     // > 0: a transposed convolution with its output phases stacked as channels: row
                                        //   co' = phase * phase_cout + co is channel co at column q * out_stride + phase + out_off
    int Tq;                             // positions computed: q in [0, Tq)
#ifdef S3_TIMELINE
    long long* timeline;                // tools/conv_probe: cycle stamps, 64 per wave
#endif
};

int launch_conv1d(const ConvArgs& a, int B, hipStream_t st);

// One step of a residual block on 64-channel tensors in one launch (hifigan.py:156-160):
//     y = conv2(snake(conv1(snake(x, alpha1), dilation), alpha2)) + x (+ res2), folded into `acc` per acc_mode.
// The intermediate tensor never leaves the chip.  All tensors [batch][64][L] with row stride ld and batch stride bs.
struct ResStepArgs {
    const float* x;
    const float *w1, *b1, *alpha1;      // conv1: tile order (ConvArgs::wt), bias, Snake alpha of its input; dilation `dil`
    const float *w2, *b2, *alpha2;      // conv2: the same, dilation 1
    float* y;                           // may be null (accumulate only)
    const float* res2;                  // may be null
    float* acc; int acc_mode;
    size_t bs; int ld; int L; int K; int dil;
};
int launch_resblock_step64(const ResStepArgs& a, int B, hipStream_t st);
int launch_stft(const float* s, int L, float* out, int M, int ld, size_t out_bs, int B, hipStream_t st);
int launch_space_to_depth(const float* x, int x_ld, size_t x_bs, int Tin, int Cin, float* y, int y_ld, size_t y_bs, int Tq,
                          int Cpad, int stride, int pad, int B, hipStream_t st);
int launch_reflect_left(float* x, int ld, int rows, hipStream_t st);
int launch_istft(const float* x, int ld, size_t x_bs, int M, float* wav, int L, int B, hipStream_t st);
int launch_f0_head(const float* x, int ld, size_t x_bs, int C, int T, const float* w, const float* bias, float* f0, int B,
                   hipStream_t st);
int launch_source(const float* f0, int T, double* prefix, const float* phase, const float* noise, uint64_t seed, const float* lw,
                  const float* lb, float* s, int B, hipStream_t st);
int launch_overwrite_head(float* s, int L, const float* cache, int n_cache, int B, hipStream_t st);

// C[z](m, n) = sum_k A[z](m, k) * B[z](k, n) in fp32 on the matrix cores, operands addressed by element strides; batch entry
// z sits at z * *_bs of each operand.
struct BGemm {
    const float* A; long long a_bs, sam, sak;
    const float* B; long long b_bs, sbk, sbn;
    float* C; long long c_bs, ldc;
    int M, N, K;
};
int launch_bgemm(const BGemm& g, int batch, hipStream_t st);

// One launch of the small-tile matrix kernel for linear layers and short convolutions (lin_kernel, cfm_decoder.hip):
// y[z][co][n] = out_act(bias[co] + sum_{tap, ci} W[co][ci][tap] x[z][ci][n + tap - pad]) (+ res[z][co][n]) for z < rows;
// x outside [0, L) reads as zero, y must not be x, Cin a multiple of 64.
// With rsum (one-tap layers only) the layer takes LayerNorm(x) over the Cin channels of a frame instead of x: the caller folds
// the norm's gain into the weights (W' = W diag(gamma)), its shift into the bias (b' = b + W beta) and passes rsum[co] =
// sum_ci W'[co][ci]; the kernel multiplies the raw x, gathers each frame's sum and sum of squares from the operands it
// loads anyway, and finishes y = rstd (W' x - mean rsum) + b'.
struct LinArgs {
    const float* x; size_t x_bs; int x_ld;
    const float* wt; const float* bias;      // wt in tile order (ConvArgs::wt)
    float* y; size_t y_bs; int y_ld;
    const float* res;                        // layout of y; may be y itself
    float* yt; size_t yt_bs; int yt_ld, yt_from;   // yt not null: channels >= yt_from go to yt[z][n][co - yt_from] (frame-major)
    int Cin, Cout, CoutPad, K, pad, L;
    const float* rsum; float ln_eps;         // rsum not null: LayerNorm of the input folded into this layer
    // qp not null (a q | k | v projection of 3 x 512 channels, 8 heads x 64): nothing goes to y / yt; the outputs are written
    // as split-bf16 attention operands instead (cfm_split.h: q scaled by q_scale, three planes each), p_ld = their row stride
    unsigned short *qp, *kp, *vp; int p_ld; float q_scale;
    // wsp not null: the weights as three bf16 planes in the operand order of the bf16 matrix instruction (cfm_decoder.hip:
    // split_weights_kernel), and the products taken on split operands (cfm_split.h); needs K * Cin / 16 steps to divide by the
    // launch's reduction split, which launch_lin checks (it falls back to wt otherwise)
    const unsigned short* wsp;
#ifdef S3_LIN_TIMELINE
    long long* timeline;                     // tools/cfm_lin_probe.py on a build with -DS3_LIN_TIMELINE: eight stamps per wave
#endif
};
enum { OUT_NONE = 0, OUT_GELU = 1, OUT_SWISH = 2, OUT_LRELU = 3 };   // OUT_LRELU: slope 0.01
int launch_lin(const LinArgs& a, int out_act, int rows, hipStream_t st);

// W[Cout][Cin][K] (row-major) -> the convolution kernel's tile order (ConvArgs::wt), output channels padded to 32
inline std::vector<float> s3_tile_conv_weights(const float* w, int Cout, int Cin, int K) {
    const int g8 = Cin / 8, nb = (Cout + 31) / 32;
    std::vector<float> t((size_t)nb * K * g8 * 256, 0.f);
    for (int cb = 0; cb < nb; ++cb)
        for (int tap = 0; tap < K; ++tap)
            for (int g = 0; g < g8; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int co = cb * 32 + (lane & 31), ci = g * 8 + 2 * j + (lane >> 5);
                        if (co < Cout)
                            t[((((size_t)cb * K + tap) * g8 + g) * 64 + lane) * 4 + j] = w[((size_t)co * Cin + ci) * K + tap];
                    }
    return t;
}

void s3_set_error(const std::string& msg);
#define S3_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            s3_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define S3_REQUIRE(cond, msg)                                                                \
    do {                                                                                     \
        if (!(cond)) { s3_set_error(std::string(msg) + " [" #cond "]"); return 1; }          \
    } while (0)
