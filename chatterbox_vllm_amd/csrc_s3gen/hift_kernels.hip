// HiFT vocoder kernels for gfx950: fp32 convolutions on the f32-input matrix instruction, and the small kernels around them.
//
// Activations live in HBM as [batch][channel][time] fp32 (time contiguous, a row stride per buffer).  A convolution
//     y[co][q] = bias[co] + sum_{ci, tap} W[co][ci][tap] * act(x[ci][q + tap * dil - pad])
// is a matrix product with M = output channels (A operand = weights), N = time (B operand = activations) and
// K = input channels x taps, on v_mfma_f32_32x32x2_f32: exact fp32 products and sums, 64 FLOP per clock and SIMD.
//  * A workgroup of 4 waves owns 64 channels x 128 positions (or 32 x 256 for narrow layers); a wave owns 32 x 64 as two
//    32 x 32 accumulators.
//  * Input channels go through LDS 16 at a time: [16][positions + halo] with the activation function (Snake, leaky ReLU,
//    ELU) applied on the way in, so every tap reads the same tile at a shifted column and a value's sine is taken once per
//    workgroup, not once per tap.  Rows are 32 banks apart modulo 64, so the two half-waves of a B read (rows k and k+1)
//    never meet in a bank.  The next 16 channels are loaded into registers before the current 16 are multiplied and
//    written to the other LDS buffer after: one barrier per 16 channels.
//  * Weights are read from memory in the instruction's own A layout (tile (32 channels, tap, 8 inputs) = 1 KiB, one
//    16-byte load per lane = 4 k-steps), prefetched one tap ahead; every workgroup of a launch reads the same few MB,
//    which stay in L2.
//  * The epilogue adds bias and up to two residual tensors and either stores, or folds into the running mean of the three
//    residual blocks of a stage.  Outputs may be strided (out_stride, out_off): a transposed convolution runs as one
//    plain convolution per output phase.
#include "hift_common.h"
#include <cstdlib>

namespace {

// periodic Hann window of 16 points (scipy get_window("hann", 16, fftbins=True) in fp32) and the 16th roots of unity
__constant__ float c_hann[16] = {0.f, 0.038060233f, 0.146446615f, 0.308658272f, 0.5f, 0.691341698f, 0.853553414f, 0.961939752f,
                                 1.f, 0.961939752f, 0.853553414f, 0.691341698f, 0.5f, 0.308658272f, 0.146446615f, 0.038060233f};
__constant__ float c_cos16[16] = {1.f, 0.923879504f, 0.707106769f, 0.382683426f, 0.f, -0.382683426f, -0.707106769f, -0.923879504f,
                                  -1.f, -0.923879504f, -0.707106769f, -0.382683426f, 0.f, 0.382683426f, 0.707106769f, 0.923879504f};
__constant__ float c_sin16[16] = {0.f, 0.382683426f, 0.707106769f, 0.923879504f, 1.f, 0.923879504f, 0.707106769f, 0.382683426f,
                                  0.f, -0.382683426f, -0.707106769f, -0.923879504f, -1.f, -0.923879504f, -0.707106769f, -0.382683426f};

// sin(t)^2 for the Snake activation.  The sign of the sine does not matter, so after reducing t by multiples of pi/2
// (three-part constant, exact products for |t| < 8192) only the parity of the multiple selects between the sine and the
// cosine polynomial of the remainder (|r| <= pi/4; coefficients of the classic single-precision kernels, error below
// 1e-7).  Larger arguments, which a trained vocoder does not produce, take the library routine out of line.
__device__ __noinline__ float sin_sq_slow(float t) {
    const float s = sinf(t);
    return s * s;
}
__device__ __forceinline__ float sin_sq(float t) {
    if (__builtin_expect(fabsf(t) > 8192.0f, 0)) return sin_sq_slow(t);
    const float k = rintf(t * 0.636619772f);
    float r = fmaf(k, -1.5703125f, t);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188216e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                          fmaf(-0.5f, z, 1.0f));
    const float v = ((int)k & 1) ? cp : sp;
    return v * v;
}

template <int ACT>
__device__ __forceinline__ float apply_act(float v, float al, float inv, float slope) {
    if constexpr (ACT == ACT_SNAKE) return fmaf(inv, sin_sq(v * al), v);
    else if constexpr (ACT == ACT_LRELU) return v > 0.f ? v : v * slope;
    else if constexpr (ACT == ACT_ELU) return v > 0.f ? v : expf(v) - 1.0f;
    else if constexpr (ACT == ACT_SWISH) return v * (1.0f / (1.0f + expf(-v)));
    else return v;
}

// WCO x WT waves along channels x time; a wave owns 32 channels x (32 WN) positions as WN accumulators.
template <int WCO, int WT, int WN, int ACT>
__global__ __launch_bounds__(256) void conv1d_kernel(ConvArgs a) {
    constexpr int MT = 32 * WCO, WV = 32 * WN, NT = WV * WT, XW = NT + 96, NCOL = (NT + CONV_MAX_HALO + 63) / 64;
    static_assert(WCO * WT == 4 && XW % 64 == 32 && NCOL * 64 <= XW, "tile shape");
    __shared__ float xs[2][CONV_CK][XW];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wco = w % WCO, wt = w / WCO;
    const int q0 = blockIdx.x * NT, co0 = blockIdx.y * MT, b = blockIdx.z;
    const float* __restrict__ xb = a.x + (size_t)b * a.x_bs;
    const int span = NT + (a.K - 1) * a.dil;                 // columns of the tile that the taps touch
    const int tin0 = q0 - a.pad;                             // input position of column 0

    // this lane's columns of a tile row: input position (clamped for the load, the value zeroed when outside)
    int tcl[NCOL];
    bool ok[NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
        const int c = lane + 64 * j, t = tin0 + c;
        ok[j] = c < span && t >= 0 && t < a.Tin;
        tcl[j] = t < 0 ? 0 : (t >= a.Tin ? a.Tin - 1 : t);
    }
    float raw[4][NCOL];
    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* xr = xb + (size_t)(ci0 + w * 4 + r) * a.x_ld;
#pragma unroll
            for (int j = 0; j < NCOL; ++j) raw[r][j] = xr[tcl[j]];
        }
    };
    auto store_chunk = [&](int buf, int ci0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float al = 0.f, inv = 0.f;
            if constexpr (ACT == ACT_SNAKE) { al = a.alpha[ci0 + w * 4 + r]; inv = 1.0f / (al + 1e-9f); }
#pragma unroll
            for (int j = 0; j < NCOL; ++j)
                xs[buf][w * 4 + r][lane + 64 * j] = ok[j] ? apply_act<ACT>(raw[r][j], al, inv, a.slope) : 0.f;
        }
    };

#ifdef S3_TIMELINE
    long long* tl = a.timeline ? a.timeline + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + w) * 64 : nullptr;
    int tli = 0;
#define S3_STAMP() do { if (tl && lane == 0 && tli < 64) tl[tli++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define S3_STAMP() do { } while (0)
#endif
    S3_STAMP();
    static_assert(WN == 1 || WN == 2 || WN == 4, "one, two or four accumulators per wave");
    f32x16 acc0, acc1, acc2, acc3;                            // acc2, acc3: WN == 4 only
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
    const int nchunk = a.Cin / CONV_CK, g8 = a.Cin / 8;
    const int cb = (co0 >> 5) + wco;                          // this wave's block of 32 output channels
    const bool live = cb * 32 < a.CoutPad;
    const f32x4* __restrict__ wbase = reinterpret_cast<const f32x4*>(a.wt) + (size_t)(live ? cb : 0) * a.K * g8 * 64 + lane;

    load_chunk(0);
    f32x4 A0 = wbase[0], A1 = wbase[64];                      // weights of (chunk 0, tap 0); then always one step ahead
    store_chunk(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunk;
        S3_STAMP();
        if (more) load_chunk((ch + 1) * CONV_CK);
        S3_STAMP();
        if (live) {
            const float* xrow = &xs[buf][lane >> 5][wt * WV + (lane & 31)];
            for (int tap = 0; tap < a.K; ++tap) {
                // the step after (ch, tap) is (ch, tap + 1), or (ch + 1, 0); the last step re-reads its own tile
                const bool wrap = tap + 1 == a.K;
                const int nch = wrap ? (more ? ch + 1 : ch) : ch, ntap = wrap ? (more ? 0 : tap) : tap + 1;
                const f32x4* np = wbase + ((size_t)ntap * g8 + nch * 2) * 64;
                const f32x4 N0 = np[0], N1 = np[64];
                const float* xp = xrow + tap * a.dil;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float av = kk < 4 ? A0[kk & 3] : A1[kk & 3];
                    const float b0 = xp[(2 * kk) * XW];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
                    if constexpr (WN >= 2) {
                        const float b1 = xp[(2 * kk) * XW + 32];
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
                    }
                    if constexpr (WN == 4) {
                        const float b2 = xp[(2 * kk) * XW + 64], b3 = xp[(2 * kk) * XW + 96];
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2, acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b3, acc3, 0, 0, 0);
                    }
                }
                A0 = N0;
                A1 = N1;
            }
        }
        S3_STAMP();
        if (more) store_chunk(buf ^ 1, (ch + 1) * CONV_CK);
        S3_STAMP();
        __syncthreads();
    }
    S3_STAMP();
    const size_t ob = (size_t)b * a.y_bs;
    // one output element: bias, residuals, store, running mean of the stage (same operations in both epilogue forms)
    auto finish = [&](float v, int co, size_t o) {
        v = v + a.bias[co];
        if (a.res) v = v + a.res[o];
        if (a.res2) v = a.res2[o] + v;
        if (a.y) a.y[o] = v;
        if (a.acc_mode == ACC_SET) a.acc[o] = v;
        else if (a.acc_mode == ACC_ADD) a.acc[o] = a.acc[o] + v;
        else if (a.acc_mode == ACC_ADD_DIV3) a.acc[o] = (a.acc[o] + v) / 3.0f;
    };
    // accumulator r of a lane: channel 8 (r / 4) + 4 (lane / 32) + r % 4 of the block, position lane % 32 of the sub-tile
    if (a.out_stride == 1 && a.phase_cout == 0 && a.out_off >= 0 && ((a.y_ld | a.out_off) & 3) == 0) {
        // contiguous, 16-byte aligned rows: each 32 x 32 accumulator goes through this wave's 5 KiB of the (now free) tile
        // buffer and comes back as four 16-byte pieces per lane, 8 rows x 128 bytes per store instruction
        float* ep = &xs[0][0][0] + w * (32 * 40);
#pragma unroll
        for (int half = 0; half < WN; ++half) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[(8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * 40 + (lane & 31)] =
                    half == 0 ? acc0[r] : (half == 1 ? acc1[r] : (half == 2 ? acc2[r] : acc3[r]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(&ep[row * 40 + c4]);
                const int co = cb * 32 + row, q = q0 + wt * WV + half * 32 + c4;
                const int t = q + a.out_off;
                if (!live || co >= a.Cout || q >= a.Tq || t >= a.Tout) continue;
                const size_t o = ob + (size_t)co * a.y_ld + t;
                if (q + 3 < a.Tq && t + 3 < a.Tout) {
                    const float bias = a.bias[co];
                    f32x4 v = {v4[0] + bias, v4[1] + bias, v4[2] + bias, v4[3] + bias};
                    if (a.res) { const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + o); v = v + rr; }
                    if (a.res2) { const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res2 + o); v = rr + v; }
                    if (a.y) *reinterpret_cast<f32x4*>(a.y + o) = v;
                    if (a.acc_mode == ACC_SET) *reinterpret_cast<f32x4*>(a.acc + o) = v;
                    else if (a.acc_mode == ACC_ADD) { f32x4 p = *reinterpret_cast<f32x4*>(a.acc + o); *reinterpret_cast<f32x4*>(a.acc + o) = p + v; }
                    else if (a.acc_mode == ACC_ADD_DIV3) {
                        f32x4 p = *reinterpret_cast<f32x4*>(a.acc + o);
                        p = p + v;
                        *reinterpret_cast<f32x4*>(a.acc + o) = f32x4{p[0] / 3.0f, p[1] / 3.0f, p[2] / 3.0f, p[3] / 3.0f};
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (q + e < a.Tq && t + e < a.Tout) finish(v4[e], co, o + e);
                }
            }
        }
    } else if (live) {
        // strided outputs (a transposed convolution's phases) or unaligned rows: one element at a time; a block of 32 rows
        // never straddles two phases (phase_cout is a multiple of 32)
        const int phase = a.phase_cout > 0 ? (cb * 32) / a.phase_cout : 0;
        const int co_base = cb * 32 + 4 * (lane >> 5) - phase * a.phase_cout;
        const int co_lim = a.Cout - phase * a.phase_cout;    // co_base + 8 (r / 4) + r % 4 must stay below
#pragma unroll
        for (int half = 0; half < WN; ++half) {
            const int q = q0 + wt * WV + half * 32 + (lane & 31);
            const long long t = (long long)q * a.out_stride + a.out_off + phase;
            if (q >= a.Tq || t < 0 || t >= a.Tout) continue;
            const unsigned o0 = (unsigned)co_base * (unsigned)a.y_ld + (unsigned)t;   // inside the utterance: below 2^32
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = 8 * (r >> 2) + (r & 3);
                if (co_base + dco >= co_lim) continue;
                finish(half == 0 ? acc0[r] : (half == 1 ? acc1[r] : (half == 2 ? acc2[r] : acc3[r])), co_base + dco, ob + o0 + (unsigned)dco * (unsigned)a.y_ld);
            }
        }
    }
    S3_STAMP();
}

// torch.stft(x, 16, 4, 16, hann, center=True, reflect): one thread per frame, 9 real and 9 imaginary outputs
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ s, int L, float* __restrict__ out, int M, int ld,
                                                   size_t out_bs) {
    const int m = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (m >= M) return;
    const float* sb = s + (size_t)b * L;
    float x[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        int i = 4 * m + n - 8;
        i = i < 0 ? -i : (i >= L ? 2 * (L - 1) - i : i);
        x[n] = sb[i] * c_hann[n];
    }
    float* ob = out + (size_t)b * out_bs + m;
#pragma unroll
    for (int f = 0; f < 9; ++f) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            re = fmaf(x[n], c_cos16[(f * n) & 15], re);
            im = fmaf(x[n], c_sin16[(f * n) & 15], im);
        }
        ob[(size_t)f * ld] = re;
        ob[(size_t)(9 + f) * ld] = -im;
    }
}

// source_downs.i (hifigan.py:356-363) is a strided convolution over the 18 STFT rows (kernel 2 x stride).  Here the stride
// is folded into the channels first: y[c * stride + r][q] = x[c][q * stride + r - pad] (zero outside, zero rows up to Cpad),
// after which the layer is a plain two-tap convolution over stride x 18 channels for the matrix kernel.
__global__ __launch_bounds__(256) void space_to_depth_kernel(const float* __restrict__ x, int x_ld, size_t x_bs, int Tin, int Cin,
                                                             float* __restrict__ y, int y_ld, size_t y_bs, int Tq, int stride,
                                                             int pad) {
    const int q = blockIdx.x * 256 + threadIdx.x, cp = blockIdx.y, b = blockIdx.z;
    if (q >= Tq) return;
    const int c = cp / stride, r = cp - c * stride;
    float v = 0.f;
    if (c < Cin) {
        const int i = q * stride + r - pad;
        if (i >= 0 && i < Tin) v = x[(size_t)b * x_bs + (size_t)c * x_ld + i];
    }
    y[(size_t)b * y_bs + (size_t)cp * y_ld + q] = v;
}

// nn.ReflectionPad1d((1, 0)) after the last upsampling (hifigan.py:421-422): the stage wrote its output one column to the
// right; column 0 is the old column 1, which now sits in column 2
__global__ void reflect_left_kernel(float* __restrict__ x, int ld, int rows) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < rows) x[(size_t)r * ld] = x[(size_t)r * ld + 2];
}

// conv_post output [18][M] -> waveform: magnitude = min(exp(x[0:9]), 100), phase = sin(x[9:18]), inverse real DFT of
// every frame, Hann window, overlap-add over the (up to) four frames that cover a sample, division by the overlap-added
// squared window, 8 samples cut from both ends, clamp (hifigan.py:439-444, 404-410).  One thread per output sample.
__global__ __launch_bounds__(256) void istft_kernel(const float* __restrict__ x, int ld, size_t x_bs, int M,
                                                    float* __restrict__ wav, int L, float limit) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= L) return;
    const int n = i + 8;
    const float* xb = x + (size_t)b * x_bs;
    int m_lo = (n - 15 + 3) >> 2;
    if (m_lo < 0) m_lo = 0;
    int m_hi = n >> 2;
    if (m_hi > M - 1) m_hi = M - 1;
    float y = 0.f, env = 0.f;
    for (int m = m_lo; m <= m_hi; ++m) {
        const int j = n - 4 * m;
        float fr = 0.f;
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            const float mag = fminf(expf(xb[(size_t)f * ld + m]), 100.0f);
            const float ph = sinf(xb[(size_t)(9 + f) * ld + m]);
            float sn, cs;
            sincosf(ph, &sn, &cs);
            const float re = mag * cs, im = mag * sn;
            const float wgt = (f == 0 || f == 8) ? 1.0f : 2.0f;
            fr += wgt * (re * c_cos16[(f * j) & 15]);
            if (f != 0 && f != 8) fr -= wgt * (im * c_sin16[(f * j) & 15]);
        }
        const float wj = c_hann[j];
        y += wj * (fr * (1.0f / 16.0f));
        env += wj * wj;
    }
    y = y / env;
    wav[(size_t)b * L + i] = fminf(fmaxf(y, -limit), limit);
}

// f0 = | w . elu(x[:, t]) + b |  (f0_predictor.py:50-55; the ELU is the last one of the condnet)
__global__ __launch_bounds__(256) void f0_head_kernel(const float* __restrict__ x, int ld, size_t x_bs, int C, int T,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ f0) {
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    const float* xb = x + (size_t)b * x_bs + t;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = xb[(size_t)c * ld];
        acc = fmaf(w[c], v > 0.f ? v : expf(v) - 1.0f, acc);
    }
    f0[(size_t)b * T + t] = fabsf(acc + bias[0]);
}

// per (utterance, harmonic): the running sum of the per-sample phase increments at every frame start, in fp64
__global__ void source_prefix_kernel(const float* __restrict__ f0, int T, double* __restrict__ prefix) {
    const int b = blockIdx.x, h = threadIdx.x;
    if (h >= S3HIFT_HARMONICS) return;
    double run = 0.0;
    for (int m = 0; m < T; ++m) {
        prefix[((size_t)b * S3HIFT_HARMONICS + h) * T + m] = run;
        const float inc = f0[(size_t)b * T + m] * (float)(h + 1) / 24000.0f;
        run += 480.0 * (double)inc;
    }
}

__device__ __forceinline__ uint32_t mix32(uint64_t z) {      // splitmix64 finaliser, upper half
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float unit_open(uint32_t u) { return ((float)(u >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

// SineGen + SourceModuleHnNSF (hifigan.py:201-231, 267-283): one thread per output sample
__global__ __launch_bounds__(256) void source_kernel(const float* __restrict__ f0, int T, const double* __restrict__ prefix,
                                                     const float* __restrict__ phase, const float* __restrict__ noise,
                                                     uint64_t seed, const float* __restrict__ lw, const float* __restrict__ lb,
                                                     float* __restrict__ s) {
    const int L = T * 480;
    const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (n >= L) return;
    const int m = n / 480, j = n - m * 480;
    const float f = f0[(size_t)b * T + m];
    const float uv = f > 10.0f ? 1.0f : 0.0f;
    const float noise_amp = uv * 0.003f + (1.0f - uv) * 0.1f / 3.0f;
    float acc = 0.f;
#pragma unroll
    for (int h = 0; h < S3HIFT_HARMONICS; ++h) {
        const float inc = f * (float)(h + 1) / 24000.0f;
        const double cum = prefix[((size_t)b * S3HIFT_HARMONICS + h) * T + m] + (double)(j + 1) * (double)inc;
        // the reference takes the running sum to fp32 before it takes the fractional part (torch.cumsum on the CPU sums in
        // double and stores fp32; `% 1` then works on that fp32 value), which quantises the phase of long utterances to the
        // fp32 spacing of the sum; the same here, so that the source is the reference's at every length
        const float cs = (float)cum;
        const float frac = cs - floorf(cs);
        const float theta = 6.283185307179586f * frac;
        float ph = 0.f;
        if (h > 0) {
            ph = phase ? phase[b * S3HIFT_HARMONICS + h]
                       : (2.0f * unit_open(mix32(seed ^ (0xA5A5ull << 48) ^ ((uint64_t)b << 8) ^ (uint64_t)h)) - 1.0f) * 3.14159265358979f;
        }
        float z;
        if (noise) {
            z = noise[((size_t)b * S3HIFT_HARMONICS + h) * L + n];
        } else {                                            // Box-Muller on two counter-based uniforms
            const uint64_t ctr = ((uint64_t)b * S3HIFT_HARMONICS + h) * (uint64_t)L + (uint64_t)n;
            const float u1 = unit_open(mix32(seed + 2 * ctr)), u2 = unit_open(mix32(seed + 2 * ctr + 1));
            z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
        }
        const float sine = 0.1f * sinf(theta + ph);
        acc = fmaf(sine * uv + noise_amp * z, lw[h], acc);
    }
    s[(size_t)b * L + n] = tanhf(acc + lb[0]);
}

__global__ void overwrite_head_kernel(float* __restrict__ s, int L, const float* __restrict__ cache, int n_cache) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i < n_cache && i < L) s[(size_t)b * L + i] = cache[(size_t)b * n_cache + i];
}

// One residual-block step for 64 channels: both convolutions of the step in one workgroup, the tensor between them kept in
// LDS.  A workgroup produces NTO = 128 - 2 ((K - 1) / 2 rounded up to a multiple of 2) consecutive outputs (a multiple of
// 4, so rows stay 16-byte aligned): conv1 is evaluated on the 128 positions that conv2's taps reach (its own taps reach
// another (K - 1) dil / 2 to each side in the input tile), Snake(alpha2) of it goes to LDS as [64][128 (+ taps)], zero
// where the position lies outside the tensor (conv2's zero padding), and conv2 reads it at shifted columns.  Both
// convolutions sum in the order of conv1d_kernel (16 channels at a time, taps, pairs of channels), so the result has the
// bits of the two-launch form; it costs 128 / NTO = 3-10 % more matrix work and saves three of five tensor passes over
// HBM and one launch.
__global__ __launch_bounds__(256) void resblock_step64_kernel(ResStepArgs a, int nto, int p2) {
    constexpr int NT = 128, XW = NT + 96, NCOL = 3, TW = 160;      // TW: row stride of the intermediate tile (32 mod 64)
    __shared__ float lds[64 * TW];                                 // 40 KiB: the input tile buffers, then the intermediate tile
    float (*xs)[CONV_CK][XW] = reinterpret_cast<float (*)[CONV_CK][XW]>(lds);   // [2][16][224] = 28 KiB of it
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wco = w & 1, wt = w >> 1;
    const int q0 = blockIdx.x * nto, b = blockIdx.y;
    const float* __restrict__ xb = a.x + (size_t)b * a.bs;
    const int p1 = (a.K - 1) * a.dil / 2;
    const int span = NT + (a.K - 1) * a.dil;
    const int tin0 = q0 - p2 - p1;                                 // input position of tile column 0
    int tcl[NCOL];
    bool ok[NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; ++j) {
        const int c = lane + 64 * j, t = tin0 + c;
        ok[j] = c < span && t >= 0 && t < a.L;
        tcl[j] = t < 0 ? 0 : (t >= a.L ? a.L - 1 : t);
    }
    float raw[4][NCOL];
    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* xr = xb + (size_t)(ci0 + w * 4 + r) * a.ld;
#pragma unroll
            for (int j = 0; j < NCOL; ++j) raw[r][j] = xr[tcl[j]];
        }
    };
    auto store_chunk = [&](int buf, int ci0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float al = a.alpha1[ci0 + w * 4 + r], inv = 1.0f / (al + 1e-9f);
#pragma unroll
            for (int j = 0; j < NCOL; ++j)
                xs[buf][w * 4 + r][lane + 64 * j] = ok[j] ? apply_act<ACT_SNAKE>(raw[r][j], al, inv, 0.f) : 0.f;
        }
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    constexpr int g8 = 8, nchunk = 4;                              // 64 input channels
    const f32x4* __restrict__ w1 = reinterpret_cast<const f32x4*>(a.w1) + (size_t)wco * a.K * g8 * 64 + lane;
    const f32x4* __restrict__ w2 = reinterpret_cast<const f32x4*>(a.w2) + (size_t)wco * a.K * g8 * 64 + lane;

    // ---- conv1 (dilated) over the 128 positions q0 - p2 .. q0 - p2 + 127
    load_chunk(0);
    f32x4 A0 = w1[0], A1 = w1[64];
    store_chunk(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunk;
        if (more) load_chunk((ch + 1) * CONV_CK);
        const float* xrow = &xs[buf][lane >> 5][wt * 64 + (lane & 31)];
        for (int tap = 0; tap < a.K; ++tap) {
            const bool wrap = tap + 1 == a.K;
            // after the last step of conv1 come the first weights of conv2
            const f32x4* np = wrap ? (more ? w1 + (size_t)((ch + 1) * 2) * 64 : w2) : w1 + ((size_t)(tap + 1) * g8 + ch * 2) * 64;
            const f32x4 N0 = np[0], N1 = np[64];
            const float* xp = xrow + tap * a.dil;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float av = kk < 4 ? A0[kk & 3] : A1[kk & 3];
                const float b0 = xp[(2 * kk) * XW], b1 = xp[(2 * kk) * XW + 32];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
            }
            A0 = N0;
            A1 = N1;
        }
        if (more) store_chunk(buf ^ 1, (ch + 1) * CONV_CK);
        __syncthreads();
    }
    // ---- Snake(alpha2) of (conv1 + bias1) -> the intermediate tile; zero outside the tensor and in the tap margin
    float (*xt)[TW] = reinterpret_cast<float (*)[TW]>(lds);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int c = wt * 64 + half * 32 + (lane & 31);
        const int pos = q0 - p2 + c;
        const bool inside = pos >= 0 && pos < a.L;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wco * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            const float al = a.alpha2[co], inv = 1.0f / (al + 1e-9f);
            const float v = (half ? acc1[r] : acc0[r]) + a.b1[co];
            xt[co][c] = inside ? apply_act<ACT_SNAKE>(v, al, inv, 0.f) : 0.f;
        }
    }
    if (tid < 64)
#pragma unroll
        for (int c = NT; c < TW; ++c) xt[tid][c] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    __syncthreads();
    // ---- conv2 (dilation 1): output column j is position q0 + j and reads intermediate columns j .. j + K - 1
    for (int ch = 0; ch < nchunk; ++ch) {
        const float* xrow = &xt[ch * CONV_CK + (lane >> 5)][wt * 64 + (lane & 31)];
        for (int tap = 0; tap < a.K; ++tap) {
            const bool wrap = tap + 1 == a.K, more = ch + 1 < nchunk;
            const f32x4* np = wrap ? (more ? w2 + (size_t)((ch + 1) * 2) * 64 : w2 + ((size_t)tap * g8 + ch * 2) * 64)
                                   : w2 + ((size_t)(tap + 1) * g8 + ch * 2) * 64;
            const f32x4 N0 = np[0], N1 = np[64];
            const float* xp = xrow + tap;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float av = kk < 4 ? A0[kk & 3] : A1[kk & 3];
                const float b0 = xp[(2 * kk) * TW], b1 = xp[(2 * kk) * TW + 32];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
            }
            A0 = N0;
            A1 = N1;
        }
    }
    __syncthreads();                                               // the intermediate tile is free: reuse it for the transpose
    // ---- epilogue as in conv1d_kernel: 32 x 32 accumulators through LDS, 16-byte pieces per lane
    const size_t ob = (size_t)b * a.bs;
    float* ep = lds + w * (32 * 40);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ep[(8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * 40 + (lane & 31)] = half ? acc1[r] : acc0[r];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 8 + (lane >> 3), c4 = (lane & 7) * 4;
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(&ep[row * 40 + c4]);
            const int co = wco * 32 + row, j = wt * 64 + half * 32 + c4, q = q0 + j;
            if (j >= nto || q >= a.L) continue;                    // nto and q0 are multiples of 4: pieces are whole or empty ...
            const size_t o = ob + (size_t)co * a.ld + q;
            const float bias = a.b2[co];
            if (q + 3 < a.L) {
                f32x4 v = {v4[0] + bias, v4[1] + bias, v4[2] + bias, v4[3] + bias};
                v = v + *reinterpret_cast<const f32x4*>(a.x + o);
                if (a.res2) { const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res2 + o); v = rr + v; }
                if (a.y) *reinterpret_cast<f32x4*>(a.y + o) = v;
                if (a.acc_mode == ACC_SET) *reinterpret_cast<f32x4*>(a.acc + o) = v;
                else if (a.acc_mode == ACC_ADD) { f32x4 p = *reinterpret_cast<f32x4*>(a.acc + o); *reinterpret_cast<f32x4*>(a.acc + o) = p + v; }
                else if (a.acc_mode == ACC_ADD_DIV3) {
                    f32x4 p = *reinterpret_cast<f32x4*>(a.acc + o);
                    p = p + v;
                    *reinterpret_cast<f32x4*>(a.acc + o) = f32x4{p[0] / 3.0f, p[1] / 3.0f, p[2] / 3.0f, p[3] / 3.0f};
                }
            } else {                                               // ... except at the tensor's end
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (q + e >= a.L) continue;
                    float v = v4[e] + bias;
                    v = v + a.x[o + e];
                    if (a.res2) v = a.res2[o + e] + v;
                    if (a.y) a.y[o + e] = v;
                    if (a.acc_mode == ACC_SET) a.acc[o + e] = v;
                    else if (a.acc_mode == ACC_ADD) a.acc[o + e] = a.acc[o + e] + v;
                    else if (a.acc_mode == ACC_ADD_DIV3) a.acc[o + e] = (a.acc[o + e] + v) / 3.0f;
                }
            }
        }
    }
}

}  // namespace

int launch_conv1d(const ConvArgs& a, int B, hipStream_t st) {
    S3_REQUIRE(a.Cin % CONV_CK == 0 && a.CoutPad % 32 == 0 && a.K >= 1, "conv1d: channel counts");
    S3_REQUIRE((a.K - 1) * a.dil <= CONV_MAX_HALO, "conv1d: kernel span above the LDS tile's halo");
    S3_REQUIRE(a.Tq >= 1 && B >= 1, "conv1d: empty launch");
    S3_REQUIRE((long long)a.Cout * a.y_ld < (1LL << 32), "conv1d: utterance above 2^32 elements");
    S3_REQUIRE(a.phase_cout % 32 == 0, "conv1d: stacked phases must be whole blocks of 32 channels");
    // tile shapes: 64 channels x 128 positions (5 workgroups per CU); 64 x 256 with four accumulators per wave (3 per CU,
    // half the weight loads and barriers per matrix instruction: 3 % faster at 128 and 256 channels when the launch still
    // has 1.5 tiles per CU, slower at 64 channels); 32 x 256 for the 18 output rows of conv_post
    static const int force = getenv("S3HIFT_TILE") ? atoi(getenv("S3HIFT_TILE")) : -1;
    const long long big_tiles = (long long)((a.Tq + 255) / 256) * ((a.CoutPad + 63) / 64) * B;
    int shape = a.CoutPad < 64 ? 2 : ((a.CoutPad >= 128 && big_tiles >= 384) ? 1 : 0);
    // short launches (one utterance): 64 x 64 tiles, so that the launch has more workgroups than the chip has CUs
    if (shape == 0 && big_tiles * 2 < 384) shape = 3;
    if (force >= 0 && a.CoutPad >= 64) shape = force;
    const dim3 grid = shape == 0 ? dim3((a.Tq + 127) / 128, (a.CoutPad + 63) / 64, B)
                    : shape == 1 ? dim3((a.Tq + 255) / 256, (a.CoutPad + 63) / 64, B)
                    : shape == 3 ? dim3((a.Tq + 63) / 64, (a.CoutPad + 63) / 64, B) : dim3((a.Tq + 255) / 256, 1, B);
#define S3_CONV_LAUNCH(ACT)                                                                                   \
    do {                                                                                                      \
        if (shape == 0) hipLaunchKernelGGL((conv1d_kernel<2, 2, 2, ACT>), grid, dim3(256), 0, st, a);         \
        else if (shape == 1) hipLaunchKernelGGL((conv1d_kernel<2, 2, 4, ACT>), grid, dim3(256), 0, st, a);    \
        else if (shape == 3) hipLaunchKernelGGL((conv1d_kernel<2, 2, 1, ACT>), grid, dim3(256), 0, st, a);    \
        else hipLaunchKernelGGL((conv1d_kernel<1, 4, 2, ACT>), grid, dim3(256), 0, st, a);                    \
    } while (0)
    switch (a.act) {
        case ACT_NONE: S3_CONV_LAUNCH(ACT_NONE); break;
        case ACT_SNAKE: S3_CONV_LAUNCH(ACT_SNAKE); break;
        case ACT_LRELU: S3_CONV_LAUNCH(ACT_LRELU); break;
        case ACT_ELU: S3_CONV_LAUNCH(ACT_ELU); break;
        case ACT_SWISH: S3_CONV_LAUNCH(ACT_SWISH); break;
        default: S3_REQUIRE(false, "conv1d: unknown activation");
    }
#undef S3_CONV_LAUNCH
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_stft(const float* s, int L, float* out, int M, int ld, size_t out_bs, int B, hipStream_t st) {
    hipLaunchKernelGGL(stft_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, s, L, out, M, ld, out_bs);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_space_to_depth(const float* x, int x_ld, size_t x_bs, int Tin, int Cin, float* y, int y_ld, size_t y_bs, int Tq,
                          int Cpad, int stride, int pad, int B, hipStream_t st) {
    hipLaunchKernelGGL(space_to_depth_kernel, dim3((Tq + 255) / 256, Cpad, B), dim3(256), 0, st, x, x_ld, x_bs, Tin, Cin, y, y_ld,
                       y_bs, Tq, stride, pad);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_reflect_left(float* x, int ld, int rows, hipStream_t st) {
    hipLaunchKernelGGL(reflect_left_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, x, ld, rows);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_istft(const float* x, int ld, size_t x_bs, int M, float* wav, int L, int B, hipStream_t st) {
    hipLaunchKernelGGL(istft_kernel, dim3((L + 255) / 256, B), dim3(256), 0, st, x, ld, x_bs, M, wav, L, 0.99f);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_f0_head(const float* x, int ld, size_t x_bs, int C, int T, const float* w, const float* bias, float* f0, int B,
                   hipStream_t st) {
    hipLaunchKernelGGL(f0_head_kernel, dim3((T + 255) / 256, B), dim3(256), 0, st, x, ld, x_bs, C, T, w, bias, f0);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_source(const float* f0, int T, double* prefix, const float* phase, const float* noise, uint64_t seed, const float* lw,
                  const float* lb, float* s, int B, hipStream_t st) {
    hipLaunchKernelGGL(source_prefix_kernel, dim3(B), dim3(64), 0, st, f0, T, prefix);
    hipLaunchKernelGGL(source_kernel, dim3((T * 480 + 255) / 256, B), dim3(256), 0, st, f0, T, prefix, phase, noise, seed, lw,
                       lb, s);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_overwrite_head(float* s, int L, const float* cache, int n_cache, int B, hipStream_t st) {
    if (n_cache <= 0) return 0;
    hipLaunchKernelGGL(overwrite_head_kernel, dim3((n_cache + 255) / 256, B), dim3(256), 0, st, s, L, cache, n_cache);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_resblock_step64(const ResStepArgs& a, int B, hipStream_t st) {
    S3_REQUIRE(a.K >= 1 && (a.K & 1) && (a.K - 1) * a.dil <= CONV_MAX_HALO && a.K - 1 <= 32, "resblock step: kernel span");
    S3_REQUIRE(a.ld % 4 == 0 && a.bs % 4 == 0 && a.L >= 1 && B >= 1, "resblock step: rows must be 16-byte aligned");
    S3_REQUIRE((long long)64 * a.ld < (1LL << 32), "resblock step: utterance above 2^32 elements");
    const int p2 = (a.K - 1) / 2;
    const int nto = (128 - 2 * p2) & ~3;                           // outputs per workgroup, a multiple of 4
    hipLaunchKernelGGL(resblock_step64_kernel, dim3((a.L + nto - 1) / nto, B), dim3(256), 0, st, a, nto, p2);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}
