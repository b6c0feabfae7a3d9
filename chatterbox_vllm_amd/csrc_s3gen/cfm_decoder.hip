// Flow-matching decoder of S3Gen's flow stage (mu -> mel) on gfx950 (C ABI: include/s3cfm_hip.h).
//
// Reference: CausalConditionalCFM.forward / ConditionalCFM.solve_euler (src/chatterbox_vllm/models/s3gen/flow_matching.py:
// 203-228, 82-133) around ConditionalDecoder.forward (decoder.py:233-317).  Every Euler step runs the estimator on two rows
// (conditioned, unconditioned) of the same x.  Activations are channel-major, [row][channel][frame] fp32, the layout of the
// vocoder's tensors: the causal convolutions and every Linear go through lin_kernel below (weights as the A operand, frames
// as N, a Linear being a one-tap convolution), a matrix kernel shaped for the few output tiles of one utterance; residual
// adds and the GELU of the feed-forward ride in its epilogue.  LayerNorm over the 256 channels of a frame, with the Mish and the
// time-embedding add that follow it in a resnet block, is one small kernel.  Attention (8 heads x 64) is one launch with an
// online softmax (attn_kernel).
#include "hift_common.h"
#include "cfm_split.h"
#include "../../include/s3cfm_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <map>
#include <vector>

namespace {

constexpr int kMel = S3CFM_MEL, kIn = 4 * S3CFM_MEL, kC = 256, kTime = 1024, kHeads = 8, kInner = 512, kFf = 1024;
constexpr int kBlocks = 4, kMid = 12, kStages = kMid + 2, kRows = 2, kMaxUtt = 8;
// Row strides are odd multiples of 256 bytes: a tile's rows (64 channels of K, 32 frames of V) then spread over the L2
// channels instead of landing on one or two of them, which strides of 2 or 4 KiB would do.
constexpr int kVtLd = kInner + 64;
constexpr float kQScale = 0.125f * 1.44269504088896340736f;     // 1 / sqrt(64) and log2(e): softmax in base 2

// ---- kernels -----------------------------------------------------------------------------------------------------

// Mish, x tanh(softplus(x)), as one exponential and one division: with n = e^x, tanh(ln(1 + n)) = n (n + 2) / (n (n + 2) + 2);
// no cancellation at either end, and like torch's softplus the identity above 20 (where the ratio is 1 to fp32 anyway)
__device__ __forceinline__ float mish(float x) {
    if (x > 20.0f) return x;
    const float n = expf(x), t = n * (n + 2.0f);
    return x * (t / (t + 2.0f));
}

// GELU, x Phi(x), with Phi(-|x|) = erfc(|x| / sqrt 2) / 2 from the rational approximation 7.1.26 of Abramowitz & Stegun
// (absolute error of erfc below 1.5e-7, so of the result below 0.75e-7 |x|; no cancellation on the negative side): a dozen
// instructions without a branch where the library erff takes both of its branches in most waves (a quarter of the
// feed-forward layer's epilogue)
__device__ __forceinline__ float gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float tail = 0.5f * (p * t) * __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);    // Phi(-|x|)
    return x * (x < 0.f ? tail : 1.0f - tail);
}

enum { LN_PLAIN = 0, LN_MISH = 1, LN_MISH_ADD = 2 };
// LayerNorm over the 256 channels of every frame of a [row][256][T] tensor (eps 1e-5): 16 frames x 16 channel groups per
// workgroup (64-byte pieces of a channel row per load), mean, then the centred sum of squares.  LN_MISH applies Mish to the
// result, LN_MISH_ADD then adds add[c] (the time-embedding projection of a resnet block, matcha/decoder.py:58).  In place is
// allowed.
constexpr int kLnFrames = 16;
template <int MODE>
__global__ __launch_bounds__(256) void ln256_kernel(const float* x, float* y, int ld, size_t bs, int T, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ add) {
    constexpr int G = 256 / kLnFrames, PER = kC / G;            // channel groups, channels per thread
    __shared__ float red[G][kLnFrames + 1];
    const int tx = threadIdx.x % kLnFrames, cy = threadIdx.x / kLnFrames;
    const int t = blockIdx.x * kLnFrames + tx;
    const bool ok = t < T;
    const float* xb = x + blockIdx.y * bs;
    float* yb = y + blockIdx.y * bs;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = ok ? xb[(size_t)(cy + G * i) * ld + t] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) s += v[i];
    red[cy][tx] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) mean += red[i][tx];
    mean *= 1.0f / kC;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    red[cy][tx] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) var += red[i][tx];
    const float rstd = 1.0f / sqrtf(var * (1.0f / kC) + 1e-5f);
    if (ok)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = cy + G * i;
            float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
            if constexpr (MODE != LN_PLAIN) o = mish(o);
            if constexpr (MODE == LN_MISH_ADD) o = o + add[c];
            yb[(size_t)c * ld + t] = o;
        }
}

// SinusoidalPosEmb(320) (matcha/decoder.py:20-29): emb[i] = sin(1000 t f_i), emb[160 + i] = cos(1000 t f_i)
__global__ void sin_emb_kernel(float t, const float* __restrict__ freq, float* __restrict__ emb) {
    const int i = threadIdx.x;
    if (i >= kIn / 2) return;
    const float a = (1000.0f * t) * freq[i];
    emb[i] = sinf(a);
    emb[kIn / 2 + i] = cosf(a);
}

enum { GV_NONE = 0, GV_SILU = 1, GV_MISH = 2 };
// y[o] = out_act(b[o] + sum_k W[o][k] in_act(x[k])), one wave per output, n_in a multiple of 64
template <int IN_ACT, int OUT_ACT>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ x,
                                                   float* __restrict__ y, int n_in, int n_out) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= n_out) return;
    const float* w = W + (size_t)o * n_in;
    float acc = 0.f;
    for (int k = lane; k < n_in; k += 64) {
        float xv = x[k];
        if constexpr (IN_ACT == GV_MISH) xv = mish(xv);
        acc = fmaf(w[k], xv, acc);
    }
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        float v = acc + b[o];
        if constexpr (OUT_ACT == GV_SILU) v = v * (1.0f / (1.0f + expf(-v)));
        y[o] = v;
    }
}

// Operand loads of the two matrix kernels below go through buffer instructions: a wave-uniform descriptor and byte offset in
// scalar registers plus a 32-bit byte offset per lane, so that no vector arithmetic is spent on 64-bit addresses (with flat
// global loads the address arithmetic was a third of the loop's vector instructions).  The wave index is taken through
// readfirstlane so that the compiler knows what depends on it is uniform.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ u32x4 buf_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// Self-attention of one transformer block, 8 heads x 64, all frames attend to all frames (the diffusers default attention
// processor: softmax(q k^T / 8) v).  One workgroup per (32 queries, head, row); its four waves take every fourth 32-key tile
// and run an online softmax each, and their partial (max, sum, output) triples are merged through LDS in wave order.
// Scores are computed transposed, S^T[key][query] = sum_d K[d][key] Q[d][query], so that a lane owns one query column: the
// softmax statistics are per lane (one exchange between the two halves of the wave), the probabilities sit in the accumulator
// layout that the second product wants as its B operand, and O^T[d][query] += V[d][key] P^T[key][query] takes V frame-major
// ([frame][512], written that way by the q/k/v projection) so that its A operand is 32 consecutive floats per key.
// K and V tiles come straight from global memory (L2 hits after the first pass, see the workgroup order); Q lives in LDS, shared
// by the workgroup's waves (or in registers, the earlier form kept for comparison).
struct AttnArgs {
    const float* q; const float* k; size_t qk_bs; int ld;     // [row][512][ld] channel-major, head h at channels 64 h ..
    const float* vt; size_t vt_bs;                            // [row][L][kVtLd] frame-major
    float* o; size_t o_bs;                                    // [row][512][ld]
    int L;                                                    // queries per row
    int rows;                                                 // rows in the launch (two per utterance: with and without conditioning)
    int Lk[kMaxUtt];                                          // keys of rows 2 u and 2 u + 1 (an utterance shorter than L: its
                                                              //   frames past Lk[u] are padding that no query may see)
};
// QT: 32-query tiles per wave (1 to 3).  With more tiles every K and V value a wave loads feeds more matrix work and the
// wave has more independent accumulators in flight (QT = 2 runs a tile 25 % faster than QT = 1); the launcher picks the QT
// whose workgroup count packs best into the CUs (launch_attn).
// QLDS: the workgroup's queries (shared by its four waves) in LDS instead of 32 registers per tile in every wave: 132 / 190 / 249
// registers for 1 / 2 / 3 tiles instead of 210 / 312 / 457, i.e. three / two / two workgroups per CU instead of two / one / one,
// so that one wave's softmax runs under another's matrix instructions (2 000 frames 166 -> 150 us, 4 000 frames 616 -> 534 =
// 123 TFLOP/s; 1 000 frames 46.9 -> 42.9 with one tile per wave).
template <int QT, bool QLDS = false>
__global__ __launch_bounds__(256, QLDS ? 2 : 1) void attn_kernel(
        // leading plain arguments travel in preloaded SGPRs (see lin_kernel)
        const float* lead_q, const float* lead_k, const float* lead_vt, size_t lead_qk_bs, size_t lead_vt_bs, int lead_ld, int lead_L,
        int lead_rows, AttnArgs a_) {
    AttnArgs a = a_;
    a.q = lead_q; a.k = lead_k; a.vt = lead_vt; a.qk_bs = lead_qk_bs; a.vt_bs = lead_vt_bs; a.ld = lead_ld; a.L = lead_L; a.rows = lead_rows;
    __shared__ float so[4 * 32 * 64];
    __shared__ float sm[4][64], sl[4][64];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, col = lane & 31;
    // workgroup i runs on XCD i % 8: give each XCD its own share of the (row, head) pairs, so that the K and V it streams
    // (0.5 KiB per frame and pair) stay in its own L2
    const int npairs = a.rows * kHeads, p = blockIdx.x % npairs, qt = blockIdx.x / npairs;
    const int pair = (p & 7) * a.rows + (p >> 3);
    const int b = pair >> 3, h = pair & 7;
    const int Lk = a_.Lk[b >> 1];                           // indexed in the argument block itself (a scalar load), not in the copy
    if (qt * QT * 32 >= Lk) {
        // queries that are all padding: zeros keep what follows finite without the work
        for (int i = tid; i < 64 * QT * 32; i += 256) {
            const int qn = qt * QT * 32 + i % (QT * 32);
            if (qn < a.L) a.o[b * a.o_bs + (size_t)(h * 64 + i / (QT * 32)) * a.ld + qn] = 0.f;
        }
        return;
    }
    __shared__ float qs[QLDS ? QT * 64 * 32 : 1];              // [tile][dim][query]
    float Q[QLDS ? 1 : QT][QLDS ? 1 : 32];
    if constexpr (QLDS) {
        for (int i = tid; i < QT * 64 * 32; i += 256) {
            const int u = i >> 11, d = (i >> 5) & 63, qn = (qt * QT + u) * 32 + (i & 31);
            qs[i] = a.q[b * a.qk_bs + (size_t)(h * 64 + d) * a.ld + (qn < a.L ? qn : a.L - 1)] * kQScale;
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const int qn = (qt * QT + u) * 32 + col;
            const float* qp = a.q + b * a.qk_bs + (size_t)(h * 64 + half) * a.ld + (qn < a.L ? qn : a.L - 1);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) Q[u][kk] = qp[(size_t)(2 * kk) * a.ld] * kQScale;
        }
    }
    auto qval = [&](int u, int kk) -> float {
        if constexpr (QLDS) return qs[(u * 64 + 2 * kk + half) * 32 + col];
        else return Q[u][kk];
    };
    f32x16 o0[QT], o1[QT];
    float m[QT], l[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[u][i] = 0.f; o1[u][i] = 0.f; }
        m[u] = -INFINITY;
        l[u] = 0.f;
    }
    // K of this (row, head): 64 channel rows of ld frames; V: ld frames of kVtLd floats.  Reads past L stay inside the
    // row stride, which the workspace zero-fills at set-up (the values are masked or multiplied by zero, but must be finite).
    const __amdgpu_buffer_rsrc_t kr_ = make_rsrc(a.k + b * a.qk_bs + (size_t)(h * 64) * a.ld, 64u * a.ld * 4u);
    const __amdgpu_buffer_rsrc_t vr_ = make_rsrc(a.vt + b * a.vt_bs + h * 64, (unsigned)a.ld * kVtLd * 4u);
    const unsigned klane = (unsigned)(half * a.ld + col) * 4u, vlane = (unsigned)(4 * half * kVtLd + col) * 4u;
    const unsigned krow2 = (unsigned)a.ld * 8u;                // bytes between channels 2 kk and 2 kk + 2
    const int ntiles = (Lk + 31) / 32;
    // A tile's 32 K loads are issued a whole tile ahead (right after the score products that free their registers), its 32 V
    // loads before the score products; the scheduling barriers keep the compiler from sinking each load next to its use,
    // which would leave one load in flight at a time.
    float kr[32], v0[16], v1[16];
    auto load_k = [&](int t) {
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) kr[kk] = buf_load(kr_, klane, kk * krow2 + (unsigned)t * 128u);
    };
    if (w < ntiles) load_k(w);
    for (int t = w; t < ntiles; t += 4) {
        const int key0 = t * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = (unsigned)(key0 + 8 * (r >> 2) + (r & 3)) * (kVtLd * 4u);
            v0[r] = buf_load(vr_, vlane, row);
            v1[r] = buf_load(vr_, vlane + 128u, row);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[u][i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk)
#pragma unroll
            for (int u = 0; u < QT; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[kk], qval(u, kk), s[u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_k(t + 4 < ntiles ? t + 4 : t);                    // the last tile loads itself again: no branch to schedule around
        __builtin_amdgcn_sched_barrier(0);
        // accumulator r of a lane: key key0 + 8 (r / 4) + 4 (lane / 32) + r % 4, query lane % 32; scores are in units of
        // ln 2 (the factor sits in Q), so the exponentials below are single v_exp_f32 instructions
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            if (key0 + 32 > Lk) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + 8 * (r >> 2) + 4 * half + (r & 3) >= Lk) s[u][r] = -INFINITY;
            }
            float mx = s[u][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m[u], mx);               // finite: every tile holds a key below L
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[u][r] = __builtin_amdgcn_exp2f(s[u][r] - m_new); ps += s[u][r]; }
            ps += __shfl_xor(ps, 32);
            if (__any(m_new != m[u])) {                        // the running maximum rarely moves after the first tiles
                const float alpha = __builtin_amdgcn_exp2f(m[u] - m_new);
                l[u] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[u][i] *= alpha; o1[u][i] *= alpha; }
                m[u] = m_new;
            }
            l[u] += ps;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                o0[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[r], s[u][r], o0[u], 0, 0, 0);
                o1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[r], s[u][r], o1[u], 0, 0, 0);
            }
    }
    // merge the four waves' partial results, one query tile at a time through the same 32 KiB
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        if (u > 0) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            so[(w * 32 + r) * 64 + lane] = o0[u][r];
            so[(w * 32 + 16 + r) * 64 + lane] = o1[u][r];
        }
        sm[w][lane] = m[u];
        sl[w][lane] = l[u];
        __syncthreads();
        const float ms = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
        float f[4], lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = __builtin_amdgcn_exp2f(sm[i][lane] - ms); lsum += f[i] * sl[i][lane]; }   // a wave without tiles: 2^-inf = 0
        const float inv = 1.0f / lsum;
        const int qn = (qt * QT + u) * 32 + col;
        if (qn < a.L) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = w * 8 + i;                       // this wave finishes a quarter of the 64 output channels
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) v += so[(j * 32 + r) * 64 + lane] * f[j];
                const int d = (r >> 4) * 32 + 8 * ((r & 15) >> 2) + 4 * half + (r & 3);
                a.o[b * a.o_bs + (size_t)(h * 64 + d) * a.ld + qn] = v * inv;
            }
        }
    }
}

// Linear layers and causal convolutions of the estimator: y[z][co][n] = out_act(bias[co] + sum_{tap, ci} W[co][ci][tap] *
// x[z][ci][n + tap - pad]) (+ res[z][co][n]); x outside [0, L) reads as zero.  The utterances here are short (N of a few
// hundred to a few thousand frames against 256 .. 1536 output channels), so a launch has few output tiles and the kernel
// is built for many small workgroups rather than deep tiles: a wave owns 64 channels x 32 frames (two 32 x 32
// accumulators sharing their B operand), takes its weights in tile order straight from L2 (one 16-byte load per lane per
// 8 input channels) and its x operand straight from the channel-major tensor (lanes 0-31 read 32 consecutive frames of
// one channel, lanes 32-63 of the next: two 128-byte segments per load), PF steps ahead in registers; nothing is staged
// in LDS.  KSPLIT = 2, 4 or 8: the waves of a workgroup split the reduction over (tap, ci) in contiguous parts and their
// partial sums meet in LDS in wave order (8 waves when the launch has under two workgroups per CU; 2 parts x 2 frame tiles
// when it has plenty); KSPLIT = 1: four waves take four neighbouring 32-frame tiles instead.
// SPLIT: the products on split-bf16 operands (cfm_split.h): a step covers 16 input channels instead of 8, the weights come
// as three bf16 planes (LinArgs::wsp), the lane's eight x values are split in registers, six matrix instructions per block of
// 32 channels and step instead of eight twice as long ones; same accumulator layout, so the epilogue is the same.
// DEEP: operands of four steps in flight (two in the split form) instead of two (one).  The deep form wins while a launch
// has few workgroups per CU (under about 1 200 columns: nothing else hides the first loads' latency), the shallow one
// above that (fewer registers, more waves per SIMD): 68.7 -> 65.6 ms per fp32 solve at 1 000 frames, 37.2 <- 38.7 at 500.
// WLDS (split form without a split of the reduction, i.e. the large launches): the four waves of a workgroup need the
// same weight planes, so the workgroup fetches them once, two steps (12 KiB) at a time, and the waves read them from LDS.
// In the split form a step's operands are 8 KiB per wave for twelve 32-cycle matrix instructions, and sixteen waves per
// CU then ask the vector memory path for more bytes per cycle (64 B) than it returns; shared, the weights are 1.5 KiB
// per wave and step there and 6 KiB from LDS (128 B per cycle).
#ifndef S3_LIN_OCC_WIDE
#define S3_LIN_OCC_WIDE 1
#endif
#ifndef S3_LIN_OCC_WIDE_SPLIT
#define S3_LIN_OCC_WIDE_SPLIT 1
#endif
#define S3_LIN_OCC(KSPLIT, DEEP, SPLIT) ((KSPLIT) == 1 && !(DEEP) ? ((SPLIT) ? S3_LIN_OCC_WIDE_SPLIT : S3_LIN_OCC_WIDE) : 1)
template <int KSPLIT, int MB, int OUT_ACT, bool LN_IN, bool SPLIT = false, bool DEEP = true, bool WLDS = false>
__global__ __launch_bounds__(KSPLIT == 8 ? 512 : 256, S3_LIN_OCC(KSPLIT, DEEP, SPLIT)) void lin_kernel(
        // Leading plain arguments reach the waves in SGPRs set up at wave launch (the library is built with kernel-argument
        // preloading, which takes no struct): the fields of LinArgs that the first addresses are made of, passed a second time.
        const float* lead_x, const float* lead_wt, size_t lead_x_bs, int lead_x_ld, int lead_Cin, int lead_CoutPad, int lead_K,
        int lead_L, int lead_pad, int lead_Cout, LinArgs a_) {
    LinArgs a = a_;
    a.x = lead_x; a.wt = lead_wt; a.x_bs = lead_x_bs; a.x_ld = lead_x_ld; a.Cin = lead_Cin; a.CoutPad = lead_CoutPad; a.K = lead_K;
    a.L = lead_L; a.pad = lead_pad; a.Cout = lead_Cout;
#ifdef S3_LIN_TIMELINE
    long long tl[8];
    tl[0] = (long long)__builtin_readcyclecounter();
    tl[6] = (long long)wall_clock64();
#define S3_LIN_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); tl[i] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define S3_LIN_STAMP(i) do { } while (0)
#endif
    constexpr int PF = SPLIT ? (DEEP ? 2 : 1) : (DEEP ? 4 : 2);     // steps of operands in flight (a split step holds twice the channels)
    constexpr int RW = 16 * MB;                                // accumulator rows per lane: MB blocks of 32 channels per wave
    static_assert(MB == 2 || (MB == 1 && KSPLIT <= 4), "one block per wave needs at least four rows per wave to finish");
    static_assert(!WLDS || (SPLIT && KSPLIT == 1 && MB == 2 && !DEEP), "weights through LDS: the split form's whole-reduction launches");
    __shared__ u32x4 wls[WLDS ? 2 * 768 : 1];                  // two stages of two steps x two blocks x three planes x 1 KiB
    __shared__ float part[KSPLIT > 1 ? (KSPLIT == 8 ? 8 : 4) * RW * 64 : 1];
    __shared__ float lnst[LN_IN && KSPLIT > 1 ? (KSPLIT == 8 ? 8 : 4) * 2 * 32 : 1];   // per wave and frame: sum, sum of squares
    constexpr int NTW = KSPLIT == 8 ? 1 : 4 / KSPLIT;          // waves of a workgroup along the frames
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
    const int kpart = w % KSPLIT, ntile = w / KSPLIT;          // which part of the reduction, which 32-frame tile
    const int n = (blockIdx.x * NTW + ntile) * 32 + (lane & 31);
    const int cb0 = blockIdx.y * MB, z = blockIdx.z;
    constexpr int CH = SPLIT ? 16 : 8;                         // input channels per step
    constexpr unsigned WSTEP = SPLIT ? 3072u : 1024u;          // weight bytes per step and block of 32 channels
    const int g8 = a.Cin / CH, steps = a.K * g8;
    const int ns = steps / KSPLIT, s0 = kpart * ns;
    const bool live1 = MB == 2 && (cb0 + 1) * 32 < a.CoutPad;
    const __amdgpu_buffer_rsrc_t wr_ = make_rsrc(SPLIT ? static_cast<const void*>(a_.wsp) : static_cast<const void*>(a.wt),
                                                 (unsigned)(a.CoutPad / 32) * steps * WSTEP);
    const unsigned w0s = (unsigned)cb0 * steps * WSTEP;       // this workgroup's first block of 32 channels
    const __amdgpu_buffer_rsrc_t xr_ = make_rsrc(a.x + z * a.x_bs, (unsigned)a.Cin * a.x_ld * 4u);
    const unsigned w1d = live1 ? (unsigned)steps * WSTEP : 0u;
    // fp32 form: a lane's four x values of a step are channels 8 g + 2 j + half; split form: its eight are 16 g + 8 half + j
    const unsigned wlane = (unsigned)lane * 16u, xlane = (unsigned)((SPLIT ? 8 : 1) * half * a.x_ld + n) * 4u;
    const unsigned xrow2 = (unsigned)a.x_ld * (SPLIT ? 4u : 8u);  // bytes between two consecutive x values of a lane

    // The rows this wave will finish (NG groups of four accumulator rows), their biases and residual values.  Where the
    // waves split the reduction a wave finishes 4 to 16 rows and asks for these values first, so that the round trip runs
    // under the main loop instead of after it.  A wave of the whole-reduction form finishes all 32 rows: 64 to 96 values
    // held through the loop would cost it a wave per SIMD, so it asks after the loop, when the operand registers are free
    // (those launches are large: other waves cover the round trip).  A layer with a folded LayerNorm has no residual.
    constexpr int NG = RW / KSPLIT / 4;
    constexpr bool EPI_EARLY = KSPLIT > 1;
    auto rbase = [&](int gi) { return kpart * (RW / KSPLIT) + 4 * gi; };
    float pbias[NG][4], pres[LN_IN ? 1 : NG][4], prsum[LN_IN ? NG : 1][4];
    const bool has_res = !LN_IN && a.res != nullptr;
    const float* resp = (has_res ? a.res : a.bias) + (has_res ? z * a.y_bs + (size_t)(n < a.L ? n : a.L - 1) : 0);
    const size_t res_ld = has_res ? a.y_ld : 0;
    auto load_epilogue_operands = [&](int gi) {
        const int cb = cb0 + (rbase(gi) >> 4), co0 = cb * 32 + 8 * ((rbase(gi) & 15) >> 2) + 4 * half;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // clamped addresses, no select on the loaded value (it would make the wave wait here); rows past Cout and
            // the frame-major rows are never stored with these values
            const int co = co0 + e < a.Cout ? co0 + e : a.Cout - 1;
            pbias[gi][e] = a.bias[co];
            if constexpr (LN_IN) prsum[gi][e] = a.rsum[co];
            else pres[gi][e] = resp[(has_res && !(a.yt && co >= a.yt_from) ? co : 0) * res_ld];
        }
    };
    if constexpr (EPI_EARLY) {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) load_epilogue_operands(gi);
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NB = SPLIT ? 8 : 4;                          // x values of a lane per step
    f32x4 A0[SPLIT ? 1 : PF], A1[SPLIT ? 1 : PF];
    u32x4 S0[SPLIT ? PF : 1][3], S1[SPLIT ? PF : 1][3];        // split form: three planes per block of 32 channels
    float B[PF][NB];
    bool okb[PF];
    // Loads are unconditional (a step past the end re-reads the last one, a frame outside [0, L) reads frame 0 of the row) and the
    // zeroing happens when the value is used: a select right after the load would make the wave wait for it at once.
    // Steps are requested in order, each once: a running (step, tap, channel group) instead of a division per step.  Requests
    // past the wave's last step repeat it and are marked unreal.
    int cu_sg = s0, cu_tap = s0 / g8, cu_g = s0 - cu_tap * g8, cu_left = ns;
    auto next_step = [&](int& sg, int& tap, int& g, bool& real) {
        sg = cu_sg; tap = cu_tap; g = cu_g; real = cu_left > 0;
        if (cu_left > 1) {
            ++cu_sg;
            if (++cu_g == g8) { cu_g = 0; ++cu_tap; }
        }
        --cu_left;
    };
    auto load = [&](int slot) {
        int sg, tap, g;
        bool real;
        next_step(sg, tap, g, real);
        if constexpr (SPLIT) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                S0[slot][pl] = buf_load4u(wr_, wlane, w0s + (unsigned)sg * WSTEP + pl * 1024u);
                if constexpr (MB == 2) S1[slot][pl] = buf_load4u(wr_, wlane, w0s + (unsigned)sg * WSTEP + pl * 1024u + w1d);
            }
        } else {
            A0[slot] = buf_load4(wr_, wlane, w0s + (unsigned)sg * 1024u);
            if constexpr (MB == 2) A1[slot] = buf_load4(wr_, wlane, w0s + (unsigned)sg * 1024u + w1d);
        }
        const int col = n + tap - a.pad;
        const bool ok = col >= 0 && col < a.L;
        okb[slot] = ok && real;
        const unsigned off = ok ? xlane + (unsigned)((tap - a.pad) * 4) : (unsigned)((SPLIT ? 8 : 1) * half * a.x_ld) * 4u;
#pragma unroll
        for (int j = 0; j < NB; ++j) B[slot][j] = buf_load(xr_, off, (unsigned)(g * (SPLIT ? 16 : 4) + j) * xrow2);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float ln_s1 = 0.f, ln_s2 = 0.f;                            // LN_IN: this lane's share of its frame's sum and sum of squares
    f32x2 ln_p1 = {0.f, 0.f}, ln_p2 = {0.f, 0.f};              // the split form keeps them in pairs (packed adds and fmas)
    // one step of the split form on a lane's eight x values and the two blocks' planes
    auto split_step = [&](const float* bx, bool ok, const u32x4* s0p, const u32x4* s1p) {
        u32x4 P[3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // a layer with a folded LayerNorm has one tap and no padding: its loads are never out of range, and what a lane
            // past the last frame reads only reaches outputs that are not stored
            const float b0 = LN_IN || ok ? bx[2 * q] : 0.f, b1 = LN_IN || ok ? bx[2 * q + 1] : 0.f;
            if constexpr (LN_IN) {
                const f32x2 bb = {b0, b1};
                ln_p1 += bb;
                ln_p2 = __builtin_elementwise_fma(bb, bb, ln_p2);
            }
            unsigned p0, p1, p2;
            split_pair(b0, b1, p0, p1, p2);
            P[0][q] = p0; P[1][q] = p1; P[2][q] = p2;
        }
        constexpr int ta[6] = {0, 2, 1, 0, 1, 0}, tb[6] = {2, 0, 1, 1, 0, 0};   // small partial products first
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, s0p[ta[k]]),
                                                           __builtin_bit_cast(bf16x8_t, P[tb[k]]), acc0, 0, 0, 0);
            if constexpr (MB == 2)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, s1p[ta[k]]),
                                                               __builtin_bit_cast(bf16x8_t, P[tb[k]]), acc1, 0, 0, 0);
        }
    };
    S3_LIN_STAMP(1);
    if constexpr (WLDS) {
        // Stage st = steps 2 st and 2 st + 1.  Its 12 KiB are 768 pieces of 16 bytes, block 0's two steps (384 pieces, contiguous
        // in memory) then block 1's; thread tid moves pieces tid, 256 + tid and 512 + tid.  One barrier per stage: a buffer is
        // rewritten two stages later, after the barrier of the stage between, which no wave passes before all have read it.
        const int nst = ns / 2;
        unsigned goff[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int pc = c * 256 + tid;
            goff[c] = (pc >= 384 ? w1d + (unsigned)(pc - 384) * 16u : (unsigned)pc * 16u);
        }
        u32x4 G[3];
        auto gload = [&](int st) {
#pragma unroll
            for (int c = 0; c < 3; ++c) G[c] = buf_load4u(wr_, goff[c], w0s + (unsigned)st * (2u * WSTEP));
        };
        float BX[2][8];                                        // x of a stage's two steps, requested a stage ahead
        bool okx[2];
        auto xload = [&](int slot) {                           // as load() above, the x half only
            int sg, tap, g;
            bool real;
            next_step(sg, tap, g, real);
            const int col = n + tap - a.pad;
            const bool ok = col >= 0 && col < a.L;
            okx[slot] = ok && real;
            const unsigned off = ok ? xlane + (unsigned)((tap - a.pad) * 4) : (unsigned)(8 * half * a.x_ld) * 4u;
#pragma unroll
            for (int j = 0; j < 8; ++j) BX[slot][j] = buf_load(xr_, off, (unsigned)(g * 16 + j) * xrow2);
        };
        gload(0);
        xload(0);
        xload(1);
        __builtin_amdgcn_sched_barrier(0);
        for (int st = 0; st < nst; ++st) {
            u32x4* const wb = wls + (st & 1) * 768;
#pragma unroll
            for (int c = 0; c < 3; ++c) wb[c * 256 + tid] = G[c];
            __builtin_amdgcn_sched_barrier(0);
            gload(st + 1 < nst ? st + 1 : st);                 // in flight under this stage's products
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 W0[3], W1[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    W0[pl] = wb[t * 192 + pl * 64 + lane];
                    W1[pl] = wb[384 + t * 192 + pl * 64 + lane];
                }
                split_step(BX[t], okx[t], W0, W1);
                __builtin_amdgcn_sched_barrier(0);
                xload(t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        load(i);
        __builtin_amdgcn_sched_barrier(0);                     // in slot order: the loop's first wait then leaves PF - 1 groups in flight
    }
    for (int s = 0; s < ns; s += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if constexpr (SPLIT) {
                split_step(B[i], okb[i], S0[i], S1[i]);
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bv = okb[i] ? B[i][j] : 0.f;
                if constexpr (LN_IN) { ln_s1 += bv; ln_s2 = fmaf(bv, bv, ln_s2); }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[i][j], bv, acc0, 0, 0, 0);
                if constexpr (MB == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[i][j], bv, acc1, 0, 0, 0);
            }
            }
            __builtin_amdgcn_sched_barrier(0);                 // keep the refill of this slot here, PF steps ahead of its use
            load(i);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
    S3_LIN_STAMP(2);
    float ln_mean = 0.f, ln_rstd = 0.f;                        // LN_IN: set once the frame's sums are complete
    // accumulator r of a lane: channel 8 (r / 4) + 4 (lane / 32) + r % 4 of its block of 32, frame lane % 32 of the tile.
    // Rows are finished four at a time (r % 4 = 0 .. 3: four consecutive channels): one 16-byte store per lane into the
    // frame-major tensor, four row stores into the channel-major one.
    auto finish4 = [&](const float* acc4, int gi) {
        const int cb = cb0 + (rbase(gi) >> 4), co0 = cb * 32 + 8 * ((rbase(gi) & 15) >> 2) + 4 * half;
        if (co0 >= a.Cout || n >= a.L || (cb != cb0 && !live1)) return;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (LN_IN) v[e] = ln_rstd * (acc4[e] - ln_mean * prsum[gi][e]) + pbias[gi][e];
            else v[e] = acc4[e] + pbias[gi][e];
            if constexpr (OUT_ACT == OUT_GELU) v[e] = gelu(v[e]);
            else if constexpr (OUT_ACT == OUT_SWISH) v[e] = v[e] * (1.0f / (1.0f + expf(-v[e])));
            else if constexpr (OUT_ACT == OUT_LRELU) v[e] = v[e] > 0.f ? v[e] : v[e] * 0.01f;
        }
        if (a.qp) {
            // four consecutive channels of q, k or v of one head at frame n, into attention operand order (cfm_split.h)
            const int part = co0 >> 9, hh = (co0 & 511) >> 6, d = co0 & 63;
            const size_t plane = (size_t)(a.p_ld / 32) * 2048, bh = ((size_t)z * kHeads + hh) * 3;
            if (part < 2) {
                unsigned short* const dst = (part == 0 ? a.qp : a.kp) + bh * plane + split_qk_offset(n, d);
                const float sc = part == 0 ? a.q_scale : 1.0f;
                unsigned lo[3], hi[3];
                split_pair(v[0] * sc, v[1] * sc, lo[0], lo[1], lo[2]);
                split_pair(v[2] * sc, v[3] * sc, hi[0], hi[1], hi[2]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(dst + pl * plane) = uint2{lo[pl], hi[pl]};
            } else {
                unsigned short* const dst = a.vp + bh * plane;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    unsigned pp[3];
                    split_pair(v[e], v[e + 1], pp[0], pp[1], pp[2]);
                    const size_t o0 = split_v_offset(n, d + e), o1 = split_v_offset(n, d + e + 1);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        dst[pl * plane + o0] = (unsigned short)(pp[pl] & 0xffffu);
                        dst[pl * plane + o1] = (unsigned short)(pp[pl] >> 16);
                    }
                }
            }
            return;
        }
        if (a.yt && co0 >= a.yt_from) {
            *reinterpret_cast<f32x4*>(a.yt + z * a.yt_bs + (size_t)n * a.yt_ld + (co0 - a.yt_from)) = f32x4{v[0], v[1], v[2], v[3]};
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (co0 + e < a.Cout) a.y[z * a.y_bs + (size_t)(co0 + e) * a.y_ld + n] = has_res ? v[e] + pres[LN_IN ? 0 : gi][e] : v[e];
    };

    if constexpr (LN_IN && SPLIT) { ln_s1 = ln_p1[0] + ln_p1[1]; ln_s2 = ln_p2[0] + ln_p2[1]; }
    if constexpr (LN_IN) {                                     // the two halves of a wave hold the two channel parities of a frame
        ln_s1 += __shfl_xor(ln_s1, 32);
        ln_s2 += __shfl_xor(ln_s2, 32);
    }
    if constexpr (KSPLIT > 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            part[(w * RW + r) * 64 + lane] = acc0[r];
            if constexpr (MB == 2) part[(w * RW + 16 + r) * 64 + lane] = acc1[r];
        }
        if constexpr (LN_IN) {
            if (lane < 32) { lnst[(w * 2) * 32 + lane] = ln_s1; lnst[(w * 2 + 1) * 32 + lane] = ln_s2; }
        }
        __syncthreads();
        S3_LIN_STAMP(3);
        if constexpr (LN_IN) {                                 // the parts of the reduction in wave order, like the accumulators
            ln_s1 = lnst[(ntile * KSPLIT * 2) * 32 + (lane & 31)];
            ln_s2 = lnst[(ntile * KSPLIT * 2 + 1) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 1; j < KSPLIT; ++j) {
                ln_s1 += lnst[((ntile * KSPLIT + j) * 2) * 32 + (lane & 31)];
                ln_s2 += lnst[((ntile * KSPLIT + j) * 2 + 1) * 32 + (lane & 31)];
            }
        }
        if constexpr (LN_IN) {
            const float inv_c = 1.0f / (float)a.Cin;
            ln_mean = ln_s1 * inv_c;
            ln_rstd = 1.0f / sqrtf(fmaxf(ln_s2 * inv_c - ln_mean * ln_mean, 0.f) + a.ln_eps);
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rbase(gi) + e;
                v[e] = part[(ntile * KSPLIT * RW + r) * 64 + lane];
#pragma unroll
                for (int j = 1; j < KSPLIT; ++j) v[e] += part[((ntile * KSPLIT + j) * RW + r) * 64 + lane];      // in wave order
            }
            finish4(v, gi);
        }
    } else {
        if constexpr (LN_IN) {
            const float inv_c = 1.0f / (float)a.Cin;
            ln_mean = ln_s1 * inv_c;
            ln_rstd = 1.0f / sqrtf(fmaxf(ln_s2 * inv_c - ln_mean * ln_mean, 0.f) + a.ln_eps);
        }
        // the operands of group gi + EPI_AHEAD are requested before group gi is finished: EPI_AHEAD + 1 groups of values live at a time
#ifndef S3_LIN_EPI_AHEAD
#define S3_LIN_EPI_AHEAD 1
#endif
        constexpr int EPI_AHEAD = S3_LIN_EPI_AHEAD;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gi = 0; gi < EPI_AHEAD && gi < NG; ++gi) load_epilogue_operands(gi);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi + EPI_AHEAD < NG) load_epilogue_operands(gi + EPI_AHEAD);
            __builtin_amdgcn_sched_barrier(0);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gi < 4 ? acc0[4 * gi + e] : acc1[4 * (gi & 3) + e];
            finish4(v, gi);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef S3_LIN_TIMELINE
    S3_LIN_STAMP(4);
    tl[7] = (long long)wall_clock64();
    if (a_.timeline && lane == 0) {
        long long* dst = a_.timeline + ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x / 64) + w) * 8;
        if constexpr (KSPLIT == 1) tl[3] = tl[2];
        tl[5] = 0;
        for (int i = 0; i < 8; ++i) dst[i] = tl[i];
    }
#endif
}

// Weights in lin_kernel's fp32 tile order -> three bf16 planes in the operand order of v_mfma_f32_32x32x16_bf16:
//   dst[(((cb * steps16 + tap * (Cin / 16) + g) * 3 + plane) * 64 + lane) * 8 + j] = plane of W[cb * 32 + lane % 32][16 g + 8 (lane / 32) + j][tap]
// one thread per (block, step, lane)
__global__ void split_weights_kernel(const float* __restrict__ wt, unsigned short* __restrict__ dst, int nblocks, int K, int Cin) {
    const int g16n = Cin / 16, g8n = Cin / 8, steps16 = K * g16n;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nblocks * steps16 * 64) return;
    const int lane = (int)(idx & 63), sg = (int)((idx >> 6) % steps16), cb = (int)((idx >> 6) / steps16);
    const int tap = sg / g16n, g = sg - tap * g16n, co = lane & 31, hh = lane >> 5;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // channel 16 g + 8 hh + j sits in fp32 group 2 g + hh at element j / 2 of source lane (j % 2) * 32 + co
        const size_t src = ((((size_t)cb * K + tap) * g8n + (2 * g + hh)) * 64 + (j & 1) * 32 + co) * 4 + (j >> 1);
        x[j] = wt[src];
    }
    unsigned p[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(x[2 * q], x[2 * q + 1], p[0][q], p[1][q], p[2][q]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<u32x4*>(dst + ((((size_t)cb * steps16 + sg) * 3 + pl) * 64 + lane) * 8) = u32x4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]};
}

// The estimator's input (decoder.py:254-260): rows 0..79 x, 80..159 mu, 160..239 the speaker vector on every frame,
// 240..319 cond.  uncond_odd: rows 2 u and 2 u + 1 are the two calls of solve_euler for utterance u (flow_matching.py:105-118):
// the odd row takes the same x and zeros for the rest; otherwise the four inputs hold one row per row of the launch.
__global__ void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ mu, const float* __restrict__ spks,
                                  const float* __restrict__ cond, int L, int uncond_odd, float* __restrict__ in, int ld) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, r = blockIdx.z;
    if (t >= L) return;
    const int part = c / kMel, cc = c % kMel;
    const int src_row = uncond_odd ? r >> 1 : r;
    float v;
    if (part == 0) v = x[((size_t)src_row * kMel + cc) * L + t];
    else if (uncond_odd && (r & 1)) v = 0.f;
    else if (part == 1) v = mu[((size_t)src_row * kMel + cc) * L + t];
    else if (part == 2) v = spks[src_row * kMel + cc];
    else v = cond[((size_t)src_row * kMel + cc) * L + t];
    in[((size_t)r * kIn + c) * ld + t] = v;
}

// One Euler step with classifier-free guidance (flow_matching.py:125-127): x += dt * ((1 + rate) d[0] - rate d[1]), written
// to the x rows of both estimator inputs and, when mel is not null, to the caller's [80][L] output; blockIdx.z is the
// utterance (rows 2 z and 2 z + 1).  Each product and sum is rounded on its own, as the reference's separate tensor
// operations are.
__global__ void euler_step_kernel(const float* __restrict__ d, size_t d_bs, float* __restrict__ in, size_t in_bs, int ld, int L, float dt,
                                  float one_plus_rate, float rate, float* __restrict__ mel) {
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, u = blockIdx.z;
    if (t >= L) return;
    const size_t o = (size_t)c * ld + t;
    d += 2 * u * d_bs;
    in += 2 * u * in_bs;
    const float a = one_plus_rate * d[o], b = rate * d[d_bs + o];      // plain operators: the pragma above binds them
    const float g = a - b;
    const float p = dt * g;
    const float xn = in[o] + p;
    in[o] = xn;
    in[in_bs + o] = xn;
    if (mel) mel[((size_t)u * kMel + c) * L + t] = xn;
}

__global__ void copy_rows_kernel(const float* __restrict__ src, size_t src_bs, float* __restrict__ dst, size_t dst_bs, int ld, int L) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, r = blockIdx.z;
    if (t < L) dst[r * dst_bs + (size_t)c * ld + t] = src[r * src_bs + (size_t)c * ld + t];
}

__global__ void unpack_output_kernel(const float* __restrict__ d, size_t d_bs, int ld, int L, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, r = blockIdx.z;
    if (t < L) out[((size_t)r * kMel + c) * L + t] = d[r * d_bs + (size_t)c * ld + t];
}

// ---- host side ---------------------------------------------------------------------------------------------------

struct Host { std::vector<float> v; std::vector<int64_t> shape; };
struct Lin { float* wt = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0, K = 1; unsigned short* wsp = nullptr; };   // wsp: split planes, built on demand
struct Norm { float* g = nullptr; float* b = nullptr; };
struct Resnet { Lin conv1, conv2, res; Norm n1, n2; };
struct Transformer { Lin qkv, out, w1, w2; float *qkv_rsum = nullptr, *w1_rsum = nullptr; };   // norm1 / norm3 folded into qkv / w1
struct Stage { Resnet res; Transformer tf[kBlocks]; Lin tail; bool has_tail = false; };

}  // namespace

struct s3cfm {
    int device = 0, max_frames = 0, ld = 0, max_batch = 1;
    int rows = kRows;                                          // rows of the running call: two per utterance
    int Lk[kMaxUtt] = {};                                      // frames of each utterance of the running call
    bool finalized = false;
    std::map<std::string, Host> raw;
    std::vector<void*> owned;
    Stage stages[kStages];
    Lin final_conv, final_proj;
    Norm final_norm;
    float *freq = nullptr, *t_w1 = nullptr, *t_b1 = nullptr, *t_w2 = nullptr, *t_b2 = nullptr, *mlp_w = nullptr, *mlp_b = nullptr;
    float *emb = nullptr, *th = nullptr, *temb = nullptr, *tvec = nullptr;
    float *IN = nullptr, *X = nullptr, *Y = nullptr, *Z = nullptr, *CAT = nullptr, *QK = nullptr, *VT = nullptr, *O = nullptr, *H = nullptr,
          *D = nullptr;
    bool split = false;                                        // attention on split-bf16 operands (s3cfm_set_split_bf16)
    bool split_lin = false;                                    // the linear layers and convolutions too (mode 2)
    unsigned short *QP = nullptr, *KP = nullptr, *VP = nullptr;     // its operand tensors (cfm_split.h), allocated on first use
};

namespace {

int dalloc(s3cfm* h, float** p, size_t n) {
    S3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float)));
    h->owned.push_back(*p);
    return 0;
}
int upload(s3cfm* h, float** dst, const float* src, size_t n) {
    if (dalloc(h, dst, n)) return 1;
    S3_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
int get(s3cfm* h, const std::string& name, std::vector<int64_t> shape, const Host** out) {
    auto it = h->raw.find(name);
    if (it == h->raw.end()) { s3_set_error("missing tensor: " + name); return 1; }
    if (it->second.shape != shape) { s3_set_error("bad shape for tensor: " + name); return 1; }
    *out = &it->second;
    return 0;
}
// Linear [Cout, Cin] (K = 0) or Conv1d [Cout, Cin, K] -> tile order; no bias tensor = zeros
int make_lin(s3cfm* h, Lin* l, const std::string& name, int Cout, int Cin, int K, bool has_bias) {
    const Host* w;
    if (K == 0 ? get(h, name + ".weight", {Cout, Cin}, &w) : get(h, name + ".weight", {Cout, Cin, K}, &w)) return 1;
    l->Cin = Cin; l->Cout = Cout; l->K = std::max(K, 1);
    const std::vector<float> t = s3_tile_conv_weights(w->v.data(), Cout, Cin, l->K);
    if (upload(h, &l->wt, t.data(), t.size())) return 1;
    std::vector<float> b(Cout, 0.f);
    if (has_bias) {
        const Host* bt;
        if (get(h, name + ".bias", {Cout}, &bt)) return 1;
        b = bt->v;
    }
    return upload(h, &l->bias, b.data(), Cout);
}
// Linear [Cout][Cin] (+ bias) applied to LayerNorm(x) with gain gamma and shift beta -> what lin_kernel's folded form takes
// (LinArgs::rsum): W' = W diag(gamma) in tile order, b' = b + W beta, rsum = row sums of W'; sums in double
int make_lin_ln(s3cfm* h, Lin* l, float** rsum, const std::vector<float>& w, const std::vector<float>* bias, int Cout, int Cin,
                const std::string& norm) {
    const Host *g, *b;
    if (get(h, norm + ".weight", {Cin}, &g) || get(h, norm + ".bias", {Cin}, &b)) return 1;
    std::vector<float> wf((size_t)Cout * Cin), bf(Cout), rs(Cout);
    for (int o = 0; o < Cout; ++o) {
        double sb = bias ? (double)(*bias)[o] : 0.0, sr = 0.0;
        for (int c = 0; c < Cin; ++c) {
            const float wv = w[(size_t)o * Cin + c] * g->v[c];
            wf[(size_t)o * Cin + c] = wv;
            sr += (double)wv;
            sb += (double)w[(size_t)o * Cin + c] * (double)b->v[c];
        }
        bf[o] = (float)sb;
        rs[o] = (float)sr;
    }
    l->Cin = Cin; l->Cout = Cout; l->K = 1;
    const std::vector<float> t = s3_tile_conv_weights(wf.data(), Cout, Cin, 1);
    return upload(h, &l->wt, t.data(), t.size()) || upload(h, &l->bias, bf.data(), Cout) || upload(h, rsum, rs.data(), Cout);
}

int make_norm(s3cfm* h, Norm* n, const std::string& name) {
    const Host *g, *b;
    if (get(h, name + ".weight", {kC}, &g) || get(h, name + ".bias", {kC}, &b)) return 1;
    return upload(h, &n->g, g->v.data(), kC) || upload(h, &n->b, b->v.data(), kC);
}

}  // namespace

int launch_lin(const LinArgs& a, int out_act, int rows, hipStream_t st) {
    S3_REQUIRE(a.Cin % 64 == 0 && a.x != a.y, "linear layer: input channels must be a multiple of 64 and the output a different tensor");
    S3_REQUIRE(!a.rsum || (a.K == 1 && a.pad == 0 && !a.res && (out_act == OUT_NONE || out_act == OUT_GELU)),
               "linear layer: a folded LayerNorm needs a one-tap layer without or with GELU and without a residual");
    const int L = a.L, mb = (a.CoutPad + 63) / 64;
    // enough 64 x 128 tiles to give every CU several workgroups: no split of the reduction
    static const int wide_min = getenv("S3CFM_WIDE_MIN") ? atoi(getenv("S3CFM_WIDE_MIN")) : 1024;
    static const int force_ks = getenv("S3CFM_KS") ? atoi(getenv("S3CFM_KS")) : 0;
    int ks = 4;
    if ((long long)mb * ((L + 127) / 128) * rows >= wide_min) ks = 1;
    else if ((long long)mb * ((L + 31) / 32) * rows < 512) ks = 8;
    else if ((long long)mb * ((L + 63) / 64) * rows >= 768) ks = 2;      // longer loops per wave once there are three workgroups per CU anyway
    if (force_ks > 0 && !(ks == 8 && force_ks < 4)) ks = force_ks;
    // One block of 32 channels per wave instead of two (twice the workgroups, each half the work) when the launch is small
    // and its workgroups pack badly into the 256 CUs: 280 workgroups take 2 units of time, 560 half-size ones 1.5.
    static const int force_mb = getenv("S3CFM_LIN_MB") ? atoi(getenv("S3CFM_LIN_MB")) : 0;
    int blocks = 2;
    if (ks == 8) {
        const long long u2 = (long long)mb * ((L + 31) / 32) * rows, u1 = (long long)(a.CoutPad / 32) * ((L + 31) / 32) * rows;
        if ((double)((u1 + 255) / 256) * 1.1 < (double)((u2 + 255) / 256) * 2.0) blocks = 1;
        if (force_mb == 1 || force_mb == 2) blocks = force_mb;
        if (blocks == 1) ks = 4;
    }
    const int ntw = ks == 8 ? 1 : 4 / ks;
    const dim3 grid((L + 32 * ntw - 1) / (32 * ntw), blocks == 2 ? mb : a.CoutPad / 32, rows), block(ks == 8 ? 512 : 256);
#define LIN_LEAD(a) (a).x, (a).wt, (a).x_bs, (a).x_ld, (a).Cin, (a).CoutPad, (a).K, (a).L, (a).pad, (a).Cout
#define S3_LIN_LAUNCH(KS, MB, DEEP)                                                                                       \
    do {                                                                                                                  \
        if (a.rsum && out_act == OUT_GELU) hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_GELU, true, false, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);   \
        else if (a.rsum) hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_NONE, true, false, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);                     \
        else if (out_act == OUT_GELU) hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_GELU, false, false, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);       \
        else if (out_act == OUT_SWISH) hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_SWISH, false, false, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);     \
        else if (out_act == OUT_LRELU) hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_LRELU, false, false, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);     \
        else hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_NONE, false, false, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);                                \
    } while (0)
    // split-bf16 operands when the planes exist, the layer is one of the decoder's forms and the 16-channel steps divide
    // evenly among the parts of the reduction
    const bool split = a.wsp && (a.K * (a.Cin / 16)) % ks == 0 && (out_act == OUT_NONE || (out_act == OUT_GELU && a.rsum));
#define S3_LIN_LAUNCH_SPLIT(KS, MB, DEEP)                                                                                 \
    do {                                                                                                                  \
        if (a.rsum && out_act == OUT_GELU) hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_GELU, true, true, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);  \
        else if (a.rsum) hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_NONE, true, true, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);                    \
        else hipLaunchKernelGGL((lin_kernel<KS, MB, OUT_NONE, false, true, DEEP>), grid, block, 0, st, LIN_LEAD(a), a);                               \
    } while (0)
#define S3_LIN_FORMS(LAUNCH, DEEP)                                                                                        \
    do {                                                                                                                  \
        if (blocks == 1) LAUNCH(4, 1, DEEP);                                                                              \
        else if (ks == 8) LAUNCH(8, 2, DEEP);                                                                             \
        else if (ks == 4) LAUNCH(4, 2, DEEP);                                                                             \
        else if (ks == 2) LAUNCH(2, 2, DEEP);                                                                             \
        else LAUNCH(1, 2, DEEP);                                                                                          \
    } while (0)
    static const int deep_below = getenv("S3CFM_DEEP_BELOW") ? atoi(getenv("S3CFM_DEEP_BELOW")) : 1200;
    const bool deep = (long long)L * rows < deep_below;        // columns of the launch
    static const bool wlds = !(getenv("S3CFM_WLDS") && atoi(getenv("S3CFM_WLDS")) == 0);
    if (split && ks == 1 && blocks == 2 && wlds) {             // the large launches of the split form: weights once per workgroup, through LDS
        if (a.rsum && out_act == OUT_GELU) hipLaunchKernelGGL((lin_kernel<1, 2, OUT_GELU, true, true, false, true>), grid, block, 0, st, LIN_LEAD(a), a);
        else if (a.rsum) hipLaunchKernelGGL((lin_kernel<1, 2, OUT_NONE, true, true, false, true>), grid, block, 0, st, LIN_LEAD(a), a);
        else hipLaunchKernelGGL((lin_kernel<1, 2, OUT_NONE, false, true, false, true>), grid, block, 0, st, LIN_LEAD(a), a);
    } else if (split) { if (deep) S3_LIN_FORMS(S3_LIN_LAUNCH_SPLIT, true); else S3_LIN_FORMS(S3_LIN_LAUNCH_SPLIT, false); }
    else { if (deep) S3_LIN_FORMS(S3_LIN_LAUNCH, true); else S3_LIN_FORMS(S3_LIN_LAUNCH, false); }
#undef S3_LIN_FORMS
#undef S3_LIN_LAUNCH
#undef S3_LIN_LAUNCH_SPLIT
#undef LIN_LEAD
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

namespace {

int launch_attn(const AttnArgs& a, hipStream_t st) {
    static const int force = getenv("S3CFM_ATTN_QT") ? atoi(getenv("S3CFM_ATTN_QT")) : 0;
    static const bool qlds = !(getenv("S3CFM_ATTN_QLDS") && atoi(getenv("S3CFM_ATTN_QLDS")) == 0);
    // Query tiles per wave (QT).  A workgroup of QT tiles costs its CU about QT units of matrix work and a launch takes as long
    // as its most loaded CU, k = ceil(workgroups / 256) of them: pick the QT with the smallest k x QT.  With the queries in LDS
    // (the default) three / two / two workgroups of 1 / 2 / 3 tiles are resident on a CU at a time, so that one wave's softmax
    // runs under another's matrix instructions; a workgroup that runs alone (k = 1, or the last one after full sets) lacks that
    // and costs a tenth more, and a tile costs less the more tiles share a wave's K and V loads.  Fitted to per-launch times at
    // 500 .. 3 000 frames with one and four utterances per call (1 000 frames: 512 workgroups of one tile 42.9 us, 256 of two
    // 45.5; 2 000: two tiles 150, one 160, three 209; 3 000: three tiles 324, two 406).
    const int tiles = (a.L + 31) / 32;
    int best = 1;
    double best_cost = 1e30;
    for (int qt = 1; qt <= 3; ++qt) {                          // four tiles per wave would spill registers
        const long long wgs = (long long)((tiles + qt - 1) / qt) * a.rows * kHeads, k = (wgs + 255) / 256;
        double cost;
        if (qlds) {
            const int occ = qt == 1 ? 3 : 2;
            const double per_tile = qt == 1 ? 1.05 : (qt == 2 ? 1.0 : 0.95);
            cost = (double)k * qt * per_tile * (k % occ == 1 ? 1.1 : 1.0);
        } else {
            cost = (double)k * qt * (qt == 1 ? 1.02 : 1.0);    // queries in registers: one workgroup of two or three tiles per CU
        }
        if (cost < best_cost - 1e-9) { best_cost = cost; best = qt; }
    }
    if (force >= 1 && force <= 3) best = force;
    const dim3 grid((tiles + best - 1) / best * a.rows * kHeads);
#define S3_ATTN_LAUNCH(QT, QL) hipLaunchKernelGGL((attn_kernel<QT, QL>), grid, dim3(256), 0, st, a.q, a.k, a.vt, a.qk_bs, a.vt_bs, a.ld, a.L, a.rows, a)
    if (qlds) { if (best == 1) S3_ATTN_LAUNCH(1, true); else if (best == 2) S3_ATTN_LAUNCH(2, true); else S3_ATTN_LAUNCH(3, true); }
    else { if (best == 1) S3_ATTN_LAUNCH(1, false); else if (best == 2) S3_ATTN_LAUNCH(2, false); else S3_ATTN_LAUNCH(3, false); }
#undef S3_ATTN_LAUNCH
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

// y[row][Cout][L] = out_act(conv(x[row][Cin][L])) (+ res) on both rows; pad = zero frames on the left.  y must not be x.
int run_lin(const s3cfm* h, const Lin& l, const float* x, int x_rows, float* y, int y_rows, int L, int pad, int out_act, const float* res,
            hipStream_t st, float* yt = nullptr, int yt_from = 0, const float* rsum = nullptr, bool to_planes = false) {
    LinArgs a{};
    a.x = x; a.x_bs = (size_t)x_rows * h->ld; a.x_ld = h->ld;
    a.wt = l.wt; a.bias = l.bias;
    a.y = y; a.y_bs = (size_t)y_rows * h->ld; a.y_ld = h->ld;
    a.res = res;
    a.yt = yt; a.yt_bs = (size_t)h->ld * kVtLd; a.yt_ld = kVtLd; a.yt_from = yt_from;
    a.Cin = l.Cin; a.Cout = l.Cout; a.CoutPad = (l.Cout + 31) / 32 * 32; a.K = l.K; a.pad = pad; a.L = L;
    a.rsum = rsum; a.ln_eps = 1e-5f;
    if (to_planes) { a.qp = h->QP; a.kp = h->KP; a.vp = h->VP; a.p_ld = h->ld; a.q_scale = kQScale; }
    if (h->split_lin) a.wsp = l.wsp;
    return launch_lin(a, out_act, h->rows, st);
}

int run_ln(const s3cfm* h, int mode, const float* x, float* y, int L, const Norm& n, const float* add, hipStream_t st) {
    const dim3 grid((L + kLnFrames - 1) / kLnFrames, h->rows);
    const size_t bs = (size_t)kC * h->ld;
    if (mode == LN_PLAIN) hipLaunchKernelGGL(ln256_kernel<LN_PLAIN>, grid, dim3(256), 0, st, x, y, h->ld, bs, L, n.g, n.b, add);
    else if (mode == LN_MISH) hipLaunchKernelGGL(ln256_kernel<LN_MISH>, grid, dim3(256), 0, st, x, y, h->ld, bs, L, n.g, n.b, add);
    else hipLaunchKernelGGL(ln256_kernel<LN_MISH_ADD>, grid, dim3(256), 0, st, x, y, h->ld, bs, L, n.g, n.b, add);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

// CausalResnetBlock1D (decoder.py:64-68, matcha/decoder.py:56-61): x [row][Cin][L] -> Y [row][256][L]; x may be X (the
// convolution kernel must not write the tensor it reads), so the caller swaps X and Y afterwards
int run_resnet(s3cfm* h, const Resnet& r, const float* x, int x_rows, const float* tvec, int L, hipStream_t st) {
    if (run_lin(h, r.conv1, x, x_rows, h->Y, kC, L, 2, OUT_NONE, nullptr, st)) return 1;
    if (run_ln(h, LN_MISH_ADD, h->Y, h->Y, L, r.n1, tvec, st)) return 1;
    if (run_lin(h, r.conv2, h->Y, kC, h->Z, kC, L, 2, OUT_NONE, nullptr, st)) return 1;
    if (run_ln(h, LN_MISH, h->Z, h->Z, L, r.n2, nullptr, st)) return 1;
    return run_lin(h, r.res, x, x_rows, h->Y, kC, L, 0, OUT_NONE, h->Z, st);
}

// BasicTransformerBlock.forward (matcha/transformer.py:243-316) on X [row][256][L], in place
int run_transformer(s3cfm* h, const Transformer& b, int L, hipStream_t st) {
    const int ld = h->ld;
    // norm1 rides in the q/k/v projection (LinArgs::rsum); q and k channel-major into QK, v frame-major into VT
    if (run_lin(h, b.qkv, h->X, kC, h->QK, 2 * kInner, L, 0, OUT_NONE, nullptr, st, h->VT, 2 * kInner, b.qkv_rsum, h->split)) return 1;
    if (h->split) {
        SplitAttnArgs sa{h->QP, h->KP, h->VP, h->O, (size_t)kInner * ld, ld, L, h->rows, {}};
        for (int u = 0; u < kMaxUtt; ++u) sa.Lk[u] = h->Lk[u];
        if (launch_attn_split(sa, st)) return 1;
    } else {
        AttnArgs at{h->QK, h->QK + (size_t)kInner * ld, (size_t)2 * kInner * ld, ld, h->VT, (size_t)ld * kVtLd, h->O, (size_t)kInner * ld, L, h->rows, {}};
        for (int u = 0; u < kMaxUtt; ++u) at.Lk[u] = h->Lk[u];
        if (launch_attn(at, st)) return 1;
    }
    if (run_lin(h, b.out, h->O, kInner, h->X, kC, L, 0, OUT_NONE, h->X, st)) return 1;
    // norm3 rides in the first feed-forward layer
    if (run_lin(h, b.w1, h->X, kC, h->H, kFf, L, 0, OUT_GELU, nullptr, st, nullptr, 0, b.w1_rsum)) return 1;
    return run_lin(h, b.w2, h->H, kFf, h->X, kC, L, 0, OUT_NONE, h->X, st);
}

int copy_rows(const float* src, int src_rows, float* dst, int dst_rows, int ld, int L, int rows, hipStream_t st) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3((L + 255) / 256, kC, rows), dim3(256), 0, st, src, (size_t)src_rows * ld, dst,
                       (size_t)dst_rows * ld, ld, L);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

// ConditionalDecoder.forward (decoder.py:233-317) on IN [rows][320][L] at time t -> D [rows][80][L]; h->rows and h->Lk are set by the caller
int run_estimator(s3cfm* h, float t, int L, hipStream_t st) {
    // time embedding and its 14 per-stage projections (decoder.py:251-252, matcha/decoder.py:58)
    hipLaunchKernelGGL(sin_emb_kernel, dim3(1), dim3(192), 0, st, t, h->freq, h->emb);
    hipLaunchKernelGGL((gemv_kernel<GV_NONE, GV_SILU>), dim3(kTime / 4), dim3(256), 0, st, h->t_w1, h->t_b1, h->emb, h->th, kIn, kTime);
    hipLaunchKernelGGL((gemv_kernel<GV_NONE, GV_NONE>), dim3(kTime / 4), dim3(256), 0, st, h->t_w2, h->t_b2, h->th, h->temb, kTime, kTime);
    hipLaunchKernelGGL((gemv_kernel<GV_MISH, GV_NONE>), dim3(kStages * kC / 4), dim3(256), 0, st, h->mlp_w, h->mlp_b, h->temb, h->tvec, kTime,
                       kStages * kC);
    S3_HIP_CHECK(hipGetLastError());
    const int ld = h->ld;
    for (int s = 0; s < kStages; ++s) {
        const Stage& sg = h->stages[s];
        const float* x = s == 0 ? h->IN : (s == kStages - 1 ? h->CAT : h->X);
        const int x_rows = s == 0 ? kIn : (s == kStages - 1 ? 2 * kC : kC);
        if (run_resnet(h, sg.res, x, x_rows, h->tvec + s * kC, L, st)) return 1;
        std::swap(h->X, h->Y);
        for (int j = 0; j < kBlocks; ++j)
            if (run_transformer(h, sg.tf[j], L, st)) return 1;
        if (s == 0) {
            // the skip connection (decoder.py:278), then the stage's trailing causal convolution
            if (copy_rows(h->X, kC, h->CAT + (size_t)kC * ld, 2 * kC, ld, L, h->rows, st)) return 1;
            if (run_lin(h, sg.tail, h->X, kC, h->Y, kC, L, 2, OUT_NONE, nullptr, st)) return 1;
            std::swap(h->X, h->Y);
        } else if (s == kStages - 2) {
            if (copy_rows(h->X, kC, h->CAT, 2 * kC, ld, L, h->rows, st)) return 1;        // decoder.py:301
        } else if (s == kStages - 1) {
            if (run_lin(h, sg.tail, h->X, kC, h->Y, kC, L, 2, OUT_NONE, nullptr, st)) return 1;
            std::swap(h->X, h->Y);
        }
    }
    // final CausalBlock1D and 1x1 projection (decoder.py:315-316)
    if (run_lin(h, h->final_conv, h->X, kC, h->Y, kC, L, 2, OUT_NONE, nullptr, st)) return 1;
    if (run_ln(h, LN_MISH, h->Y, h->Y, L, h->final_norm, nullptr, st)) return 1;
    return run_lin(h, h->final_proj, h->Y, kC, h->D, kMel, L, 0, OUT_NONE, nullptr, st);
}

}  // namespace

extern "C" int s3cfm_create_batched(int32_t device, int32_t max_frames, int32_t max_batch, s3cfm** out) {
    S3_REQUIRE(out && max_frames >= 1 && max_frames <= S3CFM_MAX_FRAMES && max_batch >= 1 && max_batch <= S3CFM_MAX_BATCH, "bad arguments");
    S3_HIP_CHECK(hipSetDevice(device));
    s3cfm* h = new s3cfm();
    h->device = device; h->max_frames = max_frames; h->max_batch = max_batch;
    h->ld = (max_frames + 63) / 64 * 64;
    if ((h->ld / 64) % 2 == 0) h->ld += 64;
    *out = h;
    return 0;
}

extern "C" int s3cfm_create(int32_t device, int32_t max_frames, s3cfm** out) { return s3cfm_create_batched(device, max_frames, 1, out); }

extern "C" int s3cfm_destroy(s3cfm* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->owned) (void)hipFree(p);
    delete h;
    return 0;
}

extern "C" int s3cfm_load_tensor(s3cfm* h, const char* name, const float* host_data, const int64_t* shape, int32_t ndim) {
    S3_REQUIRE(h && name && host_data && shape && ndim >= 1 && ndim <= 3, "bad arguments");
    S3_REQUIRE(!h->finalized, "handle already finalized");
    Host t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { S3_REQUIRE(shape[i] >= 1, "bad shape"); t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.v.assign(host_data, host_data + n);
    for (float v : t.v) S3_REQUIRE(std::isfinite(v), "tensor holds a value that is not finite");
    h->raw[name] = std::move(t);
    return 0;
}

extern "C" int s3cfm_finalize(s3cfm* h) {
    S3_REQUIRE(h && !h->finalized, "bad handle");
    S3_HIP_CHECK(hipSetDevice(h->device));
    const std::string P = "decoder.estimator.";
    const Host *w, *b;
    if (get(h, P + "time_mlp.linear_1.weight", {kTime, kIn}, &w) || upload(h, &h->t_w1, w->v.data(), w->v.size())) return 1;
    if (get(h, P + "time_mlp.linear_1.bias", {kTime}, &b) || upload(h, &h->t_b1, b->v.data(), kTime)) return 1;
    if (get(h, P + "time_mlp.linear_2.weight", {kTime, kTime}, &w) || upload(h, &h->t_w2, w->v.data(), w->v.size())) return 1;
    if (get(h, P + "time_mlp.linear_2.bias", {kTime}, &b) || upload(h, &h->t_b2, b->v.data(), kTime)) return 1;
    std::vector<float> mlp_w((size_t)kStages * kC * kTime), mlp_b((size_t)kStages * kC);
    for (int s = 0; s < kStages; ++s) {
        const std::string sp = P + (s == 0 ? std::string("down_blocks.0.") : (s == kStages - 1 ? std::string("up_blocks.0.")
                                                                                                : "mid_blocks." + std::to_string(s - 1) + "."));
        Stage& sg = h->stages[s];
        const int cin = s == 0 ? kIn : (s == kStages - 1 ? 2 * kC : kC);
        const std::string rp = sp + "0.";
        if (make_lin(h, &sg.res.conv1, rp + "block1.block.0", kC, cin, 3, true) || make_norm(h, &sg.res.n1, rp + "block1.block.2")) return 1;
        if (make_lin(h, &sg.res.conv2, rp + "block2.block.0", kC, kC, 3, true) || make_norm(h, &sg.res.n2, rp + "block2.block.2")) return 1;
        if (make_lin(h, &sg.res.res, rp + "res_conv", kC, cin, 1, true)) return 1;
        if (get(h, rp + "mlp.1.weight", {kC, kTime}, &w) || get(h, rp + "mlp.1.bias", {kC}, &b)) return 1;
        std::copy(w->v.begin(), w->v.end(), mlp_w.begin() + (size_t)s * kC * kTime);
        std::copy(b->v.begin(), b->v.end(), mlp_b.begin() + (size_t)s * kC);
        for (int j = 0; j < kBlocks; ++j) {
            const std::string tp = sp + "1." + std::to_string(j) + ".";
            Transformer& tf = sg.tf[j];
            const Host *wq, *wk, *wv;
            if (get(h, tp + "attn1.to_q.weight", {kInner, kC}, &wq) || get(h, tp + "attn1.to_k.weight", {kInner, kC}, &wk) ||
                get(h, tp + "attn1.to_v.weight", {kInner, kC}, &wv))
                return 1;
            std::vector<float> wst;                        // one projection with 1536 outputs: q, k, v (no biases)
            wst.insert(wst.end(), wq->v.begin(), wq->v.end());
            wst.insert(wst.end(), wk->v.begin(), wk->v.end());
            wst.insert(wst.end(), wv->v.begin(), wv->v.end());
            if (make_lin_ln(h, &tf.qkv, &tf.qkv_rsum, wst, nullptr, 3 * kInner, kC, tp + "norm1")) return 1;
            if (make_lin(h, &tf.out, tp + "attn1.to_out.0", kC, kInner, 0, true)) return 1;
            const Host *w1, *b1;
            if (get(h, tp + "ff.net.0.proj.weight", {kFf, kC}, &w1) || get(h, tp + "ff.net.0.proj.bias", {kFf}, &b1)) return 1;
            if (make_lin_ln(h, &tf.w1, &tf.w1_rsum, w1->v, &b1->v, kFf, kC, tp + "norm3")) return 1;
            if (make_lin(h, &tf.w2, tp + "ff.net.2", kC, kFf, 0, true)) return 1;
        }
        sg.has_tail = s == 0 || s == kStages - 1;
        if (sg.has_tail && make_lin(h, &sg.tail, sp + "2", kC, kC, 3, true)) return 1;
    }
    if (upload(h, &h->mlp_w, mlp_w.data(), mlp_w.size()) || upload(h, &h->mlp_b, mlp_b.data(), mlp_b.size())) return 1;
    if (make_lin(h, &h->final_conv, P + "final_block.block.0", kC, kC, 3, true) || make_norm(h, &h->final_norm, P + "final_block.block.2")) return 1;
    if (make_lin(h, &h->final_proj, P + "final_proj", kMel, kC, 1, true)) return 1;
    // f_i = exp(i * -(ln 10000 / 159)) in fp32, as matcha/decoder.py:24-26 computes it
    std::vector<float> freq(kIn / 2);
    const float step = -(float)(std::log(10000.0) / (kIn / 2 - 1));
    for (int i = 0; i < kIn / 2; ++i) freq[i] = std::exp((float)i * step);
    if (upload(h, &h->freq, freq.data(), freq.size())) return 1;
    const size_t ld = h->ld;
    if (dalloc(h, &h->emb, kIn) || dalloc(h, &h->th, kTime) || dalloc(h, &h->temb, kTime) || dalloc(h, &h->tvec, kStages * kC)) return 1;
    const size_t mr = (size_t)kRows * h->max_batch;
    if (dalloc(h, &h->IN, mr * kIn * ld) || dalloc(h, &h->CAT, mr * 2 * kC * ld) || dalloc(h, &h->D, mr * kMel * ld)) return 1;
    for (float** p : {&h->X, &h->Y, &h->Z})
        if (dalloc(h, p, mr * kC * ld)) return 1;
    if (dalloc(h, &h->QK, mr * 2 * kInner * ld) || dalloc(h, &h->VT, mr * kVtLd * ld) || dalloc(h, &h->O, mr * kInner * ld) ||
        dalloc(h, &h->H, mr * kFf * ld))
        return 1;
    // the attention kernel reads K and V up to the end of a 32-frame tile: what lies past an utterance must be finite
    S3_HIP_CHECK(hipMemset(h->QK, 0, mr * 2 * kInner * ld * sizeof(float)));
    S3_HIP_CHECK(hipMemset(h->VT, 0, mr * kVtLd * ld * sizeof(float)));
    S3_HIP_CHECK(hipDeviceSynchronize());
    h->raw.clear();
    h->finalized = true;
    return 0;
}

extern "C" int s3cfm_set_split_bf16(s3cfm* h, int32_t on) {
    S3_REQUIRE(h && h->finalized, "handle not finalized");
    S3_HIP_CHECK(hipSetDevice(h->device));
    S3_HIP_CHECK(hipDeviceSynchronize());
    if (on && !h->QP) {
        const size_t bytes = (size_t)kRows * h->max_batch * 3 * kInner * h->ld * sizeof(unsigned short);
        for (unsigned short** p : {&h->QP, &h->KP, &h->VP}) {
            S3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), bytes));
            h->owned.push_back(*p);
            S3_HIP_CHECK(hipMemset(*p, 0, bytes));
        }
        S3_HIP_CHECK(hipDeviceSynchronize());
    }
    if (on >= 2 && !h->final_proj.wsp) {
        // every layer's weights once more as bf16 planes, made on the GPU from the fp32 tile order already there
        std::vector<Lin*> lins = {&h->final_conv, &h->final_proj};
        for (Stage& sg : h->stages) {
            lins.insert(lins.end(), {&sg.res.conv1, &sg.res.conv2, &sg.res.res});
            for (Transformer& tf : sg.tf) lins.insert(lins.end(), {&tf.qkv, &tf.out, &tf.w1, &tf.w2});
            if (sg.has_tail) lins.push_back(&sg.tail);
        }
        for (Lin* l : lins) {
            const int nblocks = (l->Cout + 31) / 32, steps16 = l->K * (l->Cin / 16);
            const size_t n = (size_t)nblocks * steps16 * 3 * 512;              // bf16 elements
            S3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&l->wsp), n * sizeof(unsigned short)));
            h->owned.push_back(l->wsp);
            const long long threads = (long long)nblocks * steps16 * 64;
            hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr, l->wt, l->wsp, nblocks,
                               l->K, l->Cin);
            S3_HIP_CHECK(hipGetLastError());
        }
        S3_HIP_CHECK(hipDeviceSynchronize());
    }
    h->split = on != 0;
    h->split_lin = on >= 2;
    return 0;
}

extern "C" int s3cfm_estimator(s3cfm* h, const float* x_dev, const float* mu_dev, const float* spks_dev, const float* cond_dev, float t,
                               int32_t L, float* out_dev, void* stream) {
    S3_REQUIRE(h && h->finalized, "handle not finalized");
    S3_REQUIRE(x_dev && mu_dev && spks_dev && cond_dev && out_dev && L >= 1 && L <= h->max_frames,
               "bad arguments or more frames than the workspace holds");
    S3_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    h->rows = kRows;
    h->Lk[0] = L;
    hipLaunchKernelGGL(pack_input_kernel, dim3((L + 255) / 256, kIn, kRows), dim3(256), 0, st, x_dev, mu_dev, spks_dev, cond_dev, L, 0, h->IN,
                       h->ld);
    S3_HIP_CHECK(hipGetLastError());
    if (run_estimator(h, t, L, st)) return 1;
    hipLaunchKernelGGL(unpack_output_kernel, dim3((L + 255) / 256, kMel, kRows), dim3(256), 0, st, h->D, (size_t)kMel * h->ld, h->ld, L, out_dev);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int s3cfm_solve_batch(s3cfm* h, int32_t n_utt, const int32_t* frames, int32_t L, const float* mu_dev, const float* spks_dev,
                                 const float* cond_dev, const float* noise_dev, const float* t_span, int32_t n_timesteps, double cfg_rate,
                                 float* mel_dev, void* stream) {
    S3_REQUIRE(h && h->finalized, "handle not finalized");
    S3_REQUIRE(n_utt >= 1 && n_utt <= h->max_batch, "more utterances than the workspace holds");
    S3_REQUIRE(frames && mu_dev && spks_dev && cond_dev && noise_dev && mel_dev && t_span && L >= 1 && L <= h->max_frames,
               "bad arguments or more frames than the workspace holds");
    for (int u = 0; u < n_utt; ++u) S3_REQUIRE(frames[u] >= 1 && frames[u] <= L, "an utterance's frames must lie in [1, L]");
    S3_REQUIRE(n_timesteps >= 1 && n_timesteps <= 1000, "bad number of steps");
    for (int i = 0; i <= n_timesteps; ++i) S3_REQUIRE(std::isfinite(t_span[i]), "time grid holds a value that is not finite");
    S3_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Every layer but attention works frame by frame or looks back in time only, so utterances of different lengths share
    // the launches at the longest one's L: what is computed past an utterance's end stays finite and is never read by its
    // valid frames; attention gets each utterance's own key count.
    h->rows = kRows * n_utt;
    for (int u = 0; u < n_utt; ++u) h->Lk[u] = frames[u];
    hipLaunchKernelGGL(pack_input_kernel, dim3((L + 255) / 256, kIn, h->rows), dim3(256), 0, st, noise_dev, mu_dev, spks_dev, cond_dev, L, 1,
                       h->IN, h->ld);
    S3_HIP_CHECK(hipGetLastError());
    // the guidance factors are Python doubles in the reference (1.0 + 0.7 and 0.7), rounded to fp32 when they meet a tensor
    // flow_matching.py:97-131: t and dt advance in fp32, each difference taken against the running t
    float t = t_span[0], dt = t_span[1] - t_span[0];
    for (int step = 1; step <= n_timesteps; ++step) {
        if (run_estimator(h, t, L, st)) return 1;
        hipLaunchKernelGGL(euler_step_kernel, dim3((L + 255) / 256, kMel, n_utt), dim3(256), 0, st, h->D, (size_t)kMel * h->ld, h->IN,
                           (size_t)kIn * h->ld, h->ld, L, dt, (float)(1.0 + cfg_rate), (float)cfg_rate, step == n_timesteps ? mel_dev : nullptr);
        S3_HIP_CHECK(hipGetLastError());
        t = t + dt;
        if (step < n_timesteps) dt = t_span[step + 1] - t;
    }
    return 0;
}

extern "C" int s3cfm_solve(s3cfm* h, const float* mu_dev, const float* spks_dev, const float* cond_dev, const float* noise_dev, int32_t L,
                           const float* t_span, int32_t n_timesteps, double cfg_rate, float* mel_dev, void* stream) {
    const int32_t frames = L;
    return s3cfm_solve_batch(h, 1, &frames, L, mu_dev, spks_dev, cond_dev, noise_dev, t_span, n_timesteps, cfg_rate, mel_dev, stream);
}

// ---- single operations, for tests and timing --------------------------------------------------------------------------

namespace {
struct DevBuf {
    float* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { S3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(float))); return 0; }
};
int time_launches(int iters, hipStream_t st, float* avg_us, const std::function<int()>& launch) {
    hipEvent_t e0, e1;
    S3_HIP_CHECK(hipEventCreate(&e0));
    S3_HIP_CHECK(hipEventCreate(&e1));
    int rc = launch();                                           // warm-up, and the result the caller reads
    S3_HIP_CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters && !rc; ++i) rc = launch();
    S3_HIP_CHECK(hipEventRecord(e1, st));
    S3_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    S3_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (avg_us) *avg_us = iters > 0 ? ms * 1e3f / iters : 0.f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
}  // namespace

static int op_linear(const float* x_dev, int32_t Cin, int32_t L, int32_t ld, const float* w_host, const float* bias_host, int32_t Cout,
                     int32_t K, int32_t pad, int32_t gelu, const float* res_dev, float* y_dev, float* yt_dev, int32_t yt_from,
                     const float* ln_gamma_host, const float* ln_beta_host, int32_t iters, float* avg_us, void* stream, bool split) {
    S3_REQUIRE(x_dev && w_host && bias_host && y_dev && Cin >= 64 && Cout >= 1 && K >= 1 && L >= 1 && ld >= L && iters >= 0, "bad arguments");
    S3_REQUIRE(!yt_dev || (yt_from % 32 == 0 && Cout - yt_from == kInner), "the frame-major part must be the last 512 channels");
    S3_REQUIRE(!ln_gamma_host == !ln_beta_host && (!ln_gamma_host || K == 1), "a folded LayerNorm needs gain and shift and a one-tap layer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the fold of make_lin_ln: W' = W diag(gamma), b' = b + W beta, rsum = row sums of W'
    std::vector<float> wf(w_host, w_host + (size_t)Cout * Cin * K), bf(bias_host, bias_host + Cout), rs(Cout, 0.f);
    if (ln_gamma_host) {
        for (int o = 0; o < Cout; ++o) {
            double sb = bias_host[o], sr = 0.0;
            for (int c = 0; c < Cin; ++c) {
                wf[(size_t)o * Cin + c] = w_host[(size_t)o * Cin + c] * ln_gamma_host[c];
                sr += (double)wf[(size_t)o * Cin + c];
                sb += (double)w_host[(size_t)o * Cin + c] * (double)ln_beta_host[c];
            }
            bf[o] = (float)sb;
            rs[o] = (float)sr;
        }
    }
    const std::vector<float> t = s3_tile_conv_weights(wf.data(), Cout, Cin, K);
    DevBuf wt, bias, rsum;
    if (wt.alloc(t.size()) || bias.alloc(Cout) || rsum.alloc(Cout)) return 1;
    S3_HIP_CHECK(hipMemcpy(wt.p, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    S3_HIP_CHECK(hipMemcpy(bias.p, bf.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
    S3_HIP_CHECK(hipMemcpy(rsum.p, rs.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
    LinArgs a{};
    a.x = x_dev; a.x_bs = (size_t)Cin * ld; a.x_ld = ld;
    a.wt = wt.p; a.bias = bias.p;
    a.y = y_dev; a.y_bs = (size_t)(yt_dev ? yt_from : Cout) * ld; a.y_ld = ld;
    a.res = res_dev;
    a.yt = yt_dev; a.yt_bs = (size_t)ld * kVtLd; a.yt_ld = kVtLd; a.yt_from = yt_from;
    a.Cin = Cin; a.Cout = Cout; a.CoutPad = (Cout + 31) / 32 * 32; a.K = K; a.pad = pad; a.L = L;
    a.rsum = ln_gamma_host ? rsum.p : nullptr; a.ln_eps = 1e-5f;
    DevBuf wsp;
    if (split) {
        const int nblocks = (Cout + 31) / 32, steps16 = K * (Cin / 16);
        if (wsp.alloc(((size_t)nblocks * steps16 * 3 * 512 + 1) / 2)) return 1;        // bf16 elements in a float buffer
        const long long threads = (long long)nblocks * steps16 * 64;
        hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, wt.p,
                           reinterpret_cast<unsigned short*>(wsp.p), nblocks, K, Cin);
        S3_HIP_CHECK(hipGetLastError());
        a.wsp = reinterpret_cast<unsigned short*>(wsp.p);
    }
    const int rc = time_launches(iters, st, avg_us, [&]() { return launch_lin(a, gelu ? OUT_GELU : OUT_NONE, kRows, st); });
    S3_HIP_CHECK(hipStreamSynchronize(st));
#ifdef S3_LIN_TIMELINE
    if (rc == 0 && iters > 0) {
        // one more launch with stamps: per wave start, loop start, loop end, merge done, end (shader clock) and start / end in
        // the 100 MHz wall clock; printed as means over the waves and the spread of the starts and ends
        const size_t cap = (size_t)1 << 20;                    // waves
        long long* tl = nullptr;
        S3_HIP_CHECK(hipMalloc(&tl, cap * 8 * sizeof(long long)));
        S3_HIP_CHECK(hipMemset(tl, 0, cap * 8 * sizeof(long long)));
        a.timeline = tl;
        if (launch_lin(a, gelu ? OUT_GELU : OUT_NONE, kRows, st)) return 1;
        S3_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<long long> h(cap * 8);
        S3_HIP_CHECK(hipMemcpy(h.data(), tl, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        hipFree(tl);
        double sum[4] = {0, 0, 0, 0};
        long long w0 = -1, w1 = 0, n = 0;
        std::vector<long long> starts, ends;
        for (size_t i = 0; i < cap; ++i) {
            const long long* t = &h[i * 8];
            if (!t[0]) continue;
            ++n;
            sum[0] += (double)(t[1] - t[0]); sum[1] += (double)(t[2] - t[1]); sum[2] += (double)(t[3] - t[2]); sum[3] += (double)(t[4] - t[3]);
            if (w0 < 0 || t[6] < w0) w0 = t[6];
            if (t[7] > w1) w1 = t[7];
            starts.push_back(t[6]); ends.push_back(t[7]);
        }
        std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
        auto pct = [&](const std::vector<long long>& v, double q) { return (double)(v[(size_t)(q * (v.size() - 1))] - w0) / 100.0; };
        fprintf(stderr, "  timeline: %lld waves; shader cycles per wave: prologue %.0f, loop %.0f, merge wait %.0f, epilogue %.0f; launch %.1f us; "
                        "wave starts at 10/50/90/100 %%: %.1f %.1f %.1f %.1f us; ends at 10/50/90 %%: %.1f %.1f %.1f us\n",
                n, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, (double)(w1 - w0) / 100.0, pct(starts, 0.1), pct(starts, 0.5), pct(starts, 0.9),
                pct(starts, 1.0), pct(ends, 0.1), pct(ends, 0.5), pct(ends, 0.9));
    }
#endif
    return rc;
}

extern "C" int s3cfm_op_linear(const float* x_dev, int32_t Cin, int32_t L, int32_t ld, const float* w_host, const float* bias_host, int32_t Cout,
                               int32_t K, int32_t pad, int32_t gelu, const float* res_dev, float* y_dev, float* yt_dev, int32_t yt_from,
                               const float* ln_gamma_host, const float* ln_beta_host, int32_t iters, float* avg_us, void* stream) {
    return op_linear(x_dev, Cin, L, ld, w_host, bias_host, Cout, K, pad, gelu, res_dev, y_dev, yt_dev, yt_from, ln_gamma_host, ln_beta_host,
                     iters, avg_us, stream, false);
}

extern "C" int s3cfm_op_linear_split(const float* x_dev, int32_t Cin, int32_t L, int32_t ld, const float* w_host, const float* bias_host,
                                     int32_t Cout, int32_t K, int32_t pad, int32_t gelu, const float* res_dev, float* y_dev, float* yt_dev,
                                     int32_t yt_from, const float* ln_gamma_host, const float* ln_beta_host, int32_t iters, float* avg_us,
                                     void* stream) {
    return op_linear(x_dev, Cin, L, ld, w_host, bias_host, Cout, K, pad, gelu, res_dev, y_dev, yt_dev, yt_from, ln_gamma_host, ln_beta_host,
                     iters, avg_us, stream, true);
}

extern "C" int s3cfm_op_attention(const float* qk_dev, const float* vt_dev, int32_t n_utt, const int32_t* keys, int32_t L, int32_t ld,
                                  float* o_dev, int32_t iters, float* avg_us, void* stream) {
    S3_REQUIRE(qk_dev && vt_dev && o_dev && L >= 1 && ld >= (L + 31) / 32 * 32 && iters >= 0, "bad arguments (ld must cover whole 32-frame tiles)");
    S3_REQUIRE(n_utt >= 1 && n_utt <= kMaxUtt, "bad number of utterances");
    hipStream_t st = static_cast<hipStream_t>(stream);
    AttnArgs at{qk_dev, qk_dev + (size_t)kInner * ld, (size_t)2 * kInner * ld, ld, vt_dev, (size_t)ld * kVtLd, o_dev, (size_t)kInner * ld, L,
                kRows * n_utt, {}};
    for (int u = 0; u < n_utt; ++u) {
        at.Lk[u] = keys ? keys[u] : L;
        S3_REQUIRE(at.Lk[u] >= 1 && at.Lk[u] <= L, "an utterance's keys must lie in [1, L]");
    }
    const int rc = time_launches(iters, st, avg_us, [&]() { return launch_attn(at, st); });
    S3_HIP_CHECK(hipStreamSynchronize(st));
    return rc;
}
