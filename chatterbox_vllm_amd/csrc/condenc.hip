// Conditioning encoder on the GPU (gfx950, fp32): speaker embedding + prompt-speech embeddings + emotion
// scalar -> the [34, 1024] prefix every T3 prompt starts with.  Follows the reference's T3CondEnc and
// Perceiver (src/chatterbox_vllm/models/t3/modules/cond_enc.py:57-123, perceiver.py:116-215):
//
//   row 0       spkr_enc(speaker_emb)              Linear 256 -> 1024 with bias
//   rows 1..32  perceiver(prompt embeddings)       32 learned queries cross-attend to the n prompt rows, the
//                                                  result self-attends; ONE shared block both times:
//                                                  x1 + proj_out(attn(q(LN x1), k(LN x2), v(LN x2))),
//                                                  LayerNorm eps 1e-5, biased Linears, 4 heads x 256, scale 1/16
//   row 33      emotion_adv_fc(exaggeration)       Linear 1 -> 1024, no bias
//
// The reference runs this in fp32 and hands fp32 rows to the engine, so everything here is fp32 too
// (plain FMA kernels: the whole encoder is ~2.5 GFLOP once per voice, not MFMA work).  Sums run in a
// fixed order, so a voice's prefix is the same bits on every call.
#include "t3_common.h"
#include "../../include/t3_hip.h"

#include <cmath>
#include <map>
#include <string>
#include <vector>

namespace {

constexpr int CE_D = 1024, CE_H = 4, CE_HD = 256, CE_Q = 32, CE_SPK = 256, CE_MAX_PROMPT = 1024;

__device__ __forceinline__ float ce_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// y = LayerNorm(x) * w + b over rows of 1024; one workgroup of 256 threads per row (4 elements each).
__global__ __launch_bounds__(256) void ce_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y) {
    __shared__ float part[4];
    const size_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + row * CE_D + tid * 4);
    float s = ce_wave_sum(v[0] + v[1] + v[2] + v[3]);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    const float mean = (part[0] + part[1] + part[2] + part[3]) * (1.0f / CE_D);
    __syncthreads();
    const f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
    s = ce_wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    const float var = (part[0] + part[1] + part[2] + part[3]) * (1.0f / CE_D);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + tid * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + tid * 4);
    const f32x4 o = {d[0] * rstd * wv[0] + bv[0], d[1] * rstd * wv[1] + bv[1], d[2] * rstd * wv[2] + bv[2],
                     d[3] * rstd * wv[3] + bv[3]};
    *reinterpret_cast<f32x4*>(y + row * CE_D + tid * 4) = o;
}

// Y[M,N] = X[M,K] @ W[N,K]^T (+ bias[N]) (+ R[M,N]).  Workgroup = 32 rows x 64 columns, K in steps of 32 through
// LDS; thread (ty = tid / 16, tx = tid % 16) owns rows ty*2..+1 and columns tx*4..+3.
__global__ __launch_bounds__(256) void ce_linear_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ R,
                                                        float* __restrict__ Y, int ldy, int M, int N, int K) {
    __shared__ float Xs[32][33];
    __shared__ float Ws[64][33];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 64;
    float acc[2][4] = {};
    for (int k0 = 0; k0 < K; k0 += 32) {
        {
            const int r = tid >> 3, c = (tid & 7) * 4;           // 32 rows x 32 k of X: one float4 per thread
            const int gr = m0 + r < M ? m0 + r : M - 1;
            const f32x4 v = *reinterpret_cast<const f32x4*>(X + (size_t)gr * ldx + k0 + c);
            Xs[r][c] = v[0]; Xs[r][c + 1] = v[1]; Xs[r][c + 2] = v[2]; Xs[r][c + 3] = v[3];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                            // 64 rows x 32 k of W: two float4 per thread
            const int r = (tid >> 3) + 32 * i, c = (tid & 7) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + r) * K + k0 + c);
            Ws[r][c] = v[0]; Ws[r][c + 1] = v[1]; Ws[r][c + 2] = v[2]; Ws[r][c + 3] = v[3];
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float x0 = Xs[ty * 2][k], x1 = Xs[ty * 2 + 1][k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wv = Ws[tx * 4 + j][k];
                acc[0][j] = fmaf(x0, wv, acc[0][j]);
                acc[1][j] = fmaf(x1, wv, acc[1][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = m0 + ty * 2 + i;
        if (r >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            float v = acc[i][j];
            if (bias) v += bias[n];
            if (R) v += R[(size_t)r * ldy + n];
            Y[(size_t)r * ldy + n] = v;
        }
    }
}

// out[q, h*256 + d] = sum_t softmax_t(q_h . k_h[t] / 16) * v_h[t][d]; workgroup = (query, head), 256 threads.
__global__ __launch_bounds__(256) void ce_attention_kernel(const float* __restrict__ Q, const float* __restrict__ Kx,
                                                           const float* __restrict__ V, float* __restrict__ out, int n_keys) {
    __shared__ float qs[CE_HD];
    __shared__ float p[CE_MAX_PROMPT];
    __shared__ float part[4];
    const int qi = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    qs[tid] = Q[(size_t)qi * CE_D + h * CE_HD + tid];
    __syncthreads();
    float lmax = -INFINITY;
    for (int t = tid; t < n_keys; t += 256) {
        const float* kr = Kx + (size_t)t * CE_D + h * CE_HD;
        float d = 0.f;
        for (int i = 0; i < CE_HD; i += 4) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + i);
            d = fmaf(qs[i], kv[0], d); d = fmaf(qs[i + 1], kv[1], d); d = fmaf(qs[i + 2], kv[2], d); d = fmaf(qs[i + 3], kv[3], d);
        }
        d *= 0.0625f;
        p[t] = d;
        lmax = fmaxf(lmax, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((tid & 63) == 0) part[tid >> 6] = lmax;
    __syncthreads();
    lmax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    __syncthreads();
    float ls = 0.f;
    for (int t = tid; t < n_keys; t += 256) { const float e = expf(p[t] - lmax); p[t] = e; ls += e; }
    ls = ce_wave_sum(ls);
    if ((tid & 63) == 0) part[tid >> 6] = ls;
    __syncthreads();
    const float inv = 1.0f / (part[0] + part[1] + part[2] + part[3]);
    float o = 0.f;
    for (int t = 0; t < n_keys; ++t) o = fmaf(p[t] * inv, V[(size_t)t * CE_D + h * CE_HD + tid], o);
    out[(size_t)qi * CE_D + h * CE_HD + tid] = o;
}

__global__ void ce_emotion_kernel(const float* __restrict__ w, float e, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < CE_D) out[i] = w[i] * e;
}

const std::map<std::string, std::vector<int64_t>>& ce_shapes() {
    static const std::map<std::string, std::vector<int64_t>> s = {
        {"spkr_enc.weight", {CE_D, CE_SPK}}, {"spkr_enc.bias", {CE_D}}, {"emotion_adv_fc.weight", {CE_D, 1}},
        {"perceiver.pre_attention_query", {1, CE_Q, CE_D}},
        {"perceiver.attn.norm.weight", {CE_D}}, {"perceiver.attn.norm.bias", {CE_D}},
        {"perceiver.attn.to_q.weight", {CE_D, CE_D}}, {"perceiver.attn.to_q.bias", {CE_D}},
        {"perceiver.attn.to_k.weight", {CE_D, CE_D}}, {"perceiver.attn.to_k.bias", {CE_D}},
        {"perceiver.attn.to_v.weight", {CE_D, CE_D}}, {"perceiver.attn.to_v.bias", {CE_D}},
        {"perceiver.attn.proj_out.weight", {CE_D, CE_D}}, {"perceiver.attn.proj_out.bias", {CE_D}},
    };
    return s;
}

}  // namespace

struct t3_condenc {
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::string, float*> w;
    float *spk = nullptr, *prompt = nullptr, *n1 = nullptr, *n2 = nullptr, *q = nullptr, *k = nullptr, *v = nullptr,
          *att = nullptr, *pre = nullptr, *out = nullptr;
};

namespace {

int ce_linear(t3_condenc* c, const float* X, int ldx, const char* name, const float* R, float* Y, int M, int N, int K,
              bool has_bias = true) {
    const float* W = c->w[std::string(name) + ".weight"];
    const float* b = has_bias ? c->w[std::string(name) + ".bias"] : nullptr;
    T3_LAUNCH(ce_linear_kernel, dim3(N / 64, (M + 31) / 32), dim3(256), 0, c->stream, X, ldx, W, b, R, Y, N, M, N, K);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// x1 [m1, 1024], x2 [m2, 1024] -> y = x1 + proj_out(attention(...))     (perceiver.py:159-173)
int ce_block(t3_condenc* c, const float* x1, int m1, const float* x2, int m2, float* y) {
    const float *lw = c->w["perceiver.attn.norm.weight"], *lb = c->w["perceiver.attn.norm.bias"];
    T3_LAUNCH(ce_layernorm_kernel, dim3(m1), dim3(256), 0, c->stream, x1, lw, lb, c->n1);
    const float* n2 = c->n1;
    if (x2 != x1) {
        T3_LAUNCH(ce_layernorm_kernel, dim3(m2), dim3(256), 0, c->stream, x2, lw, lb, c->n2);
        n2 = c->n2;
    }
    T3_HIP_CHECK(hipGetLastError());
    if (ce_linear(c, c->n1, CE_D, "perceiver.attn.to_q", nullptr, c->q, m1, CE_D, CE_D)) return 1;
    if (ce_linear(c, n2, CE_D, "perceiver.attn.to_k", nullptr, c->k, m2, CE_D, CE_D)) return 1;
    if (ce_linear(c, n2, CE_D, "perceiver.attn.to_v", nullptr, c->v, m2, CE_D, CE_D)) return 1;
    T3_LAUNCH(ce_attention_kernel, dim3(m1, CE_H), dim3(256), 0, c->stream, c->q, c->k, c->v, c->att, m2);
    T3_HIP_CHECK(hipGetLastError());
    return ce_linear(c, c->att, CE_D, "perceiver.attn.proj_out", x1, y, m1, CE_D, CE_D);
}

}  // namespace

extern "C" int t3_condenc_create(int32_t device, t3_condenc** out) {
    T3_REQUIRE(out, "null argument");
    T3_HIP_CHECK(hipSetDevice(device));
    t3_condenc* c = new t3_condenc();
    c->device = device;
    T3_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    auto fm = [&](float** p, size_t n) -> int { T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float))); return 0; };
    const size_t big = (size_t)CE_MAX_PROMPT * CE_D, small = (size_t)CE_Q * CE_D;
    if (fm(&c->spk, CE_SPK) || fm(&c->prompt, big) || fm(&c->n1, small) || fm(&c->n2, big) || fm(&c->q, small) ||
        fm(&c->k, big) || fm(&c->v, big) || fm(&c->att, small) || fm(&c->pre, small) || fm(&c->out, (size_t)(CE_Q + 2) * CE_D)) {
        (void)t3_condenc_destroy(c);
        return 1;
    }
    *out = c;
    return 0;
}

extern "C" int t3_condenc_load_tensor(t3_condenc* c, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    T3_REQUIRE(c && name && data && shape, "null argument");
    auto it = ce_shapes().find(name);
    if (it == ce_shapes().end()) { t3_set_error(std::string("cond encoder has no tensor named ") + name); return 1; }
    T3_REQUIRE((size_t)ndim == it->second.size(), "wrong rank for this tensor");
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { T3_REQUIRE(shape[i] == it->second[i], "wrong shape for this tensor"); n *= (size_t)shape[i]; }
    T3_HIP_CHECK(hipSetDevice(c->device));
    float*& p = c->w[name];
    if (!p) T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(float)));
    T3_HIP_CHECK(hipMemcpy(p, data, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int t3_condenc_encode(t3_condenc* c, const float* speaker_emb, const float* prompt_emb, int32_t n_prompt,
                                 float exaggeration, float* out) {
    T3_REQUIRE(c && speaker_emb && prompt_emb && out, "null argument");
    T3_REQUIRE(n_prompt >= 1 && n_prompt <= CE_MAX_PROMPT, "prompt rows must be 1..1024");
    for (const auto& kv : ce_shapes())
        if (!c->w.count(kv.first)) { t3_set_error("cond encoder tensor not loaded: " + kv.first); return 1; }
    T3_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    T3_HIP_CHECK(hipMemcpyAsync(c->spk, speaker_emb, sizeof(float) * CE_SPK, hipMemcpyHostToDevice, st));
    T3_HIP_CHECK(hipMemcpyAsync(c->prompt, prompt_emb, sizeof(float) * (size_t)n_prompt * CE_D, hipMemcpyHostToDevice, st));
    // row 0: speaker projection; rows 1..32: perceiver; row 33: emotion
    if (ce_linear(c, c->spk, CE_SPK, "spkr_enc", nullptr, c->out, 1, CE_D, CE_SPK)) return 1;
    const float* query = c->w["perceiver.pre_attention_query"];
    if (ce_block(c, query, CE_Q, c->prompt, n_prompt, c->pre)) return 1;
    if (ce_block(c, c->pre, CE_Q, c->pre, CE_Q, c->out + CE_D)) return 1;
    T3_LAUNCH(ce_emotion_kernel, dim3(CE_D / 256), dim3(256), 0, st, c->w["emotion_adv_fc.weight"], exaggeration,
              c->out + (size_t)(CE_Q + 1) * CE_D);
    T3_HIP_CHECK(hipGetLastError());
    T3_HIP_CHECK(hipMemcpyAsync(out, c->out, sizeof(float) * (size_t)(CE_Q + 2) * CE_D, hipMemcpyDeviceToHost, st));
    T3_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int t3_condenc_emotion_row(t3_condenc* c, float exaggeration, float* out) {
    T3_REQUIRE(c && out, "null argument");
    T3_REQUIRE(c->w.count("emotion_adv_fc.weight"), "cond encoder tensor not loaded: emotion_adv_fc.weight");
    T3_HIP_CHECK(hipSetDevice(c->device));
    float* row = c->out + (size_t)(CE_Q + 1) * CE_D;
    T3_LAUNCH(ce_emotion_kernel, dim3(CE_D / 256), dim3(256), 0, c->stream, c->w["emotion_adv_fc.weight"], exaggeration, row);
    T3_HIP_CHECK(hipGetLastError());
    T3_HIP_CHECK(hipMemcpyAsync(out, row, sizeof(float) * CE_D, hipMemcpyDeviceToHost, c->stream));
    T3_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int t3_condenc_destroy(t3_condenc* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->w) (void)hipFree(kv.second);
    for (float* p : {c->spk, c->prompt, c->n1, c->n2, c->q, c->k, c->v, c->att, c->pre, c->out}) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}
