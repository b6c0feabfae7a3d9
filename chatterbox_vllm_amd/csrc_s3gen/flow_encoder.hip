// Token -> mu half of S3Gen's flow stage on gfx950 (C ABI: include/s3flow_hip.h).
//
// Reference: CausalMaskedDiffWithXvec.inference (src/chatterbox_vllm/models/s3gen/flow.py:228-278) up to encoder_proj,
// UpsampleConformerEncoder.forward (transformer/upsample_encoder.py:256-299).  Activations are kept channel-major,
// [channel][position] fp32, the layout of the vocoder's tensors: every Linear is then a one-tap convolution for the
// small-tile matrix kernel of cfm_decoder.hip (lin_kernel: weights as the A operand, positions as N), the three real
// convolutions (pre-lookahead k = 4 and k = 3, upsampling k = 5) use the same kernel, and residual adds and the Swish /
// leaky ReLU that follow a layer ride in its epilogue.  Attention at these lengths
// (a few hundred to a few thousand positions, 8 heads x 64) is three small batched matrix products around a softmax:
//     S_ac[h][i][j] = (q_i + u_h) . k_j        S_bd[h][i][m] = (q_i + v_h) . p_m        (p = linear_pos(pos_emb), 2T - 1 rows)
//     a[h][i][:]    = softmax_j((S_ac[h][i][j] + S_bd[h][i][T - 1 - i + j]) / 8)       (the reference's rel_shift as an index)
//     o_i           = sum_j a[h][i][j] v_j
// on a strided fp32 MFMA kernel (bgemm_kernel); the pos_bias_u / pos_bias_v additions are folded into the biases of two
// q projections.  The whole stage is under 2 % of the token -> waveform arithmetic, so the kernels here are plain.
#include "hift_common.h"
#include "../../include/s3flow_hip.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

namespace {

constexpr int kD = S3FLOW_D_MODEL, kHeads = 8, kDk = 64, kFfn = 2048, kVocab = 6561, kMel = S3FLOW_MEL, kSpk = S3FLOW_SPK;
constexpr int kBlocks = 6, kUpBlocks = 4;

// ---- kernels -----------------------------------------------------------------------------------------------------

__global__ void embed_gather_kernel(const int32_t* __restrict__ tok, int T, const float* __restrict__ emb, float* __restrict__ x,
                                    int ld) {
    const int t = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    int id = tok[t];
    id = id < 0 ? 0 : (id > kVocab - 1 ? kVocab - 1 : id);
    x[(size_t)c * ld + t] = emb[(size_t)id * kD + c];
}

// LayerNorm over the 512 channels of every position of a [512][T] tensor: 8 positions x 32 channel groups per workgroup
// (16 channels per thread, so a few hundred positions already make a hundred workgroups), mean, then the centred sum of
// squares, then y = ((x - mean) * rstd * gamma + beta) * scale.  In place is allowed.
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* x, int ldx, float* y, int ldy, int T,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         float scale) {
    __shared__ float red[32][9];
    const int tx = threadIdx.x & 7, cy = threadIdx.x >> 3;
    const int t = blockIdx.x * 8 + tx;
    const bool ok = t < T;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = ok ? x[(size_t)(cy + 32 * i) * ldx + t] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    red[cy][tx] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) mean += red[i][tx];
    mean *= 1.0f / kD;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    red[cy][tx] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) var += red[i][tx];
    const float rstd = 1.0f / sqrtf(var * (1.0f / kD) + eps);
    if (ok)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = cy + 32 * i;
            y[(size_t)c * ldy + t] = ((v[i] - mean) * rstd * gamma[c] + beta[c]) * scale;
        }
}

// EspnetRelPositionalEncoding (transformer/embedding.py:236-253, 290-293) for T positions, channel-major: column m holds
// relative position T - 1 - m; even channels sin(pos * w), odd channels cos(pos * w), products and functions in fp32
__global__ void pos_emb_kernel(float* __restrict__ pe, int ld, int T, const float* __restrict__ div_term) {
    const int m = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (m >= 2 * T - 1) return;
    const float ang = (float)(T - 1 - m) * div_term[c >> 1];
    pe[(size_t)c * ld + m] = (c & 1) ? cosf(ang) : sinf(ang);
}

__global__ void upsample2_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int T2) {
    const int j = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (j < T2) y[(size_t)c * ldy + j] = x[(size_t)c * ldx + (j >> 1)];
}

// BGemm (hift_common.h) on v_mfma_f32_32x32x2_f32: 64 x 64 output tile per workgroup (one 32 x 32 accumulator per wave),
// k in steps of 16 through LDS
__global__ __launch_bounds__(256) void bgemm_kernel(BGemm g) {
    __shared__ float As[2][16][96], Bs[2][16][96];            // rows 32 banks apart modulo 64
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const float* A = g.A + (size_t)blockIdx.z * g.a_bs;
    const float* B = g.B + (size_t)blockIdx.z * g.b_bs;
    const bool a_m_fast = g.sam == 1, b_n_fast = g.sbn == 1;  // which index consecutive threads walk
    float ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int am = a_m_fast ? (e & 63) : (e >> 4), ak = a_m_fast ? (e >> 6) : (e & 15);
            ra[i] = (m0 + am < g.M && k0 + ak < g.K) ? A[(long long)(m0 + am) * g.sam + (long long)(k0 + ak) * g.sak] : 0.f;
            const int bn = b_n_fast ? (e & 63) : (e >> 4), bk = b_n_fast ? (e >> 6) : (e & 15);
            rb[i] = (n0 + bn < g.N && k0 + bk < g.K) ? B[(long long)(k0 + bk) * g.sbk + (long long)(n0 + bn) * g.sbn] : 0.f;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int am = a_m_fast ? (e & 63) : (e >> 4), ak = a_m_fast ? (e >> 6) : (e & 15);
            As[buf][ak][am] = ra[i];
            const int bn = b_n_fast ? (e & 63) : (e >> 4), bk = b_n_fast ? (e >> 6) : (e & 15);
            Bs[buf][bk][bn] = rb[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nchunk = (g.K + 15) / 16;
    load(0);
    store(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) load((ch + 1) * 16);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float a = As[buf][2 * kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[buf][2 * kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (ch + 1 < nchunk) store(buf ^ 1);
        __syncthreads();
    }
    float* C = g.C + (size_t)blockIdx.z * g.c_bs;
    const int n = n0 + wn * 32 + (lane & 31);
    if (n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (m < g.M) C[(long long)m * g.ldc + n] = acc[r];
    }
}

// a[h][i][:] = softmax_j((S_ac[h][i][j] + S_bd[h][i][T - 1 - i + j]) / 8), written over S_ac
// (transformer/attention.py:231-252 rel_shift as the index it amounts to, :325-328, :108-110)
__global__ __launch_bounds__(256) void softmax_shift_kernel(float* __restrict__ ac, const float* __restrict__ bd, int T, long long ldbd) {
    __shared__ float red[4];
    const int i = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    float* row = ac + ((long long)h * T + i) * T;
    const float* brow = bd + ((long long)h * T + i) * ldbd + (T - 1 - i);
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 256) {
        const float s = (row[j] + brow[j]) * 0.125f;
        row[j] = s;
        mx = fmaxf(mx, s);
    }
    for (int off = 32; off; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < T; j += 256) {
        const float e = expf(row[j] - mx);
        row[j] = e;
        sum += e;
    }
    for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int j = tid; j < T; j += 256) row[j] = row[j] * inv;
}

// spk = W normalize(x) + b (flow.py:244-245; F.normalize divides by max(|x|, 1e-12))
__global__ __launch_bounds__(256) void speaker_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                      float* __restrict__ out) {
    __shared__ float e[kSpk];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const float v = tid < kSpk ? x[tid] : 0.f;
    float s = v * v;
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    if (tid < kSpk) e[tid] = v / nrm;
    __syncthreads();
    if (tid < kMel) {
        float acc = 0.f;
        for (int k = 0; k < kSpk; ++k) acc = fmaf(e[k], w[tid * kSpk + k], acc);
        out[tid] = acc + b[tid];
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------

struct Host { std::vector<float> v; std::vector<int64_t> shape; };
struct Lin { float* wt = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0, K = 1; };   // a convolution for launch_lin
struct Block {
    Lin qkv;                                   // one projection with 2048 outputs: q + pos_bias_u, q + pos_bias_v, k, v
    Lin pos, out, w1, w2;
    float *g_mha = nullptr, *b_mha = nullptr, *g_ff = nullptr, *b_ff = nullptr;
};

}  // namespace

struct s3flow_enc {
    int device = 0, max_tokens = 0, ld = 0, ldp = 0;
    bool finalized = false;
    std::map<std::string, Host> raw;
    std::vector<void*> owned;
    float *emb = nullptr, *spk_w = nullptr, *spk_b = nullptr, *div_term = nullptr;
    Lin embed, up_embed, pre1, pre2, up_conv, proj;
    float *g_embed = nullptr, *b_embed = nullptr, *g_up = nullptr, *b_up = nullptr, *g_after = nullptr, *b_after = nullptr;
    Block blocks[kBlocks + kUpBlocks];
    float *X = nullptr, *N = nullptr, *QU = nullptr, *QV = nullptr, *Kb = nullptr, *Vb = nullptr, *O = nullptr, *HID = nullptr,
          *H = nullptr, *U = nullptr, *PE = nullptr, *P = nullptr, *Sac = nullptr, *Sbd = nullptr;
    int last_T = 0;
};

namespace {

int dalloc(s3flow_enc* h, float** p, size_t n) {
    S3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float)));
    h->owned.push_back(*p);
    return 0;
}
int upload(s3flow_enc* h, float** dst, const float* src, size_t n) {
    if (dalloc(h, dst, n)) return 1;
    S3_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
int get(s3flow_enc* h, const std::string& name, std::initializer_list<int64_t> shape, const Host** out) {
    auto it = h->raw.find(name);
    if (it == h->raw.end()) { s3_set_error("missing tensor: " + name); return 1; }
    if (it->second.shape != std::vector<int64_t>(shape)) { s3_set_error("bad shape for tensor: " + name); return 1; }
    *out = &it->second;
    return 0;
}
// Linear [Cout, Cin] or Conv1d [Cout, Cin, K] -> tile order; no bias tensor = zeros
int make_lin(s3flow_enc* h, Lin* l, const std::string& name, int Cout, int Cin, int K, bool has_bias) {
    const Host* w;
    if (K == 1 ? get(h, name + ".weight", {Cout, Cin}, &w) : get(h, name + ".weight", {Cout, Cin, K}, &w)) return 1;
    l->Cin = Cin; l->Cout = Cout; l->K = K;
    const std::vector<float> t = s3_tile_conv_weights(w->v.data(), Cout, Cin, K);
    if (upload(h, &l->wt, t.data(), t.size())) return 1;
    std::vector<float> b(Cout, 0.f);
    if (has_bias) {
        const Host* bt;
        if (get(h, name + ".bias", {Cout}, &bt)) return 1;
        b = bt->v;
    }
    return upload(h, &l->bias, b.data(), Cout);
}
int vec(s3flow_enc* h, float** dst, const std::string& name, int n) {
    const Host* t;
    if (get(h, name, {n}, &t)) return 1;
    return upload(h, dst, t->v.data(), n);
}

// y[Cout][L] = out_act(conv(x[Cin][L])) (+ res), one utterance, on the small-tile matrix kernel (hift_common.h: LinArgs);
// pad = zero positions on the left, the taps beyond them look ahead
int run_lin(const Lin& l, const float* x, int x_ld, int L, float* y, int y_ld, int pad, int out_act, const float* res, hipStream_t st) {
    LinArgs a{};
    a.x = x; a.x_bs = 0; a.x_ld = x_ld;
    a.wt = l.wt; a.bias = l.bias;
    a.y = y; a.y_bs = 0; a.y_ld = y_ld;
    a.res = res;
    a.Cin = l.Cin; a.Cout = l.Cout; a.CoutPad = (l.Cout + 31) / 32 * 32; a.K = l.K; a.pad = pad; a.L = L;
    return launch_lin(a, out_act, 1, st);
}

int run_ln(const float* x, int ldx, float* y, int ldy, int T, const float* g, const float* b, float eps, float scale, hipStream_t st) {
    hipLaunchKernelGGL(layer_norm_kernel, dim3((T + 7) / 8), dim3(256), 0, st, x, ldx, y, ldy, T, g, b, eps, scale);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int run_bgemm(const BGemm& g, int batch, hipStream_t st) { return launch_bgemm(g, batch, st); }

// ConformerEncoderLayer.forward (transformer/encoder_layer.py:191-235) on X [512][T]
int run_block(s3flow_enc* h, const Block& b, int T, hipStream_t st) {
    const int ld = h->ld, ldp = h->ldp, P2 = 2 * T - 1;
    if (run_ln(h->X, ld, h->N, ld, T, b.g_mha, b.b_mha, 1e-12f, 1.0f, st)) return 1;
    if (run_lin(b.qkv, h->N, ld, T, h->QU, ld, 0, OUT_NONE, nullptr, st)) return 1;   // QU, QV, Kb, Vb are one buffer
    if (run_lin(b.pos, h->PE, ldp, P2, h->P, ldp, 0, OUT_NONE, nullptr, st)) return 1;
    const long long head = (long long)kDk * ld, headp = (long long)kDk * ldp;
    BGemm ac{h->QU, head, 1, ld, h->Kb, head, ld, 1, h->Sac, (long long)T * T, T, T, T, kDk};
    if (run_bgemm(ac, kHeads, st)) return 1;
    BGemm bd{h->QV, head, 1, ld, h->P, headp, ldp, 1, h->Sbd, (long long)T * P2, P2, T, P2, kDk};
    if (run_bgemm(bd, kHeads, st)) return 1;
    hipLaunchKernelGGL(softmax_shift_kernel, dim3(T, kHeads), dim3(256), 0, st, h->Sac, h->Sbd, T, (long long)P2);
    S3_HIP_CHECK(hipGetLastError());
    BGemm av{h->Vb, head, ld, 1, h->Sac, (long long)T * T, 1, T, h->O, head, ld, kDk, T, T};
    if (run_bgemm(av, kHeads, st)) return 1;
    if (run_lin(b.out, h->O, ld, T, h->X, ld, 0, OUT_NONE, h->X, st)) return 1;
    if (run_ln(h->X, ld, h->N, ld, T, b.g_ff, b.b_ff, 1e-12f, 1.0f, st)) return 1;
    if (run_lin(b.w1, h->N, ld, T, h->H, ld, 0, OUT_SWISH, nullptr, st)) return 1;
    return run_lin(b.w2, h->H, ld, T, h->X, ld, 0, OUT_NONE, h->X, st);
}

int run_pos_emb(s3flow_enc* h, int T, hipStream_t st) {
    hipLaunchKernelGGL(pos_emb_kernel, dim3((2 * T - 1 + 255) / 256, kD), dim3(256), 0, st, h->PE, h->ldp, T, h->div_term);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int launch_bgemm(const BGemm& g, int batch, hipStream_t st) {
    hipLaunchKernelGGL(bgemm_kernel, dim3((g.M + 63) / 64, (g.N + 63) / 64, batch), dim3(256), 0, st, g);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int s3flow_enc_create(int32_t device, int32_t max_tokens, s3flow_enc** out) {
    S3_REQUIRE(out && max_tokens >= 1 && max_tokens <= 8192, "bad arguments");
    S3_HIP_CHECK(hipSetDevice(device));
    s3flow_enc* h = new s3flow_enc();
    h->device = device; h->max_tokens = max_tokens;
    h->ld = (2 * max_tokens + 31) / 32 * 32;
    h->ldp = (4 * max_tokens + 31) / 32 * 32;
    *out = h;
    return 0;
}

extern "C" int s3flow_enc_destroy(s3flow_enc* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->owned) (void)hipFree(p);
    delete h;
    return 0;
}

extern "C" int s3flow_enc_load_tensor(s3flow_enc* h, const char* name, const float* host_data, const int64_t* shape, int32_t ndim) {
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

extern "C" int s3flow_enc_finalize(s3flow_enc* h) {
    S3_REQUIRE(h && !h->finalized, "bad handle");
    S3_HIP_CHECK(hipSetDevice(h->device));
    const Host* t;
    if (get(h, "input_embedding.weight", {kVocab, kD}, &t) || upload(h, &h->emb, t->v.data(), t->v.size())) return 1;
    if (get(h, "spk_embed_affine_layer.weight", {kMel, kSpk}, &t) || upload(h, &h->spk_w, t->v.data(), t->v.size())) return 1;
    if (vec(h, &h->spk_b, "spk_embed_affine_layer.bias", kMel)) return 1;
    if (make_lin(h, &h->embed, "encoder.embed.out.0", kD, kD, 1, true)) return 1;
    if (make_lin(h, &h->up_embed, "encoder.up_embed.out.0", kD, kD, 1, true)) return 1;
    if (vec(h, &h->g_embed, "encoder.embed.out.1.weight", kD) || vec(h, &h->b_embed, "encoder.embed.out.1.bias", kD)) return 1;
    if (vec(h, &h->g_up, "encoder.up_embed.out.1.weight", kD) || vec(h, &h->b_up, "encoder.up_embed.out.1.bias", kD)) return 1;
    if (vec(h, &h->g_after, "encoder.after_norm.weight", kD) || vec(h, &h->b_after, "encoder.after_norm.bias", kD)) return 1;
    if (make_lin(h, &h->pre1, "encoder.pre_lookahead_layer.conv1", kD, kD, 4, true)) return 1;
    if (make_lin(h, &h->pre2, "encoder.pre_lookahead_layer.conv2", kD, kD, 3, true)) return 1;
    if (make_lin(h, &h->up_conv, "encoder.up_layer.conv", kD, kD, 5, true)) return 1;
    if (make_lin(h, &h->proj, "encoder_proj", kMel, kD, 1, true)) return 1;
    for (int i = 0; i < kBlocks + kUpBlocks; ++i) {
        const std::string p = i < kBlocks ? "encoder.encoders." + std::to_string(i) + "." : "encoder.up_encoders." + std::to_string(i - kBlocks) + ".";
        Block& b = h->blocks[i];
        const Host *u, *v;
        if (get(h, p + "self_attn.pos_bias_u", {kHeads, kDk}, &u) || get(h, p + "self_attn.pos_bias_v", {kHeads, kDk}, &v)) return 1;
        {
            // rows 0..511 linear_q with pos_bias_u added to the bias, 512..1023 linear_q with pos_bias_v, then linear_k, linear_v
            const Host *wq, *bq, *wk, *bk, *wv, *bv;
            if (get(h, p + "self_attn.linear_q.weight", {kD, kD}, &wq) || get(h, p + "self_attn.linear_q.bias", {kD}, &bq)) return 1;
            if (get(h, p + "self_attn.linear_k.weight", {kD, kD}, &wk) || get(h, p + "self_attn.linear_k.bias", {kD}, &bk)) return 1;
            if (get(h, p + "self_attn.linear_v.weight", {kD, kD}, &wv) || get(h, p + "self_attn.linear_v.bias", {kD}, &bv)) return 1;
            std::vector<float> wst((size_t)4 * kD * kD), bst(4 * kD);
            for (int part = 0; part < 4; ++part) {
                const Host* ws = part < 2 ? wq : (part == 2 ? wk : wv);
                const Host* bs = part < 2 ? bq : (part == 2 ? bk : bv);
                std::copy(ws->v.begin(), ws->v.end(), wst.begin() + (size_t)part * kD * kD);
                for (int i = 0; i < kD; ++i)
                    bst[part * kD + i] = bs->v[i] + (part == 0 ? u->v[i] : (part == 1 ? v->v[i] : 0.f));
            }
            const std::vector<float> t = s3_tile_conv_weights(wst.data(), 4 * kD, kD, 1);
            b.qkv.Cin = kD; b.qkv.Cout = 4 * kD; b.qkv.K = 1;
            if (upload(h, &b.qkv.wt, t.data(), t.size()) || upload(h, &b.qkv.bias, bst.data(), bst.size())) return 1;
        }
        if (make_lin(h, &b.pos, p + "self_attn.linear_pos", kD, kD, 1, false)) return 1;
        if (make_lin(h, &b.out, p + "self_attn.linear_out", kD, kD, 1, true)) return 1;
        if (make_lin(h, &b.w1, p + "feed_forward.w_1", kFfn, kD, 1, true)) return 1;
        if (make_lin(h, &b.w2, p + "feed_forward.w_2", kD, kFfn, 1, true)) return 1;
        if (vec(h, &b.g_mha, p + "norm_mha.weight", kD) || vec(h, &b.b_mha, p + "norm_mha.bias", kD)) return 1;
        if (vec(h, &b.g_ff, p + "norm_ff.weight", kD) || vec(h, &b.b_ff, p + "norm_ff.bias", kD)) return 1;
    }
    // w_i = exp(2i * -(ln 10000 / 512)) in fp32, as transformer/embedding.py:239-242 computes it
    std::vector<float> div(kD / 2);
    const float step = (float)(-(std::log(10000.0) / kD));
    for (int i = 0; i < kD / 2; ++i) div[i] = std::exp((float)(2 * i) * step);
    if (upload(h, &h->div_term, div.data(), div.size())) return 1;
    const size_t Tm = 2 * (size_t)h->max_tokens, act = (size_t)kD * h->ld;
    for (float** p : {&h->X, &h->N, &h->O, &h->HID, &h->U})
        if (dalloc(h, p, act)) return 1;
    if (dalloc(h, &h->QU, 4 * act)) return 1;               // the stacked projection's output: q + u, q + v, k, v
    h->QV = h->QU + act; h->Kb = h->QU + 2 * act; h->Vb = h->QU + 3 * act;
    if (dalloc(h, &h->H, (size_t)kFfn * h->ld)) return 1;
    if (dalloc(h, &h->PE, (size_t)kD * h->ldp) || dalloc(h, &h->P, (size_t)kD * h->ldp)) return 1;
    if (dalloc(h, &h->Sac, (size_t)kHeads * Tm * Tm) || dalloc(h, &h->Sbd, (size_t)kHeads * Tm * (2 * Tm))) return 1;
    S3_HIP_CHECK(hipDeviceSynchronize());
    h->raw.clear();
    h->finalized = true;
    return 0;
}

extern "C" int s3flow_enc_forward(s3flow_enc* h, const int32_t* tokens_dev, int32_t T, float* mu_dev, void* stream) {
    S3_REQUIRE(h && h->finalized, "handle not finalized");
    S3_REQUIRE(tokens_dev && mu_dev && T >= 1 && T <= h->max_tokens, "bad arguments or more tokens than the workspace holds");
    S3_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ld = h->ld, T2 = 2 * T;
    const float xscale = std::sqrt((float)kD);
    hipLaunchKernelGGL(embed_gather_kernel, dim3((T + 63) / 64, kD / 4), dim3(256), 0, st, tokens_dev, T, h->emb, h->N, ld);
    S3_HIP_CHECK(hipGetLastError());
    // LinearNoSubsampling + positional scaling (subsampling.py:111-113, embedding.py:267-270)
    if (run_lin(h->embed, h->N, ld, T, h->X, ld, 0, OUT_NONE, nullptr, st)) return 1;
    if (run_ln(h->X, ld, h->X, ld, T, h->g_embed, h->b_embed, 1e-5f, xscale, st)) return 1;
    if (run_pos_emb(h, T, st)) return 1;
    // PreLookaheadLayer (upsample_encoder.py:82-99): k = 4 looking 3 positions ahead, leaky ReLU, k = 3 looking back, residual
    if (run_lin(h->pre1, h->X, ld, T, h->N, ld, 0, OUT_LRELU, nullptr, st)) return 1;
    if (run_lin(h->pre2, h->N, ld, T, h->X, ld, 2, OUT_NONE, h->X, st)) return 1;
    for (int i = 0; i < kBlocks; ++i)
        if (run_block(h, h->blocks[i], T, st)) return 1;
    // Upsample1D (upsample_encoder.py:58-63): nearest x2, k = 5 over the 4 previous positions and the current one
    hipLaunchKernelGGL(upsample2_kernel, dim3((T2 + 255) / 256, kD), dim3(256), 0, st, h->X, ld, h->U, ld, T2);
    S3_HIP_CHECK(hipGetLastError());
    if (run_lin(h->up_conv, h->U, ld, T2, h->N, ld, 4, OUT_NONE, nullptr, st)) return 1;
    if (run_lin(h->up_embed, h->N, ld, T2, h->X, ld, 0, OUT_NONE, nullptr, st)) return 1;
    if (run_ln(h->X, ld, h->X, ld, T2, h->g_up, h->b_up, 1e-5f, xscale, st)) return 1;
    if (run_pos_emb(h, T2, st)) return 1;
    for (int i = 0; i < kUpBlocks; ++i)
        if (run_block(h, h->blocks[kBlocks + i], T2, st)) return 1;
    if (run_ln(h->X, ld, h->HID, ld, T2, h->g_after, h->b_after, 1e-5f, 1.0f, st)) return 1;
    h->last_T = T;
    return run_lin(h->proj, h->HID, ld, T2, mu_dev, T2, 0, OUT_NONE, nullptr, st);
}

extern "C" int s3flow_enc_speaker(s3flow_enc* h, const float* xvec_dev, float* spk_dev, void* stream) {
    S3_REQUIRE(h && h->finalized && xvec_dev && spk_dev, "bad arguments");
    S3_HIP_CHECK(hipSetDevice(h->device));
    hipLaunchKernelGGL(speaker_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), xvec_dev, h->spk_w, h->spk_b, spk_dev);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int s3flow_enc_last_hidden(s3flow_enc* h, float* out_dev, int32_t T, void* stream) {
    S3_REQUIRE(h && h->finalized && out_dev && T == h->last_T, "no forward call of that length to read from");
    S3_HIP_CHECK(hipMemcpy2DAsync(out_dev, (size_t)2 * T * sizeof(float), h->HID, (size_t)h->ld * sizeof(float), (size_t)2 * T * sizeof(float),
                                  kD, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return 0;
}
