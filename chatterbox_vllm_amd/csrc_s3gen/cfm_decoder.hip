// Flow-matching decoder of S3Gen's flow stage (mu -> mel) on gfx950 (C ABI: include/s3cfm_hip.h).
//
// Reference: CausalConditionalCFM.forward / ConditionalCFM.solve_euler (src/chatterbox_vllm/models/s3gen/flow_matching.py:
// 203-228, 82-133) around ConditionalDecoder.forward (decoder.py:233-317).  Every Euler step runs the estimator on two rows
// (conditioned, unconditioned) of the same x.  Activations are channel-major, [row][channel][frame] fp32, the layout of the
// vocoder's tensors: the causal convolutions and every Linear go through the matrix kernel of hift_kernels.hip (weights as
// the A operand, frames as N, a Linear being a one-tap convolution), residual adds ride in its epilogue and the GELU of the
// feed-forward is applied while its input tile is staged.  LayerNorm over the 256 channels of a frame, with the Mish and the
// time-embedding add that follow it in a resnet block, is one small kernel.  Attention (8 heads x 64) is the three-launch
// form of flow_encoder.hip: scores, row softmax, weighted sum of values on the strided fp32 matrix kernel.
#include "hift_common.h"
#include "../../include/s3cfm_hip.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

namespace {

constexpr int kMel = S3CFM_MEL, kIn = 4 * S3CFM_MEL, kC = 256, kTime = 1024, kHeads = 8, kDk = 64, kInner = 512, kFf = 1024;
constexpr int kBlocks = 4, kMid = 12, kStages = kMid + 2, kRows = 2;

// ---- kernels -----------------------------------------------------------------------------------------------------

__device__ __forceinline__ float mish(float x) {             // x * tanh(softplus(x)), softplus linear above 20 as torch's
    const float sp = x > 20.0f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

enum { LN_PLAIN = 0, LN_MISH = 1, LN_MISH_ADD = 2 };
// LayerNorm over the 256 channels of every frame of a [row][256][T] tensor (eps 1e-5): 8 frames x 32 channel groups per
// workgroup, mean, then the centred sum of squares.  LN_MISH applies Mish to the result, LN_MISH_ADD then adds add[c]
// (the time-embedding projection of a resnet block, matcha/decoder.py:58).  In place is allowed.
template <int MODE>
__global__ __launch_bounds__(256) void ln256_kernel(const float* x, float* y, int ld, size_t bs, int T, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ add) {
    __shared__ float red[32][9];
    const int tx = threadIdx.x & 7, cy = threadIdx.x >> 3;
    const int t = blockIdx.x * 8 + tx;
    const bool ok = t < T;
    const float* xb = x + blockIdx.y * bs;
    float* yb = y + blockIdx.y * bs;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ok ? xb[(size_t)(cy + 32 * i) * ld + t] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    red[cy][tx] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) mean += red[i][tx];
    mean *= 1.0f / kC;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    red[cy][tx] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) var += red[i][tx];
    const float rstd = 1.0f / sqrtf(var * (1.0f / kC) + 1e-5f);
    if (ok)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cy + 32 * i;
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

// a[z][i][:] = softmax_j(s[z][i][j] / 8), in place
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int T) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float* row = s + ((long long)blockIdx.y * T + blockIdx.x) * T;
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 256) {
        const float v = row[j] * 0.125f;
        row[j] = v;
        mx = fmaxf(mx, v);
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

// The estimator's input (decoder.py:254-260): rows 0..79 x, 80..159 mu, 160..239 the speaker vector on every frame,
// 240..319 cond.  uncond_row1: row 1 is the unconditioned call of solve_euler (flow_matching.py:105-118): the same x, zeros
// for the rest; otherwise the four inputs hold two rows each.
__global__ void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ mu, const float* __restrict__ spks,
                                  const float* __restrict__ cond, int L, int uncond_row1, float* __restrict__ in, int ld) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, r = blockIdx.z;
    if (t >= L) return;
    const int part = c / kMel, cc = c % kMel;
    const int src_row = uncond_row1 ? 0 : r;
    float v;
    if (part == 0) v = x[((size_t)src_row * kMel + cc) * L + t];
    else if (uncond_row1 && r == 1) v = 0.f;
    else if (part == 1) v = mu[((size_t)src_row * kMel + cc) * L + t];
    else if (part == 2) v = spks[src_row * kMel + cc];
    else v = cond[((size_t)src_row * kMel + cc) * L + t];
    in[((size_t)r * kIn + c) * ld + t] = v;
}

// One Euler step with classifier-free guidance (flow_matching.py:125-127): x += dt * ((1 + rate) d[0] - rate d[1]), written
// to the x rows of both estimator inputs and, when mel is not null, to the caller's [80][L] output.  Each product and sum
// is rounded on its own, as the reference's separate tensor operations are.
__global__ void euler_step_kernel(const float* __restrict__ d, size_t d_bs, float* __restrict__ in, size_t in_bs, int ld, int L, float dt,
                                  float one_plus_rate, float rate, float* __restrict__ mel) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (t >= L) return;
    const size_t o = (size_t)c * ld + t;
    const float g = __fsub_rn(__fmul_rn(one_plus_rate, d[o]), __fmul_rn(rate, d[d_bs + o]));
    const float xn = __fadd_rn(in[o], __fmul_rn(dt, g));
    in[o] = xn;
    in[in_bs + o] = xn;
    if (mel) mel[(size_t)c * L + t] = xn;
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
struct Lin { float* wt = nullptr; float* bias = nullptr; int Cin = 0, Cout = 0, K = 1; };
struct Norm { float* g = nullptr; float* b = nullptr; };
struct Resnet { Lin conv1, conv2, res; Norm n1, n2; };
struct Transformer { Lin qkv, out, w1, w2; Norm n1, n3; };
struct Stage { Resnet res; Transformer tf[kBlocks]; Lin tail; bool has_tail = false; };

}  // namespace

struct s3cfm {
    int device = 0, max_frames = 0, ld = 0;
    bool finalized = false;
    std::map<std::string, Host> raw;
    std::vector<void*> owned;
    Stage stages[kStages];
    Lin final_conv, final_proj;
    Norm final_norm;
    float *freq = nullptr, *t_w1 = nullptr, *t_b1 = nullptr, *t_w2 = nullptr, *t_b2 = nullptr, *mlp_w = nullptr, *mlp_b = nullptr;
    float *emb = nullptr, *th = nullptr, *temb = nullptr, *tvec = nullptr;
    float *IN = nullptr, *X = nullptr, *Y = nullptr, *Z = nullptr, *CAT = nullptr, *N = nullptr, *QKV = nullptr, *O = nullptr, *H = nullptr,
          *S = nullptr, *D = nullptr;
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
int make_norm(s3cfm* h, Norm* n, const std::string& name) {
    const Host *g, *b;
    if (get(h, name + ".weight", {kC}, &g) || get(h, name + ".bias", {kC}, &b)) return 1;
    return upload(h, &n->g, g->v.data(), kC) || upload(h, &n->b, b->v.data(), kC);
}

// y[row][Cout][L] = conv(act(x[row][Cin][L])) (+ res) on both rows; pad = zero frames on the left
int run_lin(const s3cfm* h, const Lin& l, const float* x, int x_rows, float* y, int y_rows, int L, int pad, int act, const float* res,
            hipStream_t st) {
    ConvArgs a{};
    a.x = x; a.x_bs = (size_t)x_rows * h->ld; a.x_ld = h->ld; a.Tin = L;
    a.wt = l.wt; a.bias = l.bias;
    a.y = y; a.y_bs = (size_t)y_rows * h->ld; a.y_ld = h->ld; a.Tout = L;
    a.res = res;
    a.Cin = l.Cin; a.Cout = l.Cout; a.CoutPad = (l.Cout + 31) / 32 * 32; a.K = l.K;
    a.dil = 1; a.pad = pad; a.act = act; a.slope = 0.f;
    a.out_stride = 1; a.out_off = 0; a.Tq = L;
    return launch_conv1d(a, kRows, st);
}

int run_ln(const s3cfm* h, int mode, const float* x, float* y, int L, const Norm& n, const float* add, hipStream_t st) {
    const dim3 grid((L + 7) / 8, kRows);
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
    if (run_lin(h, r.conv1, x, x_rows, h->Y, kC, L, 2, ACT_NONE, nullptr, st)) return 1;
    if (run_ln(h, LN_MISH_ADD, h->Y, h->Y, L, r.n1, tvec, st)) return 1;
    if (run_lin(h, r.conv2, h->Y, kC, h->Z, kC, L, 2, ACT_NONE, nullptr, st)) return 1;
    if (run_ln(h, LN_MISH, h->Z, h->Z, L, r.n2, nullptr, st)) return 1;
    return run_lin(h, r.res, x, x_rows, h->Y, kC, L, 0, ACT_NONE, h->Z, st);
}

// BasicTransformerBlock.forward (matcha/transformer.py:243-316) on X [row][256][L], in place
int run_transformer(s3cfm* h, const Transformer& b, int L, hipStream_t st) {
    const int ld = h->ld;
    if (run_ln(h, LN_PLAIN, h->X, h->N, L, b.n1, nullptr, st)) return 1;
    if (run_lin(h, b.qkv, h->N, kC, h->QKV, 3 * kInner, L, 0, ACT_NONE, nullptr, st)) return 1;
    const long long head = (long long)kDk * ld, row = 3LL * kInner * ld;
    BGemm qk{h->QKV, head, 1, ld, h->QKV + (size_t)kInner * ld, head, ld, 1, h->S, (long long)L * L, L, L, L, kDk,
             kHeads, row, row, (long long)kHeads * L * L};
    if (launch_bgemm(qk, kRows * kHeads, st)) return 1;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(L, kRows * kHeads), dim3(256), 0, st, h->S, L);
    S3_HIP_CHECK(hipGetLastError());
    BGemm av{h->QKV + (size_t)2 * kInner * ld, head, ld, 1, h->S, (long long)L * L, 1, L, h->O, head, ld, kDk, L, L,
             kHeads, row, (long long)kHeads * L * L, (long long)kInner * ld};
    if (launch_bgemm(av, kRows * kHeads, st)) return 1;
    if (run_lin(h, b.out, h->O, kInner, h->X, kC, L, 0, ACT_NONE, h->X, st)) return 1;
    if (run_ln(h, LN_PLAIN, h->X, h->N, L, b.n3, nullptr, st)) return 1;
    if (run_lin(h, b.w1, h->N, kC, h->H, kFf, L, 0, ACT_NONE, nullptr, st)) return 1;
    return run_lin(h, b.w2, h->H, kFf, h->X, kC, L, 0, ACT_GELU, h->X, st);
}

int copy_rows(const float* src, int src_rows, float* dst, int dst_rows, int ld, int L, hipStream_t st) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3((L + 255) / 256, kC, kRows), dim3(256), 0, st, src, (size_t)src_rows * ld, dst,
                       (size_t)dst_rows * ld, ld, L);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

// ConditionalDecoder.forward (decoder.py:233-317) on IN [2][320][L] at time t -> D [2][80][L]
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
            if (copy_rows(h->X, kC, h->CAT + (size_t)kC * ld, 2 * kC, ld, L, st)) return 1;
            if (run_lin(h, sg.tail, h->X, kC, h->Y, kC, L, 2, ACT_NONE, nullptr, st)) return 1;
            std::swap(h->X, h->Y);
        } else if (s == kStages - 2) {
            if (copy_rows(h->X, kC, h->CAT, 2 * kC, ld, L, st)) return 1;        // decoder.py:301
        } else if (s == kStages - 1) {
            if (run_lin(h, sg.tail, h->X, kC, h->Y, kC, L, 2, ACT_NONE, nullptr, st)) return 1;
            std::swap(h->X, h->Y);
        }
    }
    // final CausalBlock1D and 1x1 projection (decoder.py:315-316)
    if (run_lin(h, h->final_conv, h->X, kC, h->Y, kC, L, 2, ACT_NONE, nullptr, st)) return 1;
    if (run_ln(h, LN_MISH, h->Y, h->Y, L, h->final_norm, nullptr, st)) return 1;
    return run_lin(h, h->final_proj, h->Y, kC, h->D, kMel, L, 0, ACT_NONE, nullptr, st);
}

}  // namespace

extern "C" int s3cfm_create(int32_t device, int32_t max_frames, s3cfm** out) {
    S3_REQUIRE(out && max_frames >= 1 && max_frames <= S3CFM_MAX_FRAMES, "bad arguments");
    S3_HIP_CHECK(hipSetDevice(device));
    s3cfm* h = new s3cfm();
    h->device = device; h->max_frames = max_frames;
    h->ld = (max_frames + 31) / 32 * 32;
    *out = h;
    return 0;
}

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
            const std::vector<float> t = s3_tile_conv_weights(wst.data(), 3 * kInner, kC, 1);
            const std::vector<float> zero(3 * kInner, 0.f);
            tf.qkv.Cin = kC; tf.qkv.Cout = 3 * kInner; tf.qkv.K = 1;
            if (upload(h, &tf.qkv.wt, t.data(), t.size()) || upload(h, &tf.qkv.bias, zero.data(), zero.size())) return 1;
            if (make_lin(h, &tf.out, tp + "attn1.to_out.0", kC, kInner, 0, true)) return 1;
            if (make_lin(h, &tf.w1, tp + "ff.net.0.proj", kFf, kC, 0, true) || make_lin(h, &tf.w2, tp + "ff.net.2", kC, kFf, 0, true)) return 1;
            if (make_norm(h, &tf.n1, tp + "norm1") || make_norm(h, &tf.n3, tp + "norm3")) return 1;
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
    const size_t ld = h->ld, F = h->max_frames;
    if (dalloc(h, &h->emb, kIn) || dalloc(h, &h->th, kTime) || dalloc(h, &h->temb, kTime) || dalloc(h, &h->tvec, kStages * kC)) return 1;
    if (dalloc(h, &h->IN, kRows * kIn * ld) || dalloc(h, &h->CAT, kRows * 2 * kC * ld) || dalloc(h, &h->D, kRows * kMel * ld)) return 1;
    for (float** p : {&h->X, &h->Y, &h->Z, &h->N})
        if (dalloc(h, p, kRows * kC * ld)) return 1;
    if (dalloc(h, &h->QKV, kRows * 3 * kInner * ld) || dalloc(h, &h->O, kRows * kInner * ld) || dalloc(h, &h->H, kRows * kFf * ld)) return 1;
    if (dalloc(h, &h->S, (size_t)kRows * kHeads * F * F)) return 1;
    S3_HIP_CHECK(hipDeviceSynchronize());
    h->raw.clear();
    h->finalized = true;
    return 0;
}

extern "C" int s3cfm_estimator(s3cfm* h, const float* x_dev, const float* mu_dev, const float* spks_dev, const float* cond_dev, float t,
                               int32_t L, float* out_dev, void* stream) {
    S3_REQUIRE(h && h->finalized, "handle not finalized");
    S3_REQUIRE(x_dev && mu_dev && spks_dev && cond_dev && out_dev && L >= 1 && L <= h->max_frames,
               "bad arguments or more frames than the workspace holds");
    S3_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pack_input_kernel, dim3((L + 255) / 256, kIn, kRows), dim3(256), 0, st, x_dev, mu_dev, spks_dev, cond_dev, L, 0, h->IN,
                       h->ld);
    S3_HIP_CHECK(hipGetLastError());
    if (run_estimator(h, t, L, st)) return 1;
    hipLaunchKernelGGL(unpack_output_kernel, dim3((L + 255) / 256, kMel, kRows), dim3(256), 0, st, h->D, (size_t)kMel * h->ld, h->ld, L, out_dev);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int s3cfm_solve(s3cfm* h, const float* mu_dev, const float* spks_dev, const float* cond_dev, const float* noise_dev, int32_t L,
                           const float* t_span, int32_t n_timesteps, double cfg_rate, float* mel_dev, void* stream) {
    S3_REQUIRE(h && h->finalized, "handle not finalized");
    S3_REQUIRE(mu_dev && spks_dev && cond_dev && noise_dev && mel_dev && t_span && L >= 1 && L <= h->max_frames,
               "bad arguments or more frames than the workspace holds");
    S3_REQUIRE(n_timesteps >= 1 && n_timesteps <= 1000, "bad number of steps");
    for (int i = 0; i <= n_timesteps; ++i) S3_REQUIRE(std::isfinite(t_span[i]), "time grid holds a value that is not finite");
    S3_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pack_input_kernel, dim3((L + 255) / 256, kIn, kRows), dim3(256), 0, st, noise_dev, mu_dev, spks_dev, cond_dev, L, 1, h->IN,
                       h->ld);
    S3_HIP_CHECK(hipGetLastError());
    // the guidance factors are Python doubles in the reference (1.0 + 0.7 and 0.7), rounded to fp32 when they meet a tensor
    // flow_matching.py:97-131: t and dt advance in fp32, each difference taken against the running t
    float t = t_span[0], dt = t_span[1] - t_span[0];
    for (int step = 1; step <= n_timesteps; ++step) {
        if (run_estimator(h, t, L, st)) return 1;
        hipLaunchKernelGGL(euler_step_kernel, dim3((L + 255) / 256, kMel), dim3(256), 0, st, h->D, (size_t)kMel * h->ld, h->IN,
                           (size_t)kIn * h->ld, h->ld, L, dt, (float)(1.0 + cfg_rate), (float)cfg_rate, step == n_timesteps ? mel_dev : nullptr);
        S3_HIP_CHECK(hipGetLastError());
        t = t + dt;
        if (step < n_timesteps) dt = t_span[step + 1] - t;
    }
    return 0;
}
