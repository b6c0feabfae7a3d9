// libs3hift.so host side: the handle (weights in tile order, workspace), the launch sequence of
// HiFTGenerator.inference (reference: src/chatterbox_vllm/models/s3gen/hifigan.py:412-474), and the C ABI.
#include "hift_common.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <vector>

namespace {

thread_local std::string g_error;

constexpr int kMel = 80, kBase = 512, kF0C = 512, kBins = 18;
constexpr int kUpRate[3] = {8, 5, 3}, kUpKernel[3] = {16, 11, 7};
constexpr int kResKernel[3] = {3, 7, 11}, kDil[3] = {1, 3, 5}, kSrcResKernel[3] = {7, 7, 11};
constexpr int kDownK[3] = {30, 6, 1}, kDownStride[3] = {15, 3, 1}, kDownPad[3] = {7, 1, 0};

struct Host { std::vector<float> v; std::vector<int64_t> shape; };

struct Conv {                        // one convolution in tile order on the device
    float* wt = nullptr;
    float* bias = nullptr;
    int Cin = 0, Cout = 0, CoutPad = 0, K = 0;
};

struct ResBlock { Conv c1[3], c2[3]; float* a1[3] = {}; float* a2[3] = {}; int k = 0; };

struct Stage {                       // one upsampling stage (hifigan.py:417-435)
    Conv up;                         // the transposed convolution as a plain one: output phases stacked as channels
    int up_taps = 0;
    Conv down;                       // source_downs.i after space-to-depth: two taps over stride x 18 channels
    int down_cpad = 0;
    float* SD = nullptr;             // its input: [down_cpad][L + 1]
    ResBlock src, res[3];
    int C = 0, L = 0, ld = 0;        // channels, length and row stride of this stage's tensors for the current call
    float *A = nullptr, *S0 = nullptr, *X = nullptr, *T1 = nullptr, *U0 = nullptr, *U1 = nullptr, *XS = nullptr;
};

}  // namespace

struct s3hift {
    int device = 0, max_batch = 0, max_frames = 0;
    bool finalized = false;
    std::map<std::string, Host> raw;
    std::vector<void*> owned;        // every device allocation, freed by destroy
    Conv conv_pre, conv_post, f0c[5];
    float *f0_w = nullptr, *f0_b = nullptr, *lin_w = nullptr, *lin_b = nullptr;
    Stage st[3];
    float *P = nullptr, *SST = nullptr, *CP = nullptr, *F0A = nullptr, *F0B = nullptr, *f0_tmp = nullptr;
    double* prefix = nullptr;
    bool timing = false;
    hipEvent_t ev[7] = {};
    float last_ms[6] = {};
};

void s3_set_error(const std::string& msg) { g_error = msg; }

namespace {

int dalloc(s3hift* h, void** p, size_t bytes) {
    S3_HIP_CHECK(hipMalloc(p, bytes));
    h->owned.push_back(*p);
    return 0;
}

int upload(s3hift* h, float** dst, const float* src, size_t n) {
    if (dalloc(h, reinterpret_cast<void**>(dst), n * sizeof(float))) return 1;
    S3_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// W[Cout][Cin][K] (host, row-major) -> tile order on the device
int make_conv_from(s3hift* h, Conv* c, const float* w, const float* bias, int Cout, int Cin, int K, int n_bias = 0) {
    S3_REQUIRE(Cin % CONV_CK == 0, "convolution input channels must be a multiple of 16");
    c->Cin = Cin; c->Cout = Cout; c->K = K; c->CoutPad = (Cout + 31) / 32 * 32;
    const std::vector<float> t = s3_tile_conv_weights(w, Cout, Cin, K);
    if (upload(h, &c->wt, t.data(), t.size())) return 1;
    return upload(h, &c->bias, bias, n_bias ? n_bias : Cout);
}

int get(s3hift* h, const std::string& name, std::initializer_list<int64_t> shape, const Host** out) {
    auto it = h->raw.find(name);
    if (it == h->raw.end()) { s3_set_error("missing tensor: " + name); return 1; }
    if (it->second.shape != std::vector<int64_t>(shape)) { s3_set_error("bad shape for tensor: " + name); return 1; }
    *out = &it->second;
    return 0;
}

int make_conv(s3hift* h, Conv* c, const std::string& name, int Cout, int Cin, int K) {
    const Host *w, *b;
    if (get(h, name + ".weight", {Cout, Cin, K}, &w) || get(h, name + ".bias", {Cout}, &b)) return 1;
    return make_conv_from(h, c, w->v.data(), b->v.data(), Cout, Cin, K);
}

int make_resblock(s3hift* h, ResBlock* rb, const std::string& prefix, int ch, int k) {
    rb->k = k;
    for (int i = 0; i < 3; ++i) {
        const std::string n = std::to_string(i);
        if (make_conv(h, &rb->c1[i], prefix + ".convs1." + n, ch, ch, k)) return 1;
        if (make_conv(h, &rb->c2[i], prefix + ".convs2." + n, ch, ch, k)) return 1;
        const Host *a1, *a2;
        if (get(h, prefix + ".activations1." + n + ".alpha", {ch}, &a1)) return 1;
        if (get(h, prefix + ".activations2." + n + ".alpha", {ch}, &a2)) return 1;
        if (upload(h, &rb->a1[i], a1->v.data(), ch) || upload(h, &rb->a2[i], a2->v.data(), ch)) return 1;
    }
    return 0;
}

ConvArgs conv_args(const Conv& c, const float* x, size_t x_bs, int x_ld, int Tin, float* y, size_t y_bs, int y_ld, int Tout) {
    ConvArgs a{};
    a.x = x; a.x_bs = x_bs; a.x_ld = x_ld; a.Tin = Tin;
    a.wt = c.wt; a.bias = c.bias;
    a.y = y; a.y_bs = y_bs; a.y_ld = y_ld; a.Tout = Tout;
    a.Cin = c.Cin; a.Cout = c.Cout; a.CoutPad = c.CoutPad; a.K = c.K;
    a.dil = 1; a.pad = 0; a.act = ACT_NONE; a.slope = 0.f;
    a.out_stride = 1; a.out_off = 0; a.Tq = Tout;
    return a;
}

// ResBlock.forward (hifigan.py:154-161) on [C][L] tensors of stage s.  The last convolution adds `res2` (the upsampled
// tensor, for the source block), stores to `final_y` (may be null) and folds into the stage's running mean per `acc_mode`.
int run_resblock(s3hift* h, Stage& s, const ResBlock& rb, const float* in, float* final_y, const float* res2, int acc_mode,
                 int B, hipStream_t st) {
    const size_t bs = (size_t)s.C * s.ld;
    const float* cur = in;
    // 64-channel tensors can take both convolutions of a step in one launch, the tensor between them kept on the chip (same
    // bits).  Built to cut three of five tensor passes over HBM; measured slower (stage 2 of 8 x 500 frames: 8.05 against
    // 7.23 ms: 3 workgroups per CU instead of 5, 7 % more matrix work, a Snake pass nothing overlaps), so off unless
    // S3HIFT_FUSED=1.
    const char* fused_env = getenv("S3HIFT_FUSED");
    const bool fused = fused_env && atoi(fused_env) == 1 && s.C == 64;
    for (int i = 0; i < 3; ++i) {
        const bool last = i == 2;
        float* out = last ? final_y : (i == 0 ? s.U0 : s.U1);
        if (fused) {
            ResStepArgs r{};
            r.x = cur;
            r.w1 = rb.c1[i].wt; r.b1 = rb.c1[i].bias; r.alpha1 = rb.a1[i];
            r.w2 = rb.c2[i].wt; r.b2 = rb.c2[i].bias; r.alpha2 = rb.a2[i];
            r.y = out;
            if (last) { r.res2 = res2; r.acc = s.XS; r.acc_mode = acc_mode; }
            r.bs = bs; r.ld = s.ld; r.L = s.L; r.K = rb.k; r.dil = kDil[i];
            if (launch_resblock_step64(r, B, st)) return 1;
            cur = out;
            continue;
        }
        ConvArgs a = conv_args(rb.c1[i], cur, bs, s.ld, s.L, s.T1, bs, s.ld, s.L);
        a.act = ACT_SNAKE; a.alpha = rb.a1[i]; a.dil = kDil[i]; a.pad = (rb.k * kDil[i] - kDil[i]) / 2;
        if (launch_conv1d(a, B, st)) return 1;
        ConvArgs c = conv_args(rb.c2[i], s.T1, bs, s.ld, s.L, out, bs, s.ld, s.L);
        c.act = ACT_SNAKE; c.alpha = rb.a2[i]; c.pad = (rb.k - 1) / 2;
        c.res = cur;
        if (last) { c.res2 = res2; c.acc = s.XS; c.acc_mode = acc_mode; }
        if (launch_conv1d(c, B, st)) return 1;
        cur = out;
    }
    return 0;
}

int check_call(s3hift* h, int B, int T) {
    S3_REQUIRE(h && h->finalized, "handle not finalized");
    S3_REQUIRE(B >= 1 && B <= h->max_batch && T >= 1 && T <= h->max_frames, "batch or frame count outside the workspace");
    S3_HIP_CHECK(hipSetDevice(h->device));
    return 0;
}

int run_f0(s3hift* h, const float* mel, int B, int T, float* f0, hipStream_t st) {
    const int ld = h->max_frames;
    const size_t bs = (size_t)kF0C * ld;
    const float* cur = mel;
    size_t cur_bs = (size_t)kMel * T;
    int cur_ld = T;
    for (int i = 0; i < 5; ++i) {
        float* out = (i & 1) ? h->F0B : h->F0A;
        ConvArgs a = conv_args(h->f0c[i], cur, cur_bs, cur_ld, T, out, bs, ld, T);
        a.pad = 1;
        a.act = i == 0 ? ACT_NONE : ACT_ELU;
        if (launch_conv1d(a, B, st)) return 1;
        cur = out; cur_bs = bs; cur_ld = ld;
    }
    return launch_f0_head(cur, ld, bs, kF0C, T, h->f0_w, h->f0_b, f0, B, st);
}

int run_decode(s3hift* h, const float* mel, const float* src, int B, int T, float* wav, hipStream_t st) {
    const int Lw = T * S3HIFT_SAMPLES_PER_FRAME, M = Lw / 4 + 1;
    const int sst_ld = h->max_frames * 120 + 32;
    const size_t sst_bs = (size_t)kBins * sst_ld;
    if (h->timing) S3_HIP_CHECK(hipEventRecord(h->ev[0], st));
    if (launch_stft(src, Lw, h->SST, M, sst_ld, sst_bs, B, st)) return 1;
    {
        ConvArgs a = conv_args(h->conv_pre, mel, (size_t)kMel * T, T, T, h->P, (size_t)kBase * h->max_frames, h->max_frames, T);
        a.pad = 3;
        if (launch_conv1d(a, B, st)) return 1;
    }
    if (h->timing) S3_HIP_CHECK(hipEventRecord(h->ev[1], st));
    const float* in = h->P;
    int Lin = T, in_ld = h->max_frames;
    size_t in_bs = (size_t)kBase * h->max_frames;
    for (int i = 0; i < 3; ++i) {
        Stage& s = h->st[i];
        const int u = kUpRate[i], kw = kUpKernel[i], p = (kw - u) / 2;
        const bool last = i == 2;
        const int Lup = (Lin - 1) * u - 2 * p + kw;
        s.L = Lup + (last ? 1 : 0);
        const size_t bs = (size_t)s.C * s.ld;
        {   // output column t = q u + phase - p takes input positions q - j through taps phase + j u: a plain convolution
            // over q with up_taps taps, the phases stacked as output channels
            ConvArgs a = conv_args(s.up, in, in_bs, in_ld, Lin, s.A, bs, s.ld, s.L);
            a.act = ACT_LRELU; a.slope = 0.1f;
            a.pad = s.up_taps - 1;
            a.out_stride = u; a.out_off = -p + (last ? 1 : 0);
            a.phase_cout = s.C;
            a.Tq = (Lup - 1 + p) / u + 1;
            if (launch_conv1d(a, B, st)) return 1;
        }
        if (last && launch_reflect_left(s.A, s.ld, B * s.C, st)) return 1;      // batch stride = C * ld: rows are contiguous
        {
            const int sd = kDownStride[i];
            const size_t sd_bs = (size_t)s.down_cpad * s.ld;
            const int Tq = s.L + s.down.K - 1;
            if (launch_space_to_depth(h->SST, sst_ld, sst_bs, M, kBins, s.SD, s.ld, sd_bs, Tq, s.down_cpad, sd, kDownPad[i], B, st))
                return 1;
            ConvArgs a = conv_args(s.down, s.SD, sd_bs, s.ld, Tq, s.S0, bs, s.ld, s.L);
            if (launch_conv1d(a, B, st)) return 1;
        }
        if (run_resblock(h, s, s.src, s.S0, s.X, s.A, ACC_NONE, B, st)) return 1;
        for (int j = 0; j < 3; ++j)
            if (run_resblock(h, s, s.res[j], s.X, nullptr, nullptr, j == 0 ? ACC_SET : (j == 1 ? ACC_ADD : ACC_ADD_DIV3), B, st))
                return 1;
        in = s.XS; Lin = s.L; in_ld = s.ld; in_bs = bs;
        if (h->timing) S3_HIP_CHECK(hipEventRecord(h->ev[2 + i], st));
    }
    S3_REQUIRE(Lin == M, "stage lengths do not meet the STFT frame count");
    {
        ConvArgs a = conv_args(h->conv_post, in, in_bs, in_ld, Lin, h->CP, sst_bs, sst_ld, M);
        a.act = ACT_LRELU; a.slope = 0.01f; a.pad = 3;
        if (launch_conv1d(a, B, st)) return 1;
    }
    if (launch_istft(h->CP, sst_ld, sst_bs, M, wav, Lw, B, st)) return 1;
    if (h->timing) {
        S3_HIP_CHECK(hipEventRecord(h->ev[5], st));
        S3_HIP_CHECK(hipEventSynchronize(h->ev[5]));
        for (int i = 0; i < 5; ++i) S3_HIP_CHECK(hipEventElapsedTime(&h->last_ms[i], h->ev[i], h->ev[i + 1]));
        S3_HIP_CHECK(hipEventElapsedTime(&h->last_ms[5], h->ev[0], h->ev[5]));
    }
    return 0;
}

}  // namespace

extern "C" int s3hift_abi_version(void) { return S3HIFT_ABI_VERSION; }
extern "C" const char* s3hift_last_error(void) { return g_error.c_str(); }

extern "C" int s3hift_create(int32_t device, int32_t max_batch, int32_t max_frames, s3hift** out) {
    S3_REQUIRE(out && max_batch >= 1 && max_frames >= 1, "bad arguments");
    S3_REQUIRE((int64_t)max_batch * max_frames <= (1 << 20), "workspace above 2^20 frames");
    S3_HIP_CHECK(hipSetDevice(device));
    s3hift* h = new s3hift();
    h->device = device; h->max_batch = max_batch; h->max_frames = max_frames;
    *out = h;
    return 0;
}

extern "C" int s3hift_destroy(s3hift* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->owned) (void)hipFree(p);
    for (hipEvent_t e : h->ev) if (e) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

extern "C" int s3hift_load_tensor(s3hift* h, const char* name, const float* host_data, const int64_t* shape, int32_t ndim) {
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

extern "C" int s3hift_finalize(s3hift* h) {
    S3_REQUIRE(h && !h->finalized, "bad handle");
    S3_HIP_CHECK(hipSetDevice(h->device));
    if (make_conv(h, &h->conv_pre, "conv_pre", kBase, kMel, 7)) return 1;
    if (make_conv(h, &h->conv_post, "conv_post", kBins, kBase >> 3, 7)) return 1;
    for (int i = 0; i < 5; ++i)
        if (make_conv(h, &h->f0c[i], "f0_predictor.condnet." + std::to_string(2 * i), kF0C, i == 0 ? kMel : kF0C, 3)) return 1;
    const Host* t;
    if (get(h, "f0_predictor.classifier.weight", {1, kF0C}, &t) || upload(h, &h->f0_w, t->v.data(), kF0C)) return 1;
    if (get(h, "f0_predictor.classifier.bias", {1}, &t) || upload(h, &h->f0_b, t->v.data(), 1)) return 1;
    if (get(h, "m_source.l_linear.weight", {1, S3HIFT_HARMONICS}, &t) || upload(h, &h->lin_w, t->v.data(), S3HIFT_HARMONICS)) return 1;
    if (get(h, "m_source.l_linear.bias", {1}, &t) || upload(h, &h->lin_b, t->v.data(), 1)) return 1;
    const size_t BT = (size_t)h->max_batch;
    int len_factor = 1;
    for (int i = 0; i < 3; ++i) {
        Stage& s = h->st[i];
        const int cin = kBase >> i, cout = kBase >> (i + 1), u = kUpRate[i], kw = kUpKernel[i];
        s.C = cout;
        len_factor *= u;
        s.ld = h->max_frames * len_factor + 32;
        // ConvTranspose1d weight [cin][cout][kw]: phase ph owns taps ph, ph + u, ...; as a plain convolution over positions
        // q its tap j' (input position q + j' - (J - 1)) carries transposed tap ph + (J - 1 - j') u
        const Host *w, *b;
        const std::string n = "ups." + std::to_string(i);
        if (get(h, n + ".weight", {cin, cout, kw}, &w) || get(h, n + ".bias", {cout}, &b)) return 1;
        {
            const int J = (kw + u - 1) / u;              // taps of phase 0, the longest; shorter phases are zero-filled
            s.up_taps = J;
            std::vector<float> wc((size_t)u * cout * cin * J, 0.f);
            for (int ph = 0; ph < u; ++ph)
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < cin; ++ci)
                        for (int jp = 0; jp < J; ++jp) {
                            const int k = ph + (J - 1 - jp) * u;
                            if (k < kw) wc[(((size_t)ph * cout + co) * cin + ci) * J + jp] = w->v[((size_t)ci * cout + co) * kw + k];
                        }
            if (make_conv_from(h, &s.up, wc.data(), b->v.data(), u * cout, cin, J, cout)) return 1;
        }
        {
            // source_downs.i: w[cout][18][2 stride] (stride 1: [cout][18][1]) -> w'[cout][c * stride + r][j] = w[cout][c][j stride + r]
            const int sd = kDownStride[i], kd = kDownK[i], taps = kd / sd;
            s.down_cpad = (kBins * sd + CONV_CK - 1) / CONV_CK * CONV_CK;
            const Host *dw, *db;
            const std::string d = "source_downs." + std::to_string(i);
            if (get(h, d + ".weight", {cout, kBins, kd}, &dw) || get(h, d + ".bias", {cout}, &db)) return 1;
            std::vector<float> wc((size_t)cout * s.down_cpad * taps, 0.f);
            for (int co = 0; co < cout; ++co)
                for (int c = 0; c < kBins; ++c)
                    for (int k = 0; k < kd; ++k)
                        wc[((size_t)co * s.down_cpad + c * sd + k % sd) * taps + k / sd] = dw->v[((size_t)co * kBins + c) * kd + k];
            if (make_conv_from(h, &s.down, wc.data(), db->v.data(), cout, s.down_cpad, taps)) return 1;
            if (dalloc(h, reinterpret_cast<void**>(&s.SD), BT * s.down_cpad * s.ld * sizeof(float))) return 1;
        }
        if (make_resblock(h, &s.src, "source_resblocks." + std::to_string(i), cout, kSrcResKernel[i])) return 1;
        for (int j = 0; j < 3; ++j)
            if (make_resblock(h, &s.res[j], "resblocks." + std::to_string(i * 3 + j), cout, kResKernel[j])) return 1;
        const size_t n_el = BT * s.C * s.ld;
        for (float** p : {&s.A, &s.S0, &s.X, &s.T1, &s.U0, &s.U1, &s.XS})
            if (dalloc(h, reinterpret_cast<void**>(p), n_el * sizeof(float))) return 1;
    }
    const size_t sst = BT * kBins * ((size_t)h->max_frames * 120 + 32);
    if (dalloc(h, reinterpret_cast<void**>(&h->SST), sst * sizeof(float))) return 1;
    if (dalloc(h, reinterpret_cast<void**>(&h->CP), sst * sizeof(float))) return 1;
    if (dalloc(h, reinterpret_cast<void**>(&h->P), BT * kBase * h->max_frames * sizeof(float))) return 1;
    if (dalloc(h, reinterpret_cast<void**>(&h->F0A), BT * kF0C * h->max_frames * sizeof(float))) return 1;
    if (dalloc(h, reinterpret_cast<void**>(&h->F0B), BT * kF0C * h->max_frames * sizeof(float))) return 1;
    if (dalloc(h, reinterpret_cast<void**>(&h->f0_tmp), BT * h->max_frames * sizeof(float))) return 1;
    if (dalloc(h, reinterpret_cast<void**>(&h->prefix), BT * S3HIFT_HARMONICS * h->max_frames * sizeof(double))) return 1;
    for (hipEvent_t& e : h->ev) S3_HIP_CHECK(hipEventCreate(&e));
    S3_HIP_CHECK(hipDeviceSynchronize());
    h->raw.clear();
    h->finalized = true;
    return 0;
}

extern "C" int s3hift_f0(s3hift* h, const float* mel_dev, int32_t B, int32_t T, float* f0_dev, void* stream) {
    if (check_call(h, B, T)) return 1;
    S3_REQUIRE(mel_dev && f0_dev, "null argument");
    return run_f0(h, mel_dev, B, T, f0_dev, static_cast<hipStream_t>(stream));
}

extern "C" int s3hift_source(s3hift* h, const float* f0_dev, int32_t B, int32_t T, uint64_t seed, const float* phase_dev,
                             const float* noise_dev, float* source_dev, void* stream) {
    if (check_call(h, B, T)) return 1;
    S3_REQUIRE(f0_dev && source_dev, "null argument");
    return launch_source(f0_dev, T, h->prefix, phase_dev, noise_dev, seed, h->lin_w, h->lin_b, source_dev, B,
                         static_cast<hipStream_t>(stream));
}

extern "C" int s3hift_decode(s3hift* h, const float* mel_dev, const float* source_dev, int32_t B, int32_t T, float* wav_dev,
                             void* stream) {
    if (check_call(h, B, T)) return 1;
    S3_REQUIRE(mel_dev && source_dev && wav_dev, "null argument");
    return run_decode(h, mel_dev, source_dev, B, T, wav_dev, static_cast<hipStream_t>(stream));
}

extern "C" int s3hift_inference(s3hift* h, const float* mel_dev, int32_t B, int32_t T, uint64_t seed,
                                const float* cache_source_dev, int32_t n_cache, float* wav_dev, float* source_dev, void* stream) {
    if (check_call(h, B, T)) return 1;
    S3_REQUIRE(mel_dev && wav_dev && source_dev && n_cache >= 0 && (n_cache == 0 || cache_source_dev), "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (run_f0(h, mel_dev, B, T, h->f0_tmp, st)) return 1;
    if (launch_source(h->f0_tmp, T, h->prefix, nullptr, nullptr, seed, h->lin_w, h->lin_b, source_dev, B, st)) return 1;
    if (launch_overwrite_head(source_dev, T * S3HIFT_SAMPLES_PER_FRAME, cache_source_dev, n_cache, B, st)) return 1;
    return run_decode(h, mel_dev, source_dev, B, T, wav_dev, st);
}

extern "C" int s3hift_set_timing(s3hift* h, int32_t on) {
    S3_REQUIRE(h, "null handle");
    h->timing = on != 0;
    return 0;
}

extern "C" int s3hift_last_timing(s3hift* h, float* out_ms, int32_t n) {
    S3_REQUIRE(h && out_ms && n >= 1, "bad arguments");
    for (int i = 0; i < n && i < 6; ++i) out_ms[i] = h->last_ms[i];
    return 0;
}

extern "C" int s3hift_op_conv1d(const float* x_dev, const float* w_host, const float* bias_host, const float* alpha_dev,
                                const float* res_dev, float* y_dev, int32_t B, int32_t Cin, int32_t Cout, int32_t K, int32_t Tin,
                                int32_t dil, int32_t pad, int32_t act, float slope, void* stream) {
    S3_REQUIRE(x_dev && w_host && bias_host && y_dev && B >= 1 && Cin >= 16 && Cout >= 1 && K >= 1 && Tin >= 1, "bad arguments");
    S3_REQUIRE(act != ACT_SNAKE || alpha_dev, "Snake needs alpha");
    const int Tout = Tin + 2 * pad - dil * (K - 1);
    S3_REQUIRE(Tout >= 1, "empty output");
    s3hift tmp;                                   // owns the temporary weight buffers
    Conv c;
    int rc = make_conv_from(&tmp, &c, w_host, bias_host, Cout, Cin, K);
    if (!rc) {
        ConvArgs a = conv_args(c, x_dev, (size_t)Cin * Tin, Tin, Tin, y_dev, (size_t)Cout * Tout, Tout, Tout);
        a.dil = dil; a.pad = pad; a.act = act; a.slope = slope; a.alpha = alpha_dev; a.res = res_dev;
        rc = launch_conv1d(a, B, static_cast<hipStream_t>(stream));
        if (!rc && hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) { s3_set_error("conv1d launch failed"); rc = 1; }
    }
    for (void* p : tmp.owned) (void)hipFree(p);
    return rc;
}
