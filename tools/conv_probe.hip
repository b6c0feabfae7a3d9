// Times the vocoder's convolution kernel on its own:  conv_probe B Cin Cout K dil T act repeats
// Build: hipcc -O3 --offload-arch=gfx950 tools/conv_probe.hip chatterbox_vllm_amd/csrc_s3gen/build/hift_kernels.o -o tools/_build/conv_probe
#include "../chatterbox_vllm_amd/csrc_s3gen/hift_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
void s3_set_error(const std::string& m) { fprintf(stderr, "error: %s\n", m.c_str()); }
int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage: conv_probe B Cin Cout K dil T act repeats\n"); return 2; }
    const int B = atoi(argv[1]), Cin = atoi(argv[2]), Cout = atoi(argv[3]), K = atoi(argv[4]), dil = atoi(argv[5]), T = atoi(argv[6]),
              act = atoi(argv[7]), R = atoi(argv[8]);
    const int ld = (T + 31) / 32 * 32, CoutPad = (Cout + 31) / 32 * 32;
    float *x, *y, *res, *wt, *bias, *alpha;
    const size_t nx = (size_t)B * Cin * ld, ny = (size_t)B * Cout * ld, nw = (size_t)CoutPad * Cin * K;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&res, ny * 4); hipMalloc(&wt, nw * 4); hipMalloc(&bias, Cout * 4); hipMalloc(&alpha, Cin * 4);
    std::vector<float> h(std::max(nx, std::max(ny, nw)));
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(res, h.data(), ny * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < nw; ++i) h[i] *= 0.05f;
    hipMemcpy(wt, h.data(), nw * 4, hipMemcpyHostToDevice); hipMemcpy(bias, h.data(), Cout * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < Cin; ++i) h[i] = 1.0f;
    hipMemcpy(alpha, h.data(), Cin * 4, hipMemcpyHostToDevice);
    ConvArgs a{};
    a.x = x; a.x_bs = (size_t)Cin * ld; a.x_ld = ld; a.Tin = T; a.wt = wt; a.bias = bias; a.alpha = alpha;
    a.y = y; a.y_bs = (size_t)Cout * ld; a.y_ld = ld; a.Tout = T; a.res = getenv("PROBE_RES") ? res : nullptr;
    a.Cin = Cin; a.Cout = Cout; a.CoutPad = CoutPad; a.K = K; a.dil = dil; a.pad = dil * (K - 1) / 2; a.act = act; a.slope = 0.1f;
    a.out_stride = 1; a.Tq = T;
#ifdef S3_TIMELINE
    const int nwg = ((T + 127) / 128) * ((CoutPad + 63) / 64) * B;
    long long* tl; hipMalloc(&tl, (size_t)nwg * 4 * 64 * 8); hipMemset(tl, 0, (size_t)nwg * 4 * 64 * 8);
    a.timeline = tl;
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (launch_conv1d(a, B, nullptr)) return 1;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < R; ++i) launch_conv1d(a, B, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= R;
    const double flop = 2.0 * B * Cin * Cout * K * (double)T;
    printf("B=%d Cin=%d Cout=%d K=%d dil=%d T=%d act=%d: %.1f us, %.1f TFLOP/s (%.3f of 157.3)\n", B, Cin, Cout, K, dil, T, act, ms * 1e3,
           flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 157.3e12);
#ifdef S3_TIMELINE
    {
        std::vector<long long> ht((size_t)nwg * 4 * 64);
        hipMemcpy(ht.data(), tl, ht.size() * 8, hipMemcpyDeviceToHost);
        const int nchunk = Cin / 16, nst = 1 + 4 * nchunk + 2;
        std::vector<double> sum(nst, 0.0); long long cnt = 0; double total = 0;
        for (int g = 0; g < nwg * 4; ++g) {
            const long long* p = &ht[(size_t)g * 64];
            if (!p[nst - 1]) continue;
            for (int i = 1; i < nst; ++i) sum[i] += (double)(p[i] - p[i - 1]);
            total += (double)(p[nst - 1] - p[0]); ++cnt;
        }
        printf("timeline over %lld waves (cycles of the 100 MHz counter x ... raw units): total %.0f\n", cnt, total / cnt);
        printf("  prologue %.0f\n", sum[1] / cnt);
        for (int c = 0; c < nchunk; ++c)
            printf("  chunk %d: load-issue %.0f  mfma %.0f  store %.0f  barrier %.0f\n", c, sum[2 + 4 * c] / cnt, sum[3 + 4 * c] / cnt,
                   sum[4 + 4 * c] / cnt, sum[5 + 4 * c] / cnt);
        printf("  epilogue %.0f\n", sum[nst - 1] / cnt);
    }
#endif
    return 0;
}
