// Stand-alone timeline of the decode projection kernels: builds csrc/gemm.hip with stage stamps
// (T3_GEMM_TIMELINE) and replays a graph of back-to-back launches over rotating weight copies.
// Prints, per kernel shape, when (relative to the first wave's start) waves start, have their
// MFMA results, pass the LDS barrier and finish, and the idle time between consecutive launches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DT3_GEMM_TIMELINE tools/gemm_timeline.hip -o gpurun_out/gemm_timeline
#include "../chatterbox_vllm_amd/csrc/gemm.hip"

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

void t3_set_error(const std::string& msg) { fprintf(stderr, "error: %s\n", msg.c_str()); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run_case(const char* name, int epi, int M, int N, int K, int nblocks_x, int split, int waves) {
    const int L = 12;
    const size_t wbytes = (size_t)N * K * 2;
    const int ncopies = (int)std::max<size_t>(2, 600000000 / wbytes);
    std::vector<bf16_t*> W(ncopies);
    for (auto& p : W) { CK(hipMalloc((void**)&p, wbytes)); CK(hipMemset(p, 0, wbytes)); }
    bf16_t *A, *C; float* slab; long long* ts;
    CK(hipMalloc((void**)&A, (size_t)(M + 16) * K * 2)); CK(hipMemset(A, 0, (size_t)(M + 16) * K * 2));
    CK(hipMalloc((void**)&C, (size_t)M * N * 2));
    CK(hipMalloc((void**)&slab, (size_t)GEMM_SLAB_SPLIT * M * N * 4));
    const size_t nblk = (size_t)nblocks_x * split;
    const size_t per = nblk * 128;
    CK(hipMalloc((void**)&ts, per * L * 8));
    CK(hipMemset(ts, 0, per * L * 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < L; ++i) {
        GemmArgs a{};
        a.A = A; a.lda = K; a.Wt = W[i % ncopies]; a.C = C; a.M = M; a.N = N; a.K = K; a.n_valid = N; a.slab = slab;
        a.ldc = epi == EPI_SILU ? N / 2 : N;
        a.ts = ts + per * i;
        if ((epi == EPI_SLAB ? launch_gemm_slab(a, st) : launch_gemm(epi, a, st))) return 1;
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    std::vector<long long> h(per * L);
    CK(hipMemcpy(h.data(), ts, per * L * 8, hipMemcpyDeviceToHost));
    printf("%s  M=%d N=%d K=%d blocks=%zu waves/block=%d  (times in us from the launch's first wave start; min/median/max over waves)\n",
           name, M, N, K, nblk, waves);
    long long prev_end = 0;
    for (int i = L - 4; i < L; ++i) {
        std::vector<double> s[5];
        long long t0 = -1, tend = 0;
        for (size_t b = 0; b < nblk; ++b)
            for (int w = 0; w < waves; ++w) {
                const long long v = h[per * i + b * 128 + w * 8];
                if (t0 < 0 || v < t0) t0 = v;
            }
        for (size_t b = 0; b < nblk; ++b)
            for (int w = 0; w < waves; ++w)
                for (int k = 0; k < 5; ++k) {
                    const long long v = h[per * i + b * 128 + w * 8 + k];
                    s[k].push_back((v - t0) * 0.01);
                    if (v > tend) tend = v;
                }
        printf("  launch %2d: idle before %5.2f |", i, prev_end ? (t0 - prev_end) * 0.01 : 0.0);
        const char* lab[5] = {"start", "mfma issued", "mfma done", "barrier", "end"};
        for (int k = 0; k < 5; ++k) {
            std::sort(s[k].begin(), s[k].end());
            printf(" %s %.2f/%.2f/%.2f |", lab[k], s[k].front(), s[k][s[k].size() / 2], s[k].back());
        }
        printf("\n");
        prev_end = tend;
    }
    for (auto p : W) (void)hipFree(p);
    (void)hipFree(A); (void)hipFree(C); (void)hipFree(slab); (void)hipFree(ts);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(st);
    return 0;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 52;
    if (run_case("gate_up silu", EPI_SILU, M, 8192, 1024, 256, 1, 8)) return 1;
    if (run_case("qkv-shaped silu", EPI_SILU, M, 3072, 1024, 96, 1, 8)) return 1;
    if (run_case("down slab", EPI_SLAB, M, 1024, 4096, 32, GEMM_SLAB_SPLIT, 8)) return 1;
    if (run_case("o slab", EPI_SLAB, M, 1024, 1024, 32, GEMM_SLAB_SPLIT, 4)) return 1;
    return 0;
}
