// Stand-alone timeline of attn_decode_kernel: builds csrc/attn.hip with stage stamps (T3_ATTN_TIMELINE)
// and replays a graph of back-to-back launches over different layers of a synthetic KV pool (so no
// launch finds its K/V in a cache).  Prints when waves start, know their page, have their first
// tile, leave the tile loop and finish, and the whole-launch time and algorithmic bandwidth.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DT3_ATTN_TIMELINE tools/attn_timeline.hip -o tools/_build/attn_timeline
#include "../chatterbox_vllm_amd/csrc/attn.hip"

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

void t3_set_error(const std::string& msg) { fprintf(stderr, "error: %s\n", msg.c_str()); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 64;
    const int sorted = argc > 2 ? atoi(argv[2]) : 0;       // 1: rows in descending context order
    const int H = 16, hd = 64, max_pages = 16, L = 12;
    const int layers = argc > 3 ? atoi(argv[3]) : 12;       // 1: every launch re-reads the same K/V (cache-resident)
    const int n_streams = rows;
    const int64_t n_pages = (int64_t)n_streams * max_pages;
    const int64_t page_elems = 2LL * H * T3_PAGE * hd;
    const int64_t layer_stride = n_pages * page_elems;
    bf16_t* pool; CK(hipMalloc((void**)&pool, (size_t)layer_stride * layers * 2));
    CK(hipMemset(pool, 0, (size_t)layer_stride * layers * 2));
    std::vector<int32_t> bt(n_streams * max_pages), rs(rows), rc(rows);
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    std::vector<int32_t> perm(n_pages);
    for (int64_t i = 0; i < n_pages; ++i) perm[i] = (int32_t)i;
    for (int64_t i = n_pages - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (i + 1)]);
    for (size_t i = 0; i < bt.size(); ++i) bt[i] = perm[i];
    int64_t ctx_total = 0;
    for (int r = 0; r < rows; ++r) { rs[r] = r; rc[r] = 100 + (int)(rnd() % 690); }
    if (sorted) std::sort(rc.begin(), rc.end(), [](int a, int b) { return a > b; });
    if (sorted == 2) {                                     // longest, shortest, 2nd longest, 2nd shortest, ...
        std::vector<int32_t> z;
        for (int i = 0, j = rows - 1; i <= j; ++i, --j) { z.push_back(rc[i]); if (i != j) z.push_back(rc[j]); }
        rc = z;
    }
    if (sorted == 3) {                                     // long rows first, in groups of 8 with 8 short rows in between
        std::vector<int32_t> z;
        for (int g = 0; g < rows / 16; ++g) {
            for (int k = 0; k < 8; ++k) z.push_back(rc[g * 8 + k]);
            for (int k = 0; k < 8; ++k) z.push_back(rc[rows - 1 - g * 8 - k]);
        }
        if ((int)z.size() == rows) rc = z;
    }
    for (int r = 0; r < rows; ++r) ctx_total += rc[r];
    int32_t *d_bt, *d_rs, *d_rc; bf16_t *qkv, *out; long long* ts;
    CK(hipMalloc((void**)&d_bt, bt.size() * 4)); CK(hipMemcpy(d_bt, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_rs, rows * 4)); CK(hipMemcpy(d_rs, rs.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_rc, rows * 4)); CK(hipMemcpy(d_rc, rc.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&qkv, (size_t)rows * 3 * H * hd * 2)); CK(hipMemset(qkv, 0, (size_t)rows * 3 * H * hd * 2));
    CK(hipMalloc((void**)&out, (size_t)rows * H * hd * 2));
    const size_t per = (size_t)rows * H * ATT_NW * 8;
    CK(hipMalloc((void**)&ts, per * L * 8)); CK(hipMemset(ts, 0, per * L * 8));
    KvView kv; kv.pool = pool; kv.layer_stride = layer_stride; kv.block_table = d_bt; kv.max_pages = max_pages; kv.H = H; kv.hd = hd;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < L; ++i)
        hipLaunchKernelGGL((attn_decode_kernel<1, false, false>), dim3(H, rows), dim3(ATT_NW * 64), 0, st, qkv, 3 * H * hd, out, H * hd,
                           rows, d_rs, d_rc, kv, i % layers, (const int32_t*)nullptr, 0, (const bf16_t*)nullptr,
                           (float*)nullptr, ts + per * i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    std::vector<long long> h(per * L);
    CK(hipMemcpy(h.data(), ts, per * L * 8, hipMemcpyDeviceToHost));
    const double bytes = (double)ctx_total * H * hd * 2 * 2;
    printf("attn_decode rows=%d mean ctx=%.0f  %.1f MB per launch%s  (us from the launch's first wave start; min/median/max over waves)\n",
           rows, (double)ctx_total / rows, bytes / 1e6, sorted ? "  rows sorted by context, longest first" : "");
    long long prev_end = 0;
    for (int i = L - 4; i < L; ++i) {
        std::vector<double> s[5];
        long long t0 = -1, tend = 0;
        const size_t nw = (size_t)rows * H * ATT_NW;
        for (size_t w = 0; w < nw; ++w) { const long long v = h[per * i + w * 8]; if (t0 < 0 || v < t0) t0 = v; }
        for (size_t w = 0; w < nw; ++w)
            for (int k = 0; k < 5; ++k) {
                const long long v = h[per * i + w * 8 + k];
                if (v == 0) continue;                        // wave without a tile
                s[k].push_back((v - t0) * 0.01);
                if (v > tend) tend = v;
            }
        printf("  launch %2d: idle before %5.2f | total %6.2f us = %5.0f GB/s |", i, prev_end ? (t0 - prev_end) * 0.01 : 0.0,
               (tend - t0) * 0.01, bytes / ((tend - t0) * 10.0));
        const char* lab[5] = {"start", "page known", "tile 1 done", "tiles done", "end"};
        for (int k = 0; k < 5; ++k) {
            std::sort(s[k].begin(), s[k].end());
            printf(" %s %.2f/%.2f/%.2f |", lab[k], s[k].front(), s[k][s[k].size() / 2], s[k].back());
        }
        printf("\n");
        prev_end = tend;
    }
    return 0;
}
