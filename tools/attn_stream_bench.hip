// Stand-alone A/B of the two decode attention launches: attn_decode_kernel (one workgroup per (row, head)) against
// attn_decode_stream_kernel (one persistent workgroup per CU, tiles dealt in equal shares).  Random K/V and q, the
// outputs must be bit-identical; then a graph of back-to-back launches over different layers of the pool (no launch
// finds its K/V in a cache) is timed for each.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 tools/attn_stream_bench.hip -o tools/_build/attn_stream_bench
//   tools/_build/attn_stream_bench [rows] [bins] [form: 0 one tile in flight, 1 two tiles in registers, 2 two tiles through LDS] [ctx_lo] [ctx_hi]
#include "../chatterbox_vllm_amd/csrc/attn.hip"
#include "attn_stream.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

void t3_set_error(const std::string& msg) { fprintf(stderr, "error: %s\n", msg.c_str()); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 62;
    const int bins = argc > 2 ? atoi(argv[2]) : 256;
    const int dbuf = argc > 3 ? atoi(argv[3]) : 1;
    const int ctx_lo = argc > 4 ? atoi(argv[4]) : 100, ctx_hi = argc > 5 ? atoi(argv[5]) : 790;
    const int H = 16, hd = 64, max_pages = 16, L = 24, layers = 12;
    const int n_streams = rows;
    const int64_t n_pages = (int64_t)n_streams * max_pages;
    const int64_t page_elems = 2LL * H * T3_PAGE * hd;
    const int64_t layer_stride = n_pages * page_elems;
    const size_t pool_bytes = (size_t)layer_stride * layers * 2;
    bf16_t* pool; CK(hipMalloc((void**)&pool, pool_bytes));
    {   // random bf16 values in (-2, 2): sign, exponent 126..127 region, random mantissa
        std::vector<uint16_t> hb(pool_bytes / 2);
        uint32_t s = 99;
        for (auto& x : hb) { s = s * 1664525u + 1013904223u; x = (uint16_t)(((s >> 16) & 0x807f) | (0x3f00 + ((s >> 8) & 0x80))); }
        CK(hipMemcpy(pool, hb.data(), pool_bytes, hipMemcpyHostToDevice));
    }
    std::vector<int32_t> bt(n_streams * max_pages), rs(rows), rc(rows);
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    std::vector<int32_t> perm(n_pages);
    for (int64_t i = 0; i < n_pages; ++i) perm[i] = (int32_t)i;
    for (int64_t i = n_pages - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (i + 1)]);
    for (size_t i = 0; i < bt.size(); ++i) bt[i] = perm[i];
    int64_t ctx_total = 0;
    for (int r = 0; r < rows; ++r) { rs[r] = r; rc[r] = ctx_lo + (int)(rnd() % (ctx_hi - ctx_lo + 1)); ctx_total += rc[r]; }

    // ---- the deal: pairs longest first onto the least loaded bin; a pair's items adjacent, v ascending
    struct Pair { int r, h, nt; };
    std::vector<Pair> pairs;
    for (int r = 0; r < rows; ++r) for (int h = 0; h < H; ++h) pairs.push_back({r, h, (rc[r] + 63) / 64});
    std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.nt > b.nt; });
    std::vector<std::vector<Pair>> bin(bins);
    std::priority_queue<std::pair<int, int>, std::vector<std::pair<int, int>>, std::greater<>> heap;
    for (int b = 0; b < bins; ++b) heap.push({0, b});
    for (auto& p : pairs) { auto [load, b] = heap.top(); heap.pop(); bin[b].push_back(p); heap.push({load + p.nt, b}); }
    std::vector<AttItem> items;
    std::vector<int32_t> bstart(bins + 1, 0);
    int max_items = 0, max_load = 0, min_load = 1 << 30;
    for (int b = 0; b < bins; ++b) {
        int load = 0;
        for (auto& p : bin[b]) {
            const int nv = std::min(p.nt, 8);
            for (int v = 0; v < nv; ++v) {
                AttItem it{};
                it.page[0] = bt[p.r * max_pages + v];
                it.page[1] = v + 8 < p.nt ? bt[p.r * max_pages + v + 8] : -1;
                it.row = p.r; it.hv = p.h | (v << 8) | (nv << 16); it.ctx = rc[p.r]; it.pfx = -1;
                items.push_back(it);
            }
            load += p.nt;
        }
        bstart[b + 1] = (int32_t)items.size();
        max_items = std::max(max_items, bstart[b + 1] - bstart[b]);
        max_load = std::max(max_load, load); min_load = std::min(min_load, load);
    }
    printf("rows %d, mean ctx %.0f, %zu pairs, %zu items in %d bins: tiles per bin %d..%d, most items in a bin %d\n", rows,
           (double)ctx_total / rows, pairs.size(), items.size(), bins, min_load, max_load, max_items);
    if (max_items > (dbuf == 2 ? ATT_LDS_MAX_ITEMS : ATT_STREAM_MAX_ITEMS)) { printf("too many items in a bin\n"); return 1; }

    int32_t *d_bt, *d_rs, *d_rc, *d_bs; AttItem* d_items; bf16_t *qkv, *out0, *out1;
    CK(hipMalloc((void**)&d_bt, bt.size() * 4)); CK(hipMemcpy(d_bt, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_rs, rows * 4)); CK(hipMemcpy(d_rs, rs.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_rc, rows * 4)); CK(hipMemcpy(d_rc, rc.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_bs, bstart.size() * 4)); CK(hipMemcpy(d_bs, bstart.data(), bstart.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_items, items.size() * sizeof(AttItem)));
    CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(AttItem), hipMemcpyHostToDevice));
    const size_t qbytes = (size_t)rows * 3 * H * hd * 2, obytes = (size_t)rows * H * hd * 2;
    CK(hipMalloc((void**)&qkv, qbytes));
    {
        std::vector<uint16_t> hq(qbytes / 2);
        uint32_t s = 7;
        for (auto& x : hq) { s = s * 1664525u + 1013904223u; x = (uint16_t)(((s >> 16) & 0x807f) | 0x3e80); }
        CK(hipMemcpy(qkv, hq.data(), qbytes, hipMemcpyHostToDevice));
    }
    CK(hipMalloc((void**)&out0, obytes)); CK(hipMalloc((void**)&out1, obytes));
    CK(hipMemset(out0, 0, obytes)); CK(hipMemset(out1, 0xff, obytes));
    KvView kv; kv.pool = pool; kv.layer_stride = layer_stride; kv.block_table = d_bt; kv.max_pages = max_pages; kv.H = H; kv.hd = hd;
    hipStream_t st; CK(hipStreamCreate(&st));

    auto base = [&](int layer, bf16_t* o) {
        hipLaunchKernelGGL((attn_decode_kernel<1, false, false>), dim3(H, rows), dim3(ATT_NW * 64), 0, st, qkv, 3 * H * hd, o, H * hd,
                           rows, d_rs, d_rc, kv, layer, (const int32_t*)nullptr, 0, (const bf16_t*)nullptr, (float*)nullptr);
    };
    auto stream = [&](int layer, bf16_t* o) {
        const bf16_t* pl = pool + (size_t)layer * layer_stride;
        if (dbuf == 2) hipLaunchKernelGGL(attn_decode_lds_kernel, dim3(bins), dim3(ATT_LDS_NW * 64), 0, st, qkv, 3 * H * hd, o, H * hd,
                                          (const AttItem*)d_items, (const int32_t*)d_bs, pl, H, 0);
        else if (dbuf) hipLaunchKernelGGL((attn_decode_stream_kernel<true>), dim3(bins), dim3(ATT_NW * 64), 0, st, qkv, 3 * H * hd, o, H * hd,
                                     (const AttItem*)d_items, (const int32_t*)d_bs, pl, H, 0);
        else hipLaunchKernelGGL((attn_decode_stream_kernel<false>), dim3(bins), dim3(ATT_NW * 64), 0, st, qkv, 3 * H * hd, o, H * hd,
                                (const AttItem*)d_items, (const int32_t*)d_bs, pl, H, 0);
    };
    // ---- same bits?
    for (int layer : {0, 5}) {
        base(layer, out0); stream(layer, out1);
        CK(hipStreamSynchronize(st));
        std::vector<uint16_t> a(obytes / 2), b(obytes / 2);
        CK(hipMemcpy(a.data(), out0, obytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out1, obytes, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
        printf("layer %d: %zu of %zu output values differ (first values %04x %04x %04x)\n", layer, bad, a.size(), a[0], a[1], a[2]);
        if (bad) return 2;
    }
    // ---- time
    const double bytes = (double)ctx_total * H * hd * 2 * 2;
    for (int which = 0; which < 2; ++which) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < L; ++i) { if (which) stream(i % layers, out1); else base(i % layers, out0); }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 20;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * L);
        printf("%-28s %6.2f us per launch (back to back, boundaries included) = %5.0f GB/s of %.1f MB\n",
               which ? (dbuf == 2 ? "stream, tiles through LDS" : dbuf ? "stream, two tiles in flight" : "stream, one tile in flight") : "one workgroup per pair", us, bytes / us / 1e3, bytes / 1e6);
    }
    return 0;
}
