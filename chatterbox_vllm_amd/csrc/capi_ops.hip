// Per-op C entry points: thin argument checks around the kernel launchers so that parity
// tests can drive each kernel through the same C ABI the engine is shipped behind.
#include "t3_common.h"
#include "../../include/t3_hip.h"

#include <mutex>
#include <vector>

static KvView make_kv(void* pool, int64_t layer_stride, const int32_t* bt, int32_t max_pages, int H, int hd) {
    KvView v; v.pool = static_cast<bf16_t*>(pool); v.layer_stride = layer_stride; v.block_table = bt;
    v.max_pages = max_pages; v.H = H; v.hd = hd; return v;
}

extern "C" int t3_op_repack_weight(const void* w_dev, void* wt_dev, int32_t N, int32_t K, void* stream) {
    T3_REQUIRE(w_dev && wt_dev, "null argument");
    return launch_repack(static_cast<const bf16_t*>(w_dev), static_cast<bf16_t*>(wt_dev), N, K, static_cast<hipStream_t>(stream));
}

extern "C" int t3_op_add_rmsnorm(const void* x_dev, void* res_dev, const void* w_dev, void* xn_dev, int32_t rows,
                                 int32_t d, float eps, void* stream) {
    T3_REQUIRE(res_dev && w_dev && xn_dev, "null argument");
    return launch_add_rmsnorm(static_cast<const bf16_t*>(x_dev), static_cast<bf16_t*>(res_dev),
                              static_cast<const bf16_t*>(w_dev), static_cast<bf16_t*>(xn_dev), rows, d, eps,
                              static_cast<hipStream_t>(stream));
}

extern "C" int t3_op_gemm(const void* a_dev, const void* wt_dev, void* out_dev, int32_t M, int32_t N, int32_t K,
                          int32_t ldc, int32_t epilogue, void* stream) {
    T3_REQUIRE(a_dev && wt_dev && out_dev, "null argument");
    T3_REQUIRE(epilogue == 0 || epilogue == 1, "epilogue must be 0 (store) or 1 (residual add)");
    T3_REQUIRE(ldc >= 1, "bad ldc");
    GemmArgs g{};
    g.A = static_cast<const bf16_t*>(a_dev); g.lda = K; g.Wt = static_cast<const bf16_t*>(wt_dev);
    g.C = static_cast<bf16_t*>(out_dev); g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.n_valid = ldc < N ? ldc : N;
    return launch_gemm(epilogue == 0 ? EPI_STORE : EPI_RESID, g, static_cast<hipStream_t>(stream));
}

static int op_slab(bool prompt, const void* a_dev, const void* wt_dev, float* slab_dev, int32_t M, int32_t N, int32_t K,
                   void* stream) {
    T3_REQUIRE(a_dev && wt_dev && slab_dev, "null argument");
    GemmArgs g{};
    g.A = static_cast<const bf16_t*>(a_dev); g.lda = K; g.Wt = static_cast<const bf16_t*>(wt_dev);
    g.M = M; g.N = N; g.K = K; g.n_valid = N; g.slab = slab_dev;
    return prompt ? launch_gemm_prompt(EPI_SLAB, g, static_cast<hipStream_t>(stream))
                  : launch_gemm_slab(g, static_cast<hipStream_t>(stream));
}
extern "C" int t3_op_gemm_slab(const void* a_dev, const void* wt_dev, float* slab_dev, int32_t M, int32_t N, int32_t K,
                               void* stream) {
    return op_slab(false, a_dev, wt_dev, slab_dev, M, N, K, stream);
}
extern "C" int t3_op_gemm_prompt_slab(const void* a_dev, const void* wt_dev, float* slab_dev, int32_t M, int32_t N,
                                      int32_t K, void* stream) {
    return op_slab(true, a_dev, wt_dev, slab_dev, M, N, K, stream);
}

extern "C" int t3_op_slab_split(void) { return GEMM_SLAB_SPLIT; }
void set_prompt_gemm_form(int walk);
extern "C" int t3_op_set_prompt_gemm_form(int32_t walk) { set_prompt_gemm_form(walk); return 0; }

extern "C" int t3_op_slab_add_rmsnorm(const float* slab_dev, int32_t nsplit, void* res_dev, const void* w_dev, void* xn_dev,
                                      int32_t rows, int32_t d, float eps, void* stream) {
    T3_REQUIRE(res_dev && w_dev && xn_dev, "null argument");
    return launch_slab_add_rmsnorm(slab_dev, nsplit, static_cast<bf16_t*>(res_dev), static_cast<const bf16_t*>(w_dev),
                                   static_cast<bf16_t*>(xn_dev), rows, d, eps, static_cast<hipStream_t>(stream));
}

static int op_silu(bool prompt, const void* a_dev, const void* wt_dev, void* h_dev, int32_t M, int32_t F, int32_t K,
                   void* stream) {
    T3_REQUIRE(a_dev && wt_dev && h_dev, "null argument");
    GemmArgs g{};
    g.A = static_cast<const bf16_t*>(a_dev); g.lda = K; g.Wt = static_cast<const bf16_t*>(wt_dev);
    g.C = static_cast<bf16_t*>(h_dev); g.ldc = F; g.M = M; g.N = 2 * F; g.K = K; g.n_valid = 2 * F;
    return prompt ? launch_gemm_prompt(EPI_SILU, g, static_cast<hipStream_t>(stream))
                  : launch_gemm(EPI_SILU, g, static_cast<hipStream_t>(stream));
}
extern "C" int t3_op_gemm_silu(const void* a_dev, const void* wt_dev, void* h_dev, int32_t M, int32_t F, int32_t K,
                               void* stream) {
    return op_silu(false, a_dev, wt_dev, h_dev, M, F, K, stream);
}
extern "C" int t3_op_gemm_prompt_silu(const void* a_dev, const void* wt_dev, void* h_dev, int32_t M, int32_t F, int32_t K,
                                      void* stream) {
    return op_silu(true, a_dev, wt_dev, h_dev, M, F, K, stream);
}

static int op_qkv(bool prompt, const void* a_dev, const void* wt_dev, void* qkv_dev, int32_t M, int32_t H, int32_t hd,
                  int32_t K, const void* rope_dev, const int32_t* pos_dev, const int32_t* stream_dev,
                  void* kv_pool_dev, int64_t layer_stride_elems, int32_t layer,
                  const int32_t* block_table_dev, int32_t max_pages, void* stream) {
    T3_REQUIRE(a_dev && wt_dev && qkv_dev && rope_dev && pos_dev && stream_dev && kv_pool_dev && block_table_dev,
               "null argument");
    GemmArgs g{};
    g.A = static_cast<const bf16_t*>(a_dev); g.lda = K; g.Wt = static_cast<const bf16_t*>(wt_dev);
    g.C = static_cast<bf16_t*>(qkv_dev); g.ldc = 3 * H * hd; g.M = M; g.N = 3 * H * hd; g.K = K; g.n_valid = g.N;
    g.rope = static_cast<const bf16_t*>(rope_dev); g.pos = pos_dev; g.row_stream = stream_dev;
    g.kv = make_kv(kv_pool_dev, layer_stride_elems, block_table_dev, max_pages, H, hd); g.layer = layer;
    return prompt ? launch_gemm_prompt(EPI_QKV_ROPE, g, static_cast<hipStream_t>(stream))
                  : launch_gemm(EPI_QKV_ROPE, g, static_cast<hipStream_t>(stream));
}
extern "C" int t3_op_gemm_qkv_rope(const void* a_dev, const void* wt_dev, void* qkv_dev, int32_t M, int32_t H, int32_t hd,
                                   int32_t K, const void* rope_dev, const int32_t* pos_dev, const int32_t* stream_dev,
                                   void* kv_pool_dev, int64_t layer_stride_elems, int32_t layer,
                                   const int32_t* block_table_dev, int32_t max_pages, void* stream) {
    return op_qkv(false, a_dev, wt_dev, qkv_dev, M, H, hd, K, rope_dev, pos_dev, stream_dev, kv_pool_dev, layer_stride_elems,
                  layer, block_table_dev, max_pages, stream);
}
extern "C" int t3_op_gemm_prompt_qkv_rope(const void* a_dev, const void* wt_dev, void* qkv_dev, int32_t M, int32_t H,
                                          int32_t hd, int32_t K, const void* rope_dev, const int32_t* pos_dev,
                                          const int32_t* stream_dev, void* kv_pool_dev, int64_t layer_stride_elems,
                                          int32_t layer, const int32_t* block_table_dev, int32_t max_pages, void* stream) {
    return op_qkv(true, a_dev, wt_dev, qkv_dev, M, H, hd, K, rope_dev, pos_dev, stream_dev, kv_pool_dev, layer_stride_elems,
                  layer, block_table_dev, max_pages, stream);
}

extern "C" int t3_op_slab_add_rmsnorm_tiled(const float* slab_dev, int32_t nsplit, void* res_dev, const void* w_dev,
                                            void* xn_dev, int32_t rows, int32_t d, float eps, void* stream) {
    T3_REQUIRE(res_dev && w_dev && xn_dev, "null argument");
    return launch_slab_add_rmsnorm(slab_dev, nsplit, static_cast<bf16_t*>(res_dev), static_cast<const bf16_t*>(w_dev),
                                   static_cast<bf16_t*>(xn_dev), rows, d, eps, static_cast<hipStream_t>(stream), true);
}

extern "C" int t3_op_quantize_weight(const void* w_dev, void* wq_dev, float* scale_dev, int32_t N, int32_t K, void* stream) {
    T3_REQUIRE(w_dev && wq_dev && scale_dev, "null argument");
    return launch_quantize(static_cast<const bf16_t*>(w_dev), static_cast<uint8_t*>(wq_dev), scale_dev, N, K,
                           static_cast<hipStream_t>(stream));
}

extern "C" int t3_op_gemm_ex(const t3_gemm_desc* d, void* stream) {
    T3_REQUIRE(d && (d->a_dev || d->fn_norm_w_dev) && d->w_dev, "null argument");
    T3_REQUIRE(d->epilogue >= 0 && d->epilogue <= 4, "epilogue must be 0..4");
    T3_REQUIRE(!d->prompt_form || d->epilogue >= 2, "the prompt form exists for epilogues 2, 3 and 4");
    T3_REQUIRE(!d->prompt_form || (!d->a_tiled && !d->out_tiled), "the prompt form reads and writes row-major activations");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GemmArgs g{};
    g.A = static_cast<const bf16_t*>(d->a_dev); g.lda = d->K; g.Wt = static_cast<const bf16_t*>(d->w_dev);
    g.wscale = d->w_scale_dev; g.C = static_cast<bf16_t*>(d->out_dev); g.M = d->M; g.N = d->N; g.K = d->K;
    g.slab = d->slab_dev; g.a_tiled = d->a_tiled != 0; g.c_tiled = d->out_tiled != 0;
    if (d->fn_norm_w_dev) {
        T3_REQUIRE(!d->prompt_form && (d->epilogue == 2 || d->epilogue == 3), "the norm prologue belongs to the streaming forms 2 and 3");
        g.fn_w = static_cast<const bf16_t*>(d->fn_norm_w_dev); g.fn_eps = d->fn_eps; g.fn_nsplit = d->fn_nsplit;
        g.fn_slab = d->fn_slab_dev; g.fn_res_in = static_cast<const bf16_t*>(d->fn_res_in_dev);
        g.fn_res_out = static_cast<bf16_t*>(d->fn_res_out_dev);
        g.fn_pair_scale = d->fn_pair_scale_dev;
    }
    switch (d->epilogue) {
        case 0: case 1:
            T3_REQUIRE(d->out_dev && d->ldc >= 1, "bad output");
            g.ldc = d->ldc; g.n_valid = d->ldc < d->N ? d->ldc : d->N;
            return launch_gemm(d->epilogue == 0 ? EPI_STORE : EPI_RESID, g, st);
        case 2:
            T3_REQUIRE(d->out_dev, "bad output");
            g.ldc = d->N / 2; g.n_valid = d->N;
            return d->prompt_form ? launch_gemm_prompt(EPI_SILU, g, st) : launch_gemm(EPI_SILU, g, st);
        case 3:
            T3_REQUIRE(d->out_dev && d->rope_dev && d->pos_dev && d->stream_dev && d->kv_pool_dev && d->block_table_dev,
                       "null argument");
            T3_REQUIRE(d->N == 3 * d->H * d->hd, "N must be 3*H*hd");
            g.ldc = d->N; g.n_valid = d->N; g.rope = static_cast<const bf16_t*>(d->rope_dev); g.pos = d->pos_dev;
            g.row_stream = d->stream_dev; g.layer = d->layer;
            g.kv = make_kv(d->kv_pool_dev, d->layer_stride_elems, d->block_table_dev, d->max_pages, d->H, d->hd);
            return d->prompt_form ? launch_gemm_prompt(EPI_QKV_ROPE, g, st) : launch_gemm(EPI_QKV_ROPE, g, st);
        default:
            T3_REQUIRE(d->slab_dev, "null slab");
            g.n_valid = d->N;
            return d->prompt_form ? launch_gemm_prompt(EPI_SLAB, g, st) : launch_gemm_slab(g, st);
    }
}

extern "C" int t3_op_attn_decode(const void* qkv_dev, void* out_dev, int32_t rows, int32_t H, int32_t hd,
                                 const int32_t* stream_dev, const int32_t* ctx_dev, const void* kv_pool_dev,
                                 int64_t layer_stride_elems, int32_t layer, const int32_t* block_table_dev,
                                 int32_t max_pages, void* stream) {
    T3_REQUIRE(qkv_dev && out_dev && stream_dev && ctx_dev && kv_pool_dev && block_table_dev, "null argument");
    const KvView kv = make_kv(const_cast<void*>(kv_pool_dev), layer_stride_elems, block_table_dev, max_pages, H, hd);
    return launch_attn_decode(static_cast<const bf16_t*>(qkv_dev), 3 * H * hd, static_cast<bf16_t*>(out_dev), rows,
                              stream_dev, ctx_dev, kv, layer, static_cast<hipStream_t>(stream));
}

extern "C" int t3_op_attn_decode_prefix(const void* qkv_dev, void* out_dev, int32_t rows, int32_t H, int32_t hd,
                                        const int32_t* stream_dev, const int32_t* ctx_dev, const void* kv_pool_dev,
                                        int64_t layer_stride_elems, int32_t layer, const int32_t* block_table_dev,
                                        int32_t max_pages, const int32_t* prefix_page_dev, int32_t n_prefix, void* stream) {
    T3_REQUIRE(qkv_dev && out_dev && stream_dev && ctx_dev && kv_pool_dev && block_table_dev && prefix_page_dev, "null argument");
    const KvView kv = make_kv(const_cast<void*>(kv_pool_dev), layer_stride_elems, block_table_dev, max_pages, H, hd);
    return launch_attn_decode(static_cast<const bf16_t*>(qkv_dev), 3 * H * hd, static_cast<bf16_t*>(out_dev), rows,
                              stream_dev, ctx_dev, kv, layer, static_cast<hipStream_t>(stream), prefix_page_dev, n_prefix);
}

extern "C" int t3_op_attn_decode_oproj(const void* qkv_dev, float* slab_dev, int32_t rows, int32_t H, int32_t hd,
                                       const int32_t* stream_dev, const int32_t* ctx_dev, const void* kv_pool_dev,
                                       int64_t layer_stride_elems, int32_t layer, const int32_t* block_table_dev,
                                       int32_t max_pages, const void* wo_dev, int32_t wo_int8, int32_t rows_per_group,
                                       const int32_t* prefix_page_dev, int32_t n_prefix, void* stream) {
    T3_REQUIRE(qkv_dev && slab_dev && stream_dev && ctx_dev && kv_pool_dev && block_table_dev && wo_dev, "null argument");
    const KvView kv = make_kv(const_cast<void*>(kv_pool_dev), layer_stride_elems, block_table_dev, max_pages, H, hd);
    return launch_attn_decode_oproj(static_cast<const bf16_t*>(qkv_dev), 3 * H * hd, slab_dev, rows, stream_dev, ctx_dev, kv, layer,
                                    static_cast<const bf16_t*>(wo_dev), wo_int8 != 0, rows_per_group,
                                    static_cast<hipStream_t>(stream), prefix_page_dev, n_prefix);
}

extern "C" int t3_op_oproj_split(void) { return ATTN_OPROJ_SPLIT; }

extern "C" int t3_op_attn_states_floats(int32_t H) { return H * 8 * 8 * ATT_STATE_FLOATS; }

extern "C" int t3_op_attn_decode_split_oproj(const void* qkv_dev, float* states_dev, float* slab_dev, int32_t rows, int32_t H,
                                             int32_t hd, const int32_t* stream_dev, const int32_t* ctx_dev,
                                             const void* kv_pool_dev, int64_t layer_stride_elems, int32_t layer,
                                             const int32_t* block_table_dev, int32_t max_pages, const void* wo_dev,
                                             const float* wo_scale_dev, const int32_t* prefix_page_dev, int32_t n_prefix,
                                             void* stream) {
    T3_REQUIRE(qkv_dev && states_dev && slab_dev && stream_dev && ctx_dev && kv_pool_dev && block_table_dev && wo_dev, "null argument");
    T3_REQUIRE(rows >= 1 && rows <= GEMM_FUSED_NORM_ROWS && H == 16 && hd == 64, "the split form takes 1..4 rows of 16 heads x 64");
    const KvView kv = make_kv(const_cast<void*>(kv_pool_dev), layer_stride_elems, block_table_dev, max_pages, H, hd);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = launch_attn_decode_split(static_cast<const bf16_t*>(qkv_dev), 3 * H * hd, states_dev, rows, stream_dev, ctx_dev, kv,
                                          layer, st, prefix_page_dev, n_prefix)) return rc;
    GemmArgs o{};
    o.a_states = states_dev; o.lda = H * hd; o.Wt = static_cast<const bf16_t*>(wo_dev); o.wscale = wo_scale_dev;
    o.M = rows; o.N = H * hd; o.K = H * hd; o.n_valid = o.N; o.ldc = o.N; o.slab = slab_dev;
    return launch_gemm_slab(o, st);
}

extern "C" int t3_op_head_slab_add_rmsnorm(const float* slab_dev, const float* wo_scale_dev, void* res_dev, const void* w_dev,
                                           void* xn_dev, int32_t rows, int32_t d, float eps, void* stream) {
    T3_REQUIRE(slab_dev && res_dev && w_dev && xn_dev, "null argument");
    return launch_slab_add_rmsnorm(slab_dev, ATTN_OPROJ_SPLIT, static_cast<bf16_t*>(res_dev), static_cast<const bf16_t*>(w_dev),
                                   static_cast<bf16_t*>(xn_dev), rows, d, eps, static_cast<hipStream_t>(stream), false, nullptr,
                                   nullptr, nullptr, wo_scale_dev);
}

static int attn_prefill_op(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                           const int32_t* seq_start_host, int32_t H, int32_t hd, const int32_t* prefix_page_dev,
                           int32_t n_prefix, const KvView* kv, int32_t layer, void* stream);

extern "C" int t3_op_attn_prefill(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                                  const int32_t* seq_start_host, int32_t H, int32_t hd, void* stream) {
    return attn_prefill_op(qkv_dev, out_dev, n_seqs, seq_start_dev, seq_start_host, H, hd, nullptr, 0, nullptr, 0, stream);
}

extern "C" int t3_op_attn_prefill_prefix(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                                         const int32_t* seq_start_host, int32_t H, int32_t hd,
                                         const int32_t* prefix_page_dev, int32_t n_prefix, const void* kv_pool_dev,
                                         int64_t layer_stride_elems, int32_t layer, void* stream) {
    T3_REQUIRE(prefix_page_dev && kv_pool_dev, "null argument");
    const KvView kv = make_kv(const_cast<void*>(kv_pool_dev), layer_stride_elems, nullptr, 0, H, hd);
    return attn_prefill_op(qkv_dev, out_dev, n_seqs, seq_start_dev, seq_start_host, H, hd, prefix_page_dev, n_prefix, &kv,
                           layer, stream);
}

static int attn_prefill_op(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                           const int32_t* seq_start_host, int32_t H, int32_t hd, const int32_t* prefix_page_dev,
                           int32_t n_prefix, const KvView* kv, int32_t layer, void* stream) {
    T3_REQUIRE(qkv_dev && out_dev && seq_start_dev && seq_start_host && n_seqs >= 1, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::vector<int32_t> tiles;
    for (int i = 0; i < n_seqs; ++i) {
        T3_REQUIRE(seq_start_host[i + 1] > seq_start_host[i], "empty sequence");
        for (int q0 = 0; q0 < seq_start_host[i + 1] - seq_start_host[i]; q0 += 64) { tiles.push_back(i); tiles.push_back(q0); }
    }
    int32_t* d_tiles = nullptr;
    T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_tiles), tiles.size() * sizeof(int32_t)));
    T3_HIP_CHECK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    const int rc = launch_attn_prefill(static_cast<const bf16_t*>(qkv_dev), 3 * H * hd, static_cast<bf16_t*>(out_dev),
                                       seq_start_dev, d_tiles, (int)tiles.size() / 2, H, hd, st, prefix_page_dev, n_prefix,
                                       kv, layer);
    const hipError_t se = hipStreamSynchronize(st);
    (void)hipFree(d_tiles);
    if (rc) return rc;
    T3_HIP_CHECK(se);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// t3_op_attn_prefill for callers that cannot allocate (code under HIP graph capture, serving loops or harnesses that
// forbid hipMalloc).  attn_prefill_op above builds the (sequence, first query row) list of the launch's 64-query tiles
// on the host and stages it through a hipMalloc'ed buffer; here the list is built by a one-workgroup kernel from
// seq_start_dev straight into `scratch`, a file-scope device array every GPU holds its own copy of until the process
// ends, and the attention kernel reads it from there.  Nothing is allocated, copied or synchronised.
//
// In host code the array's NAME is not a device address; the address of the current device's copy comes from
// hipGetSymbolAddress, once per device, and only that pointer is handed to kernels.  Every call on a GPU uses that
// GPU's one array: calls that overlap on one GPU (two streams, two threads) race on it.  Every entry a launch reads
// has been written by the same call's build kernel, so nothing needs clearing between calls.
// ------------------------------------------------------------------------------------------------------------------
#define T3_STATIC_TILES_MAX 4096       // 64-query tiles per call (32 KiB of scratch); sum over sequences of ceil(rows / 64)
#define T3_STATIC_SEQS_MAX 1024        // sequences per call
static __device__ __align__(256) int32_t scratch[2 * T3_STATIC_TILES_MAX];

// Writes exactly `ntiles` (sequence, first query row) pairs, the number the host sized the launch for from ITS copy of
// seq_start.  Should the device copy disagree (a stale upload), tiles past `ntiles` are dropped instead of written
// behind the array, and pairs the device copy does not produce are filled with (0, 0), a valid tile of sequence 0, so
// the attention launch never reads a pair this call did not write.
__global__ __launch_bounds__(256) void build_prefill_tiles_kernel(const int32_t* __restrict__ seq_start, int n_seqs,
                                                                  int32_t* __restrict__ tiles, int ntiles) {
    __shared__ int32_t first_tile[T3_STATIC_SEQS_MAX + 1];
    for (int i = threadIdx.x; i < n_seqs; i += 256) first_tile[i + 1] = (seq_start[i + 1] - seq_start[i] + 63) >> 6;
    __syncthreads();
    if (threadIdx.x == 0) {                              // exclusive prefix sum; n_seqs <= 1024
        first_tile[0] = 0;
        for (int i = 0; i < n_seqs; ++i) first_tile[i + 1] += first_tile[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_seqs; i += 256)
        for (int t = first_tile[i], q0 = 0; t < first_tile[i + 1]; ++t, q0 += 64)
            if (t < ntiles) { tiles[2 * t] = i; tiles[2 * t + 1] = q0; }
    for (int t = first_tile[n_seqs] + (int)threadIdx.x; t < ntiles; t += 256) { tiles[2 * t] = 0; tiles[2 * t + 1] = 0; }
}

static int32_t* g_scratch_ptr[T3_MAX_DEVICES];           // address of `scratch` on each device, looked up on first use
static std::mutex g_scratch_mutex;

static int scratch_of_current_device(int32_t** out) {
    int dev = -1;
    T3_HIP_CHECK(hipGetDevice(&dev));
    T3_REQUIRE(dev >= 0 && dev < T3_MAX_DEVICES, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        T3_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));      // the copy of the device that is current now
        T3_REQUIRE(p != nullptr, "hipGetSymbolAddress returned no address");
        g_scratch_ptr[dev] = static_cast<int32_t*>(p);
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int t3_op_attn_prefill_static(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                                         const int32_t* seq_start_host, int32_t H, int32_t hd, void* stream) {
    T3_REQUIRE(qkv_dev && out_dev && seq_start_dev && seq_start_host, "null argument");
    T3_REQUIRE(n_seqs >= 1 && n_seqs <= T3_STATIC_SEQS_MAX, "attn_prefill_static: 1..1024 sequences per call");
    int64_t ntiles = 0;
    for (int i = 0; i < n_seqs; ++i) {
        T3_REQUIRE(seq_start_host[i + 1] > seq_start_host[i], "empty sequence");
        ntiles += (seq_start_host[i + 1] - seq_start_host[i] + 63) / 64;
    }
    T3_REQUIRE(ntiles <= T3_STATIC_TILES_MAX,
               "attn_prefill_static: more than 4096 query tiles (sum of ceil(rows / 64)); use t3_op_attn_prefill");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (st) {                                            // the array used is the current device's: the stream must live there
        hipDevice_t sdev = -1;
        int cur = -1;
        T3_HIP_CHECK(hipStreamGetDevice(st, &sdev));
        T3_HIP_CHECK(hipGetDevice(&cur));
        T3_REQUIRE((int)sdev == cur, "attn_prefill_static: the stream belongs to another device than the current one");
    }
    int32_t* tiles = nullptr;
    if (scratch_of_current_device(&tiles)) return 1;
    T3_LAUNCH(build_prefill_tiles_kernel, dim3(1), dim3(256), 0, st, seq_start_dev, n_seqs, tiles, (int)ntiles);
    T3_HIP_CHECK(hipGetLastError());
    return launch_attn_prefill(static_cast<const bf16_t*>(qkv_dev), 3 * H * hd, static_cast<bf16_t*>(out_dev), seq_start_dev,
                               tiles, (int)ntiles, H, hd, st, nullptr, 0, nullptr, 0);
}

extern "C" int t3_op_sample_greedy(const void* logits_dev, int32_t ldl, int32_t vocab, const int32_t* row_c_dev,
                                   const int32_t* row_u_dev, int32_t n, float cfg_scale, const float* rep_penalty_dev,
                                   uint8_t* seen_dev, int32_t seen_stride, int32_t* out_tokens_dev,
                                   float* out_logits_dev, void* stream) {
    T3_REQUIRE(logits_dev && row_c_dev && row_u_dev && rep_penalty_dev && seen_dev && out_tokens_dev, "null argument");
    SampleArgs s{};
    s.logits = static_cast<const bf16_t*>(logits_dev); s.ldl = ldl; s.vocab = vocab;
    s.row_c = row_c_dev; s.row_u = row_u_dev; s.n = n; s.cfg_scale = cfg_scale; s.rep_penalty = rep_penalty_dev;
    s.seen = seen_dev; s.seen_stride = seen_stride; s.out_tokens = out_tokens_dev; s.out_logits = out_logits_dev;
    return launch_sample(s, static_cast<hipStream_t>(stream));
}
