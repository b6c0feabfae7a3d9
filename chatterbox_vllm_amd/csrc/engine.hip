// T3 decode engine: request queue, continuous-batching scheduler, KV page allocator and the
// per-step kernel sequence, behind the C ABI of include/t3_hip.h.
//
// Streams and slots.  A request occupies one slot; with CFG pairing it owns two KV streams
// (2*slot = conditional, 2*slot+1 = unconditional: same positions, the text rows of the
// unconditional prompt are zero embeddings).  A decode step runs rows 2i = cond stream and
// 2i+1 = uncond stream of its B requests through the 30 layers, requests ordered by context length,
// longest first: the attention grid is dispatched in row order, and starting the long rows first
// keeps the launch's tail short (measured: 9 % less attention time than arrival order).  Full steps, whose attention
// takes four rows per workgroup, pair long requests with short ones instead, so that the workgroups are even.
//
// Step = (a) admit waiting requests while a slot, the page reservation and the prompt-row
// budget allow, (b) prefill the admitted prompts whole and sample their first token,
// (c) decode one token for every request that was already running, (d) stop checks and
// slot/page release.  Sampled ids stay on the device (last_token[slot]) and feed the next
// step's embedding gather; the host only reads them for stop detection and output.
//
// Prefill runs beside decode: the prompts of newly admitted requests go through the layers on a second stream with
// their own activation buffers, as one prefill job, while the running requests keep decoding on the main stream (the
// two touch different KV pages, slots and sampler rows, and both chains are latency-bound, so most of the prefill
// time disappears behind the decode steps).  The job starts behind everything already queued on the main stream (the
// block-table upload, the step in flight); the main stream never waits for it: a step call polls the job's event,
// reports its first tokens once it has completed and only then lets its requests join the decode rows.  One job is
// outstanding at a time; admission pauses until it has been reported.  The measuring and debugging modes keep
// prefill inside the step, on the main stream.
//
// Steps are pipelined one deep: a call to t3_engine_step launches step k+1 and only then waits
// for step k and reports its tokens, so the GPU always has the next step queued while the host
// does its bookkeeping.  Step k+1 is planned before step k's tokens are known: length limits are
// known in advance, a stop token is not, so a request that stops in step k still rides along in
// step k+1; that one speculative row is discarded and the request's slot and pages are only
// released once the step that carried it has completed.  Host-side metadata is double-buffered
// by step parity.  The measuring/debugging modes (timing, kernel timing, logit taps,
// T3_DEBUG_SYNC, T3_ASYNC=0) run each step to completion inside the call instead.
//
// Shared conditioning prefix (cfg.prefix_sharing).  Positions 0..33 of every stream are the 34
// conditioning rows: identical in the cond and the uncond stream of a request and in every
// request that speaks with the same voice, so their K/V are identical too (the layers are
// causal).  The engine keeps them once per voice in a page of its own: the first request that
// brings a voice carries the 34 rows through the layers as a sequence of their own, inside the
// same prefill launch; every stream of that voice then prefills only positions 34.. and reads
// keys 0..33 from the voice page, in prefill and in decode.  Tile boundaries and summation
// orders are those of private copies, so ids and logits are bit-identical to an engine without
// sharing; what changes is 34 fewer prompt rows per stream and 34 fewer positions of KV to
// fetch from HBM per row and layer (all rows read the same page, which stays in L2).
//
// KV pages hold 64 positions of one stream for all layers.  Pages are handed out lazily as a
// stream grows; admission reserves the worst case (prompt + max_tokens, capped by
// max_model_len) so a running request can never starve, and nothing is ever preempted.
#include "t3_common.h"
#include "../../include/t3_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

static thread_local std::string g_err;
void t3_set_error(const std::string& msg) { g_err = msg; }
extern "C" const char* t3_last_error(void) { return g_err.c_str(); }
extern "C" int t3_abi_version(void) { return T3_ABI_VERSION; }

namespace {

constexpr int kCondRows = 34;        // models/t3/t3.py:42
constexpr int kOffset = 1000;        // models/t3/t3.py:49
constexpr int kSOS = 6561;           // modules/t3_config.py:8
constexpr int kCondStart = 695, kCondEnd = 696, kPrefillEnd = 697;   // t3.py:38-40
constexpr int kSeenStride = 9216;    // >= 8194 + 1000, multiple of 4

struct Request {
    int64_t id = 0;
    std::vector<int32_t> text;
    std::vector<float> cond;          // [34, d]
    t3_sampling sp{};
    int slot = -1;
    int prompt_len = 0;               // 34 + T + 1
    int n_out = 0;                    // tokens reported to the caller
    int n_launched = 0;               // tokens whose computation has been enqueued (>= n_out)
    int limit = 0;                    // tokens this request may produce before it ends on length
    bool zombie = false;              // stopped or aborted while a launched step still carries it
    int reserve_pages = 0;            // worst-case pages per stream
    int have_pages = 0;               // pages allocated per stream
    bool aborted = false;
    bool poisoned = false;            // ended on non-finite logits: its KV pages hold NaN and are cleared before reuse
    bool prefilling = false;          // admitted into a prefill job that has not been reported yet
    int pf_pos = 0;                   // next position of the prompt that has not gone through the layers (chunked prompts)
    int chunks_out = 0;               // launched, unreported jobs that carry a chunk of its prompt without finishing it
    int voice = -1;                   // shared-prefix entry this request reads positions 0..33 from
    bool builds_voice = false;        // its prefill step also carries the voice's 34 rows
};

// A conditioning prefix whose K/V live in a page of their own (prefix sharing).
struct Voice {
    std::vector<float> cond;          // [34, d] as given by the first request that brought it
    uint64_t hash = 0;
    int refs = 0;                     // requests between admission and release
    uint64_t last_use = 0;
    bool used = false;
};

uint64_t hash_floats(const float* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    for (size_t i = 0; i < n; ++i) {
        uint32_t w;
        std::memcpy(&w, p + i, 4);
        h = (h ^ w) * 0x100000001B3ULL;
        h ^= h >> 29;
    }
    return h;
}

struct Layer {
    bf16_t *ln1 = nullptr, *ln2 = nullptr, *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wd = nullptr;
    float *sqkv = nullptr, *so = nullptr, *sgu = nullptr, *sd = nullptr;   // int8 weights: per-column scales
};

// kernel classes for the per-launch event timing mode
enum KClass { K_EMBED, K_NORM, K_QKV, K_ATTN_DECODE, K_ATTN_PREFILL, K_OPROJ, K_GATEUP, K_DOWN, K_HEAD, K_SAMPLE,
              K_SEEN, K_NCLASS };
const char* const kClassName[K_NCLASS] = {"embed", "add_rmsnorm", "gemm_qkv_rope", "attn_decode", "attn_prefill",
                                          "gemm_o_resid", "gemm_gateup_silu", "gemm_down_resid", "gemm_head",
                                          "sample", "seen_init"};

template <typename T>
struct Pinned {
    T* h = nullptr; T* d = nullptr; size_t n = 0;
    int alloc(size_t count) {
        n = count;
        T3_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h), sizeof(T) * count));
        T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), sizeof(T) * count));
        std::memset(h, 0, sizeof(T) * count);
        return 0;
    }
    void release() { if (h) (void)hipHostFree(h); if (d) (void)hipFree(d); h = d = nullptr; }
};

}  // namespace

struct t3_engine {
    t3_config cfg{};
    hipStream_t stream = nullptr;                       // main stream: decode steps, block-table uploads, step events
    bool stream_owned = true;                           // false: the caller's stream (t3_engine_create_on_stream), not destroyed here
    hipStream_t pstream = nullptr;                      // prefill of the step's admissions (== stream in the synchronous modes)
    hipStream_t pstream_own = nullptr;                  // the second stream itself
    hipStream_t cur = nullptr;                          // the stream the phase being enqueued uses
    hipEvent_t fork_ev[2] = {nullptr, nullptr};
    // the prefill job that has been launched on the prefill stream and not yet reported
    struct PrefillJob { bool valid = false; int parity = 0; std::vector<Request*> reqs, partial; } pjob;   // partial: chunk carried, prompt not finished
    hipEvent_t pjob_done = nullptr;
    bool overlap_prefill = true;
    bool fused_norm = true;                             // few-row decode steps: norm inside the QKV / gate_up projections
    bool fused_oproj = false;                           // full decode steps: o_proj inside the grouped attention launch (split-K over the heads)
    bool split_attn = false;                            // few-row decode steps: attention split over single-wave workgroups
                                                        // (T3_SPLIT_ATTN=1; measured: attention 7.6 -> 6.0 us, o_proj 3.5 -> 4.3 us,
                                                        // single stream 1330 -> 1295 tokens/s, so off)
    int group_rows = ATTN_GROUP_ROWS;                   // rows per workgroup of that form (T3_GROUP_ROWS: 1, 2 or 4)
    int group_min_rows = 48;                            // decode rows from which the grouped attention + o_proj form runs
    int rows_per_req = 1;                               // rows a request contributes to a step (2 with CFG pairing)
    int kv_per_req = 1;                                 // KV streams a request owns (2 when the uncond stream attends)
    bool uncond_ghost = false;                          // cfg_paired == 2: the uncond row gets no attention and no KV
    int n_streams = 0, n_rows = 0, max_pages = 0;       // KV streams / decode rows of a full engine
    int head_n = 0;                                     // speech vocab padded to 16
    bool finalized = false;

    // raw tensors as loaded (device), released at finalize when repacked
    struct Raw { bf16_t* p; std::vector<int64_t> shape; };
    std::map<std::string, Raw> raw;

    bf16_t *text_emb = nullptr, *speech_emb = nullptr, *text_pos = nullptr, *speech_pos = nullptr;
    bf16_t *rope = nullptr, *norm_w = nullptr, *head = nullptr;
    int rope_rows = 0;
    std::vector<Layer> layers;

    // KV pool
    bf16_t* kv_pool = nullptr;
    int64_t n_pages = 0, layer_stride = 0;
    std::vector<int32_t> free_pages;
    int64_t reserved_unallocated = 0;                   // sum over running requests, in pages
    Pinned<int32_t> block_table;                        // [n_streams][max_pages]; the device copy serves the decode steps (main stream)
    bool block_table_dirty = true;
    // A prefill job reads its own device copy, uploaded from its own pinned snapshot on the prefill stream: the job spans
    // several decode steps, each of which may re-upload the decode copy.  One job is outstanding at a time.
    int32_t* bt_prefill_d = nullptr;
    int32_t* bt_prefill_stage[2] = {nullptr, nullptr};  // by step parity: with prefill inside pipelined steps two uploads can be queued
    const int32_t* bt_cur = nullptr;                    // the copy the phase being enqueued reads

    // activations (shared by prefill and decode: steps execute in stream order)
    struct Bufs {
        bf16_t *res = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *hbuf = nullptr, *logits = nullptr;
        bf16_t* res2 = nullptr;                         // second residual buffer of the few-row decode form (GEMM_FUSED_NORM_ROWS rows)
        float* att_states = nullptr;                    // few-row decode form: per-wave softmax states of the split attention
        float* slab = nullptr;                          // [GEMM_SLAB_SPLIT][rows][d] fp32 split-K partials (decode)
    } bufs, pbufs;                                      // decode rows / prompt rows
    Bufs* b = &bufs;
    int max_rows = 0;
    int64_t cur_ctx_tokens = 0;                         // positions of K/V the decode rows being enqueued must fetch: sum of
                                                        // their context lengths, a shared prefix counted once per voice
    float* cond_buf = nullptr;                          // [max_num_seqs or n_voices][34][d] fp32

    // shared conditioning prefixes: voice v owns KV stream n_streams + v, whose only page is n_pages + v
    bool share_prefix = false;
    int n_voices = 0;
    std::vector<Voice> voices;
    uint64_t voice_clock = 0;
    uint8_t* seen = nullptr;                            // [max_num_seqs][kSeenStride]
    int32_t* last_token = nullptr;                      // [max_num_seqs]
    float* dbg_logits = nullptr;                        // [max_num_seqs][vocab+1000] when enabled
    bool keep_logits = false;

    // per-step metadata: one pinned block, one H2D copy per phase.  The sampled ids travel the other way without a copy:
    // the sampler stores them straight into pinned host memory (out_tokens / out_tokens_dev are the two addresses of it),
    // which the host reads after the step's event.
    struct Meta {
        Pinned<uint8_t> blk;
        int32_t* out_tokens_dev = nullptr;
        int32_t *row_stream, *row_pos, *row_ctx, *row_pfx, *ekind, *eidx, *eaux, *row_c, *row_u, *slot, *step, *a_rows,
                *seq_start, *seq_pfx, *seq_first, *seq_stream, *tiles, *prompt_ids, *out_tokens = nullptr;   // host views (out_tokens: its own pinned buffer)
        float *rep, *temp, *top_p, *min_p;
        uint64_t* seed;
        size_t rows_cap = 0, req_cap = 0, ids_cap = 0, seq_cap = 0, tile_cap = 0;
    } dm[2], pm[2];                                      // decode / prefill, by step parity
    int32_t* bt_stage[2] = {nullptr, nullptr};          // pinned snapshots of the block table, by step parity

    // the step that has been launched but not yet reported
    struct InFlight {
        bool valid = false;
        int parity = 0;
        std::vector<Request*> admitted, running;
        std::vector<Request*> partial;                  // prompts the step carried a chunk of without finishing them
    } inflight;
    int parity = 0;
    hipEvent_t step_done[2] = {nullptr, nullptr};
    bool async_steps = true;

    std::deque<std::unique_ptr<Request>> waiting;
    int schedule_policy = 0;                            // 0: first come first served; 1: longest prompt + max_tokens first; 2: longest prompt first
    int prefill_chunk = 0;                              // > 0: positions a prefill job may carry; a prompt that does not fit is cut
    std::vector<std::unique_ptr<Request>> slots;

    std::unordered_map<int, hipGraphExec_t> graphs;
    std::unordered_map<int, int> rows_seen;

    t3_step_stats stats{};
    bool timing = false;
    bool debug_sync = false;

    // per-launch event timing (t3_engine_set_kernel_timing): [phase 0 decode / 1 prefill][class]
    bool ktiming = false;
    int kphase = 0;
    std::vector<hipEvent_t> kpool;
    struct KPending { int phase, cls, ev; int64_t bytes; };
    std::vector<KPending> kpending;
    size_t kused = 0;
    double kms[2][K_NCLASS] = {};
    int64_t klaunches[2][K_NCLASS] = {};
    int64_t kbytes[2][K_NCLASS] = {};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};

    KvView kvview() const {
        KvView v; v.pool = kv_pool; v.layer_stride = layer_stride; v.block_table = bt_cur ? bt_cur : block_table.d;
        v.max_pages = max_pages; v.H = cfg.n_heads; v.hd = cfg.head_dim; return v;
    }
};

namespace {

template <typename T>
T* carve(T*& base, size_t n) { T* p = base; base += n; return p; }

// With T3_DEBUG_SYNC=1 every launch is followed by a synchronize, so a faulting kernel is named.
// In kernel-timing mode every launch carries two HIP events bound to its own dispatch.
int stage_check(t3_engine* e, const char* label, int layer = -1);
int ktime_begin(t3_engine* e, int cls, int64_t bytes);
int ktime_end(t3_engine* e);
#define T3_RUN(e, cls, layer, bytes, call)                              \
    do {                                                                \
        if (ktime_begin(e, cls, bytes)) return 1;                       \
        if (call) return 1;                                             \
        if (ktime_end(e)) return 1;                                     \
        if (stage_check(e, kClassName[cls], layer)) return 1;           \
    } while (0)

int meta_alloc(t3_engine::Meta& m, size_t rows, size_t reqs, size_t prompt_ids) {
    m.rows_cap = rows; m.req_cap = reqs; m.ids_cap = prompt_ids;
    // sequences of a prefill step: up to two streams per request plus one per voice built in the step
    const size_t seqs = 3 * reqs + 2, tiles = rows / 32 + 5 * reqs + 64;
    m.seq_cap = seqs - 1; m.tile_cap = tiles;
    const size_t ni = rows * 7 + reqs * 5 + reqs * 2 /*a_rows*/ + 4 * seqs + 2 * tiles + prompt_ids + 64;
    const size_t nu = reqs + 4, nf = reqs * 4 + 16;
    if (m.blk.alloc(sizeof(uint64_t) * nu + sizeof(float) * nf + sizeof(int32_t) * ni)) return 1;
    m.seed = reinterpret_cast<uint64_t*>(m.blk.h);
    float* f = reinterpret_cast<float*>(m.seed + nu);
    int32_t* p = reinterpret_cast<int32_t*>(f + nf);
    T3_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m.out_tokens), sizeof(int32_t) * reqs,
                           hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: device stores reach the host without a cache flush
    T3_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&m.out_tokens_dev), m.out_tokens, 0));
    std::memset(m.out_tokens, 0, sizeof(int32_t) * reqs);
    m.row_stream = carve(p, rows); m.row_pos = carve(p, rows); m.row_ctx = carve(p, rows); m.row_pfx = carve(p, rows);
    m.ekind = carve(p, rows); m.eidx = carve(p, rows); m.eaux = carve(p, rows);
    m.row_c = carve(p, reqs); m.row_u = carve(p, reqs); m.slot = carve(p, reqs); m.step = carve(p, reqs);
    m.a_rows = carve(p, 2 * reqs); m.seq_start = carve(p, seqs); m.seq_pfx = carve(p, seqs); m.seq_first = carve(p, seqs);
    m.seq_stream = carve(p, seqs); m.tiles = carve(p, 2 * tiles);
    m.prompt_ids = carve(p, prompt_ids);
    m.rep = carve(f, reqs); m.temp = carve(f, reqs); m.top_p = carve(f, reqs); m.min_p = carve(f, reqs);
    return 0;
}
template <typename T> T* dev_of(const t3_engine::Meta& m, const T* host_view) {
    return reinterpret_cast<T*>(m.blk.d + (reinterpret_cast<const uint8_t*>(host_view) - m.blk.h));
}

int upload(t3_engine* e, t3_engine::Meta& m) {
    T3_HIP_CHECK(hipMemcpyAsync(m.blk.d, m.blk.h, m.blk.n, hipMemcpyHostToDevice, e->cur));
    return 0;
}

int stage_check(t3_engine* e, const char* label, int layer) {
    if (!e->debug_sync) return 0;
    const hipError_t err = hipStreamSynchronize(e->cur);
    if (err != hipSuccess) {
        t3_set_error(std::string("stage ") + label + (layer >= 0 ? " layer " + std::to_string(layer) : std::string()) +
                     ": " + hipGetErrorString(err));
        return 1;
    }
    return 0;
}

int ktime_begin(t3_engine* e, int cls, int64_t bytes) {
    if (!e->ktiming) return 0;
    if (e->kused + 2 > e->kpool.size()) {
        const size_t want = e->kpool.size() + 1024;
        while (e->kpool.size() < want) {
            hipEvent_t ev;
            T3_HIP_CHECK(hipEventCreate(&ev));
            e->kpool.push_back(ev);
        }
    }
    e->kpending.push_back({e->kphase, cls, (int)e->kused, bytes});
    t3_launch_events = T3LaunchEvents{e->kpool[e->kused], e->kpool[e->kused + 1]};   // consumed by the next T3_LAUNCH
    return 0;
}
int ktime_end(t3_engine* e) {
    if (!e->ktiming) return 0;
    if (t3_launch_events.start) {                     // the call launched nothing
        t3_launch_events = T3LaunchEvents{nullptr, nullptr};
        e->kpending.pop_back();
        return 0;
    }
    e->kused += 2;
    return 0;
}
// after the step's synchronize: fold the event pairs into the per-class totals
int ktime_collect(t3_engine* e) {
    for (const auto& p : e->kpending) {
        float ms = 0.f;
        T3_HIP_CHECK(hipEventElapsedTime(&ms, e->kpool[p.ev], e->kpool[p.ev + 1]));
        e->kms[p.phase][p.cls] += ms;
        e->klaunches[p.phase][p.cls] += 1;
        e->kbytes[p.phase][p.cls] += p.bytes;
    }
    e->kpending.clear();
    e->kused = 0;
    return 0;
}

int dmalloc(bf16_t** p, size_t elems) { T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), elems * sizeof(bf16_t))); return 0; }

// ---- the layer stack over `rows` rows whose embeddings are already in e->b->res ----------
// d_pfx: the shared-prefix page per decode row / per prompt sequence (nullptr without prefix sharing)
// d_emb_slot (decode steps of more than GEMM_FUSED_NORM_ROWS rows): the first norm fetches the rows' token embeddings itself
int run_layers(t3_engine* e, int rows, const int32_t* d_pos, const int32_t* d_stream, const int32_t* d_ctx,
               bool prefill, const int32_t* d_seq_start, const int32_t* d_tiles, int ntiles, const int32_t* d_pfx,
               const int32_t* d_emb_slot = nullptr, const int32_t* d_seq_first = nullptr, const int32_t* d_seq_stream = nullptr) {
    const t3_config& c = e->cfg;
    const int D = c.d_model, F = c.d_ff, H = c.n_heads, hd = c.head_dim;
    hipStream_t st = e->cur;
    const KvView kv = e->kvview();
    const long long WB = c.weight_int8 ? 1 : 2;          // bytes per projection weight (algorithmic-bytes accounting)
    // decode rows: the normalised activations and the MLP hidden rows travel in activation tile order (t3_common.h),
    // so every projection operand load is 1 KiB of contiguous memory; prompt rows stay row-major for the LDS-tiled kernels
    const bool tiled = !prefill;
    // Decode steps of a few rows (a single stream, the tail of a queue) are nothing but launch latency: there the two norm
    // launches of a layer move into the prologue of the projection that follows them (same arithmetic in the same order,
    // so the same bits as the separate kernel; gemm.hip).  Every workgroup then reads the residual rows, so the updated
    // rows go to the other of two buffers.
    const bool fuse = !prefill && e->fused_norm && rows <= GEMM_FUSED_NORM_ROWS;
    // Full decode steps: the attention workgroup of (head, 4 rows) also multiplies its outputs with the head's columns of Wo
    // (attn.hip): o_proj as a split-K over the heads, no launch of its own, folded by the norm kernel with the additions of
    // the stand-alone projection's eight slabs (same bits, so a request may pass from one form to the other as the batch
    // grows and shrinks).  One workgroup per (head, row) would re-read the head's 128 KiB of Wo from L2 once per row, which
    // costs more than the launch saves (measured at 62 rows: attention 21.2 -> 26.6 us); with few rows there are too few
    // workgroups to pull Wo at the chip's rate (measured with one row: 1287 against 1333 tokens/s).
    const bool oproj_in_attn = !prefill && e->fused_oproj && rows >= e->group_min_rows;
    const int o_split = oproj_in_attn ? ATTN_OPROJ_SPLIT : GEMM_SLAB_SPLIT;
    const bool split_attn = !prefill && !oproj_in_attn && e->split_attn && rows <= GEMM_FUSED_NORM_ROWS;
    bf16_t *res = e->b->res, *res_other = e->b->res2;
    for (int li = 0; li < c.n_layers; ++li) {
        const Layer& L = e->layers[li];
        // o_proj / down_proj leave split-K partial slabs, folded in by the next norm kernel.  Prompt rows always
        // take the LDS-tiled kernels and decode rows the weight-streaming ones, whatever the row count, so that
        // a row's bits never depend on what shares its launch.
        if (!fuse)
            T3_RUN(e, K_NORM, li, 4LL * rows * D, launch_slab_add_rmsnorm(e->b->slab, li > 0 ? GEMM_SLAB_SPLIT : 0, res, L.ln1,
                                                                          e->b->xn, rows, D, c.rms_eps, st, tiled,
                                                                          li == 0 && d_emb_slot ? e->speech_emb : nullptr,
                                                                          d_emb_slot, e->last_token));
        GemmArgs g{};
        if (fuse) {
            g.fn_w = L.ln1; g.fn_eps = c.rms_eps; g.fn_res_in = res; g.fn_nsplit = li > 0 ? GEMM_SLAB_SPLIT : 0;
            g.fn_slab = e->b->slab; g.fn_res_out = res_other;
            if (li > 0) std::swap(res, res_other);
        }
        g.A = e->b->xn; g.lda = D; g.a_tiled = tiled; g.Wt = L.wqkv; g.wscale = L.sqkv; g.C = e->b->qkv; g.ldc = 3 * D; g.M = rows; g.N = 3 * D; g.K = D;
        g.n_valid = 3 * D; g.rope = e->rope; g.pos = d_pos; g.row_stream = d_stream; g.kv = kv; g.layer = li;
        T3_RUN(e, K_QKV, li, WB * 3 * D * D, prefill ? launch_gemm_prompt(EPI_QKV_ROPE, g, st) : launch_gemm(EPI_QKV_ROPE, g, st));
        if (prefill) {
            T3_RUN(e, K_ATTN_PREFILL, li, 2LL * rows * 4 * D, launch_attn_prefill(e->b->qkv, 3 * D, e->b->att, d_seq_start, d_tiles, ntiles, H, hd, st,
                                                                                          d_pfx, kCondRows, &kv, li, d_seq_first, d_seq_stream));
        } else if (split_attn) {
            // a few rows: one single-wave workgroup per (head, row, wave), so that a row's K/V are pulled by eight CUs;
            // o_proj merges the wave states itself (same arithmetic, same bits as the one-workgroup form).  The merge in
            // the prologue of all 256 o_proj workgroups costs what the shorter attention launch gains.
            T3_RUN(e, K_ATTN_DECODE, li, e->cur_ctx_tokens * 4LL * D,
                   launch_attn_decode_split(e->b->qkv, 3 * D, e->b->att_states, rows, d_stream, d_ctx, kv, li, st, d_pfx, kCondRows));
        } else if (oproj_in_attn) {
            // decode rows: o_proj runs inside the attention launch, one fp32 partial slab per head
            T3_RUN(e, K_ATTN_DECODE, li, e->cur_ctx_tokens * 4LL * D + WB * D * D,
                   launch_attn_decode_oproj(e->b->qkv, 3 * D, e->b->slab, rows, d_stream, d_ctx, kv, li, L.wo, L.so != nullptr,
                                            e->group_rows, st, d_pfx, kCondRows));
        } else {
            T3_RUN(e, K_ATTN_DECODE, li, e->cur_ctx_tokens * 4LL * D, launch_attn_decode(e->b->qkv, 3 * D, e->b->att, rows, d_stream, d_ctx, kv, li, st,
                                                                                                     d_pfx, kCondRows));
        }
        if (!oproj_in_attn) {
            GemmArgs o{};
            o.A = e->b->att; o.lda = D; o.Wt = L.wo; o.wscale = L.so; o.C = e->b->res; o.ldc = D; o.M = rows; o.N = D; o.K = D; o.n_valid = D;
            o.slab = e->b->slab;
            if (split_attn) o.a_states = e->b->att_states;
            T3_RUN(e, K_OPROJ, li, WB * D * D, prefill ? launch_gemm_prompt(EPI_SLAB, o, st) : launch_gemm_slab(o, st));
        }
        if (!fuse)
            T3_RUN(e, K_NORM, li, 4LL * rows * D, launch_slab_add_rmsnorm(e->b->slab, o_split, res, L.ln2, e->b->xn, rows, D,
                                                                          c.rms_eps, st, tiled, nullptr, nullptr, nullptr,
                                                                          oproj_in_attn ? L.so : nullptr));
        GemmArgs u{};
        if (fuse) {
            u.fn_w = L.ln2; u.fn_eps = c.rms_eps; u.fn_res_in = res; u.fn_nsplit = o_split;
            u.fn_pair_scale = oproj_in_attn ? L.so : nullptr;
            u.fn_slab = e->b->slab; u.fn_res_out = res_other;
            std::swap(res, res_other);
        }
        u.A = e->b->xn; u.lda = D; u.a_tiled = tiled; u.c_tiled = tiled; u.Wt = L.wgu; u.wscale = L.sgu; u.C = e->b->hbuf; u.ldc = F; u.M = rows; u.N = 2 * F; u.K = D; u.n_valid = 2 * F;
        T3_RUN(e, K_GATEUP, li, WB * 2 * F * D, prefill ? launch_gemm_prompt(EPI_SILU, u, st) : launch_gemm(EPI_SILU, u, st));
        GemmArgs dn{};
        dn.A = e->b->hbuf; dn.lda = F; dn.a_tiled = tiled; dn.Wt = L.wd; dn.wscale = L.sd; dn.C = e->b->res; dn.ldc = D; dn.M = rows; dn.N = D; dn.K = F; dn.n_valid = D;
        dn.slab = e->b->slab;
        T3_RUN(e, K_DOWN, li, WB * F * D, prefill ? launch_gemm_prompt(EPI_SLAB, dn, st) : launch_gemm_slab(dn, st));
    }
    T3_RUN(e, K_NORM, -1, 4LL * rows * D, launch_slab_add_rmsnorm(e->b->slab, GEMM_SLAB_SPLIT, res, e->norm_w, e->b->xn, rows, D,
                                                                  c.rms_eps, st, tiled));
    return 0;
}

int run_head_and_sample(t3_engine* e, t3_engine::Meta& m, int head_rows, const int32_t* d_a_rows, int n_req) {
    const t3_config& c = e->cfg;
    GemmArgs g{};
    g.A = e->b->xn; g.lda = c.d_model; g.a_rows = d_a_rows; g.a_tiled = d_a_rows == nullptr;   // decode: tile-ordered xn
    g.Wt = e->head; g.C = e->b->logits; g.ldc = e->head_n;
    g.M = head_rows; g.N = e->head_n; g.K = c.d_model; g.n_valid = e->head_n;
    T3_RUN(e, K_HEAD, -1, 2LL * e->head_n * c.d_model, launch_gemm(EPI_STORE, g, e->cur));
    SampleArgs s{};
    s.logits = e->b->logits; s.ldl = e->head_n; s.vocab = c.speech_vocab;
    s.row_c = dev_of(m, m.row_c); s.row_u = dev_of(m, m.row_u); s.slot = dev_of(m, m.slot);
    s.n = n_req; s.cfg_scale = c.cfg_scale;
    s.rep_penalty = dev_of(m, m.rep); s.temperature = dev_of(m, m.temp);
    s.top_p = dev_of(m, m.top_p); s.min_p = dev_of(m, m.min_p);
    s.seed = dev_of(m, m.seed); s.step = dev_of(m, m.step);
    s.seen = e->seen; s.seen_stride = kSeenStride; s.last_token = e->last_token;
    s.out_tokens = m.out_tokens_dev;
    s.out_logits = e->keep_logits ? e->dbg_logits : nullptr;
    T3_RUN(e, K_SAMPLE, -1, 4LL * n_req * c.speech_vocab, launch_sample(s, e->cur));
    return 0;
}

void fill_sampling(t3_engine::Meta& m, int i, const Request& r) {
    m.slot[i] = r.slot;
    m.step[i] = r.n_launched;                          // index of the token being sampled
    m.rep[i] = r.sp.repetition_penalty;
    m.temp[i] = r.sp.temperature;
    m.top_p[i] = r.sp.top_p;
    m.min_p[i] = r.sp.min_p;
    m.seed[i] = r.sp.has_seed ? r.sp.seed : (0x5DEECE66DULL * (uint64_t)(r.id + 1));
}

int pages_for(int positions) { return (positions + T3_PAGE - 1) / T3_PAGE; }

int grow_pages(t3_engine* e, Request& r, int positions) {
    const int need = pages_for(positions);
    while (r.have_pages < need) {
        for (int u = 0; u < e->kv_per_req; ++u) {
            if (e->free_pages.empty()) { t3_set_error("KV pool exhausted despite reservation"); return 1; }
            const int32_t pg = e->free_pages.back(); e->free_pages.pop_back();
            e->block_table.h[(size_t)(r.slot * e->kv_per_req + u) * e->max_pages + r.have_pages] = pg;
        }
        e->reserved_unallocated -= e->kv_per_req;
        r.have_pages++;
        e->block_table_dirty = true;
    }
    return 0;
}

void release_request(t3_engine* e, Request& r) {
    const size_t page_elems = (size_t)2 * e->cfg.n_heads * T3_PAGE * e->cfg.head_dim;
    for (int u = 0; u < e->kv_per_req; ++u)
        for (int p = 0; p < r.have_pages; ++p) {
            const int32_t pg = e->block_table.h[(size_t)(r.slot * e->kv_per_req + u) * e->max_pages + p];
            // The attention kernels weight the unused positions of a page by zero, which does not silence a NaN: pages
            // that a request with non-finite values wrote go back to the pool cleared (behind the steps that used them).
            if (r.poisoned)
                for (int li = 0; li < e->cfg.n_layers; ++li)
                    (void)hipMemsetAsync(e->kv_pool + (size_t)li * e->layer_stride + (size_t)pg * page_elems, 0,
                                         page_elems * sizeof(bf16_t), e->stream);
            e->free_pages.push_back(pg);
        }
    e->reserved_unallocated -= (int64_t)(r.reserve_pages - r.have_pages) * e->kv_per_req;
    if (r.voice >= 0) { e->voices[r.voice].refs--; r.voice = -1; }
}

// Find the voice entry holding r's conditioning rows, or claim one for them (an unused entry, else the
// least recently used one that no admitted request reads).  A claimed entry is built by r's prefill step.
int acquire_voice(t3_engine* e, Request& r) {
    const size_t n = (size_t)kCondRows * e->cfg.d_model;
    const uint64_t h = hash_floats(r.cond.data(), n);
    int pick = -1;
    for (int v = 0; v < e->n_voices; ++v) {
        Voice& vc = e->voices[v];
        if (vc.used && vc.hash == h && std::memcmp(vc.cond.data(), r.cond.data(), n * sizeof(float)) == 0) { pick = v; break; }
    }
    if (pick < 0) {
        for (int v = 0; v < e->n_voices; ++v) {
            const Voice& vc = e->voices[v];
            if (vc.refs > 0) continue;
            if (!vc.used) { pick = v; break; }
            if (pick < 0 || vc.last_use < e->voices[pick].last_use) pick = v;
        }
        if (pick < 0) { t3_set_error("no free shared-prefix entry (max_voices too small)"); return 1; }
        Voice& vc = e->voices[pick];
        vc.cond = std::move(r.cond);
        vc.hash = h;
        vc.used = true;
        r.builds_voice = true;
    }
    std::vector<float>().swap(r.cond);
    Voice& vc = e->voices[pick];
    vc.refs++;
    vc.last_use = ++e->voice_clock;
    r.voice = pick;
    return 0;
}

// The host table keeps changing while earlier copies may still be queued, so each step uploads
// from its own parity's snapshot.
int sync_block_table(t3_engine* e) {
    if (!e->block_table_dirty) return 0;
    int32_t* stage = e->bt_stage[e->parity];
    std::memcpy(stage, e->block_table.h, sizeof(int32_t) * e->block_table.n);
    T3_HIP_CHECK(hipMemcpyAsync(e->block_table.d, stage, sizeof(int32_t) * e->block_table.n,
                                hipMemcpyHostToDevice, e->stream));
    e->block_table_dirty = false;
    return 0;
}

// The prefill stream waits for what has been enqueued on the main stream up to now (nothing to do when the two
// are one stream).
int prefill_after_main(t3_engine* e) {
    if (e->pstream == e->stream) return 0;
    T3_HIP_CHECK(hipEventRecord(e->fork_ev[e->parity], e->stream));
    T3_HIP_CHECK(hipStreamWaitEvent(e->pstream, e->fork_ev[e->parity], 0));
    return 0;
}

// ---- prefill of newly admitted requests ------------------------------------------------
// One (request, position range) of a prefill job.  Whole prompts: [first position, prompt_len).  Chunked prompts
// (prefill_chunk > 0): a job carries at most prefill_chunk positions, so a prompt may take several jobs; the chunk that
// ends at prompt_len samples the request's first token.
struct PrefillPart { Request* r; int p0, p1; };

// reqs_out: the requests whose prompt ends in this job, in the order of the sampler's rows
int run_prefill(t3_engine* e, const std::vector<PrefillPart>& parts, std::vector<Request*>& reqs_out) {
    const t3_config& c = e->cfg;
    t3_engine::Meta& m = e->pm[e->parity];
    const int S = e->rows_per_req, Q = e->kv_per_req;
    const bool share = e->share_prefix;
    const bool chunked = e->prefill_chunk > 0;
    int row = 0, nseq = 0, ntiles = 0, pid = 0;
    reqs_out.clear();
    for (const PrefillPart& pt : parts) if (pt.p1 == pt.r->prompt_len) reqs_out.push_back(pt.r);
    const int n = (int)reqs_out.size();                   // requests that reach the sampler
    {   // the job must fit the step's metadata and activation buffers: checked before anything is written
        size_t need_rows = e->uncond_ghost ? (size_t)n : 0, need_ids = 0, need_seqs = 0, need_tiles = 0;
        for (const PrefillPart& pt : parts) {
            T3_REQUIRE(pt.p0 >= 0 && pt.p0 < pt.p1 && pt.p1 <= pt.r->prompt_len, "prefill job: bad position range");
            need_rows += (size_t)(pt.p1 - pt.p0) * Q;
            need_seqs += Q;
            need_tiles += (size_t)Q * ((pt.p1 - pt.p0 + 63) / 64);
            if (share && pt.r->builds_voice) { need_rows += kCondRows; need_seqs += 1; need_tiles += 1; }
            if (pt.p1 == pt.r->prompt_len) need_ids += pt.r->prompt_len;
        }
        T3_REQUIRE(need_rows <= m.rows_cap && need_rows <= (size_t)e->max_rows && need_ids <= m.ids_cap && need_seqs <= m.seq_cap &&
                   need_tiles <= m.tile_cap && (size_t)n <= m.req_cap, "prefill job larger than the step's buffers");
    }
    // the prompts run on the prefill stream, behind everything the main stream holds so far
    e->cur = e->pstream;
    e->b = &e->pbufs;
    if (prefill_after_main(e)) return 1;
    // voices first met in this step: their 34 rows go through the layers as sequences of their own and
    // leave K/V in the voice page, which the requests' rows read in the same launches
    if (share)
        for (const PrefillPart& pt : parts) {
            Request& r = *pt.r;
            if (!r.builds_voice) continue;
            r.builds_voice = false;
            const int v = r.voice;
            T3_HIP_CHECK(hipMemcpyAsync(e->cond_buf + (size_t)v * kCondRows * c.d_model, e->voices[v].cond.data(),
                                        sizeof(float) * kCondRows * c.d_model, hipMemcpyHostToDevice, e->cur));
            m.seq_start[nseq] = row; m.seq_pfx[nseq] = -1; m.seq_first[nseq] = 0; m.seq_stream[nseq] = e->n_streams + v;
            m.tiles[2 * ntiles] = nseq; m.tiles[2 * ntiles + 1] = 0; ++ntiles;
            for (int p = 0; p < kCondRows; ++p, ++row) {
                m.row_stream[row] = e->n_streams + v; m.row_pos[row] = p; m.row_ctx[row] = p + 1;
                m.ekind[row] = 0; m.eidx[row] = p; m.eaux[row] = v;
            }
            ++nseq;
        }
    int fin = 0;                                          // index among the finishing requests
    for (const PrefillPart& pt : parts) {
        Request& r = *pt.r;
        const int T = (int)r.text.size(), P = r.prompt_len;
        const bool last = pt.p1 == P, first = pt.p0 == (share ? kCondRows : 0);
        if (grow_pages(e, r, pt.p1)) return 1;
        if (!share && first)
            T3_HIP_CHECK(hipMemcpyAsync(e->cond_buf + (size_t)r.slot * kCondRows * c.d_model, r.cond.data(),
                                        sizeof(float) * kCondRows * c.d_model, hipMemcpyHostToDevice, e->cur));
        for (int u = 0; u < Q; ++u) {
            m.seq_start[nseq] = row;
            m.seq_pfx[nseq] = share ? (int32_t)(e->n_pages + r.voice) : -1;
            m.seq_first[nseq] = pt.p0;
            m.seq_stream[nseq] = r.slot * Q + u;
            for (int q0 = 0; q0 < pt.p1 - pt.p0; q0 += 64) { m.tiles[2 * ntiles] = nseq; m.tiles[2 * ntiles + 1] = q0; ++ntiles; }
            for (int p = pt.p0; p < pt.p1; ++p, ++row) {
                m.row_stream[row] = r.slot * Q + u;
                m.row_pos[row] = p;
                m.row_ctx[row] = p + 1;
                if (p < kCondRows) { m.ekind[row] = 0; m.eidx[row] = p; m.eaux[row] = r.slot; }
                else if (p < kCondRows + T) {
                    m.ekind[row] = u == 0 ? 1 : 2; m.eidx[row] = r.text[p - kCondRows]; m.eaux[row] = p - kCondRows;
                } else { m.ekind[row] = 3; m.eidx[row] = 0; m.eaux[row] = 0; }
            }
            if (last) m.a_rows[u * n + fin] = row - 1;      // last prompt row of this stream
            ++nseq;
        }
        r.pf_pos = pt.p1;
        if (!last) continue;
        m.row_c[fin] = fin;
        m.row_u[fin] = S == 2 ? n + fin : -1;
        fill_sampling(m, fin, r);
        // repetition-penalty set starts as the prompt ids of t3.py:229-240
        m.prompt_ids[pid++] = kCondStart;
        for (int k = 0; k < kCondRows - 2; ++k) m.prompt_ids[pid++] = r.text[0];
        m.prompt_ids[pid++] = kCondEnd;
        for (int k = 0; k < T; ++k) m.prompt_ids[pid++] = r.text[k];
        m.prompt_ids[pid++] = kPrefillEnd;
        m.step[fin] = 0;
        ++fin;
    }
    m.seq_start[nseq] = row;
    // An uncond stream without attention has no history: only its last prompt row (the start-of-speech row) reaches the
    // sampler.  Those rows follow the real sequences, belong to no attention tile and keep an all-zero attention input.
    const int first_ghost = row;
    if (e->uncond_ghost)
        for (int i = 0; i < n; ++i, ++row) {
            m.row_stream[row] = -1; m.row_pos[row] = reqs_out[i]->prompt_len - 1; m.row_ctx[row] = 0;
            m.ekind[row] = 3; m.eidx[row] = 0; m.eaux[row] = 0;
            m.a_rows[n + i] = row;
        }
    const int rows = row;
    e->kphase = 1;
    // the job's own snapshot of the block table (the step of this parity two calls back has completed: its snapshot is free)
    std::memcpy(e->bt_prefill_stage[e->parity], e->block_table.h, sizeof(int32_t) * e->block_table.n);
    T3_HIP_CHECK(hipMemcpyAsync(e->bt_prefill_d, e->bt_prefill_stage[e->parity], sizeof(int32_t) * e->block_table.n,
                                hipMemcpyHostToDevice, e->cur));
    e->bt_cur = e->bt_prefill_d;
    if (upload(e, m)) return 1;
    if (stage_check(e, "prefill_upload")) return 1;
    // seen rows
    pid = 0;
    for (int i = 0; i < n; ++i) {
        const int len = reqs_out[i]->prompt_len;
        T3_RUN(e, K_SEEN, -1, (int64_t)kSeenStride, launch_seen_init(e->seen + (size_t)reqs_out[i]->slot * kSeenStride, kSeenStride,
                                                     dev_of(m, m.prompt_ids) + pid, len, e->cur));
        pid += len;
    }
    EmbedArgs ea{};
    ea.out = e->b->res; ea.d = c.d_model; ea.rows = rows;
    ea.kind = dev_of(m, m.ekind); ea.idx = dev_of(m, m.eidx); ea.aux = dev_of(m, m.eaux);
    ea.cond = e->cond_buf; ea.text_emb = e->text_emb; ea.text_pos = e->text_pos;
    ea.speech_emb = e->speech_emb; ea.speech_pos = e->speech_pos; ea.last_token = e->last_token; ea.sos_id = kSOS;
    T3_RUN(e, K_EMBED, -1, 2LL * rows * c.d_model, launch_embed(ea, e->cur));
    if (rows > first_ghost)
        T3_HIP_CHECK(hipMemsetAsync(e->b->att + (size_t)first_ghost * c.d_model, 0,
                                    sizeof(bf16_t) * (size_t)(rows - first_ghost) * c.d_model, e->cur));
    if (run_layers(e, rows, dev_of(m, m.row_pos), dev_of(m, m.row_stream), dev_of(m, m.row_ctx), true,
                   dev_of(m, m.seq_start), dev_of(m, m.tiles), ntiles,
                   share ? dev_of(m, m.seq_pfx) : nullptr, nullptr,
                   chunked ? dev_of(m, m.seq_first) : nullptr, chunked ? dev_of(m, m.seq_stream) : nullptr)) return 1;
    if (n > 0 && run_head_and_sample(e, m, n * S, dev_of(m, m.a_rows), n)) return 1;
    e->stats.prefill_requests = n;
    e->stats.prefill_rows = rows;
    e->cur = e->stream;
    e->b = &e->bufs;
    e->bt_cur = e->block_table.d;
    return 0;
}

// ---- one decode token for the running requests -------------------------------------------
int enqueue_decode(t3_engine* e, t3_engine::Meta& m, int B) {
    const t3_config& c = e->cfg;
    const int rows = B * e->rows_per_req;
    EmbedArgs ea{};
    ea.out = e->b->res; ea.d = c.d_model; ea.rows = rows;
    ea.kind = dev_of(m, m.ekind); ea.idx = dev_of(m, m.eidx); ea.aux = dev_of(m, m.eaux);
    ea.cond = e->cond_buf; ea.text_emb = e->text_emb; ea.text_pos = e->text_pos;
    ea.speech_emb = e->speech_emb; ea.speech_pos = e->speech_pos; ea.last_token = e->last_token; ea.sos_id = kSOS;
    // every decode row is the embedding of its request's last token (kind 4, idx = slot): above the few-row form the
    // first norm gathers it itself
    const bool gather_in_norm = !(e->fused_norm && rows <= GEMM_FUSED_NORM_ROWS);
    if (!gather_in_norm) T3_RUN(e, K_EMBED, -1, 2LL * rows * c.d_model, launch_embed(ea, e->cur));
    if (run_layers(e, rows, dev_of(m, m.row_pos), dev_of(m, m.row_stream), dev_of(m, m.row_ctx), false,
                   nullptr, nullptr, 0, e->share_prefix ? dev_of(m, m.row_pfx) : nullptr,
                   gather_in_norm ? dev_of(m, m.eidx) : nullptr)) return 1;
    return run_head_and_sample(e, m, rows, nullptr, B);
}

int run_decode(t3_engine* e, const std::vector<Request*>& reqs) {
    const int S = e->rows_per_req, Q = e->kv_per_req, B = (int)reqs.size();
    t3_engine::Meta& m = e->dm[e->parity];
    int64_t ctx_total = 0;
    std::vector<char> voice_met(e->n_voices, 0);
    int voices_met = 0;
    for (int i = 0; i < B; ++i) {
        Request& r = *reqs[i];
        const int pos = r.prompt_len + r.n_launched - 1;      // position of the token fed this step
        if (grow_pages(e, r, pos + 1)) return 1;
        for (int u = 0; u < S; ++u) {
            const int row = i * S + u;
            const bool ghost = u >= Q;                        // uncond row without attention: no stream, no context
            m.row_stream[row] = ghost ? -1 : r.slot * Q + u;
            m.row_pos[row] = pos;
            m.row_ctx[row] = ghost ? 0 : pos + 1;
            m.row_pfx[row] = (e->share_prefix && !ghost) ? (int32_t)(e->n_pages + r.voice) : -1;
            m.ekind[row] = 4; m.eidx[row] = r.slot; m.eaux[row] = 0;
            if (!ghost) ctx_total += pos + 1;
        }
        if (e->share_prefix && !voice_met[r.voice]) { voice_met[r.voice] = 1; ++voices_met; }
        m.row_c[i] = i * S;
        m.row_u[i] = S == 2 ? i * S + 1 : -1;
        fill_sampling(m, i, r);
    }
    // with shared prefixes the rows of a voice fetch its 34 positions once between them
    e->cur_ctx_tokens = !e->share_prefix ? ctx_total
                                         : ctx_total - (int64_t)kCondRows * B * e->kv_per_req + (int64_t)kCondRows * voices_met;
    e->bt_cur = e->block_table.d;
    if (sync_block_table(e)) return 1;
    if (upload(e, m)) return 1;
    e->kphase = 0;
    const bool plain = !e->keep_logits && !e->debug_sync && !e->ktiming;   // those modes need eager launches
    const int key = B * 2 + e->parity;                                     // graphs bake in the metadata pointers
    if (e->cfg.use_graphs && plain && e->rows_seen[key]++ >= 1) {
        auto it = e->graphs.find(key);
        if (it == e->graphs.end()) {
            hipGraph_t graph = nullptr;
            T3_HIP_CHECK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_decode(e, m, B);
            const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
            if (rc) return 1;
            T3_HIP_CHECK(ce);
            hipGraphExec_t exec = nullptr;
            T3_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            T3_HIP_CHECK(hipGraphDestroy(graph));
            it = e->graphs.emplace(key, exec).first;
        }
        T3_HIP_CHECK(hipGraphLaunch(it->second, e->stream));
    } else {
        if (enqueue_decode(e, m, B)) return 1;
    }
    e->stats.decode_requests = B;
    e->stats.decode_rows = B * S;
    e->stats.decode_ctx_tokens = ctx_total;
    return 0;
}

int find_tensor(t3_engine* e, const std::string& name, int64_t d0, int64_t d1, bf16_t** out) {
    auto it = e->raw.find(name);
    if (it == e->raw.end()) { t3_set_error("missing tensor: " + name); return 1; }
    const auto& sh = it->second.shape;
    const bool ok = d1 < 0 ? (sh.size() == 1 && sh[0] == d0) : (sh.size() == 2 && sh[0] == d0 && sh[1] == d1);
    if (!ok) { t3_set_error("bad shape for tensor: " + name); return 1; }
    *out = it->second.p;
    return 0;
}

}  // namespace

// =========================================================================================
static int engine_create(const t3_config* cfg, hipStream_t caller_stream, bool on_caller_stream, t3_engine** out);

extern "C" int t3_engine_create(const t3_config* cfg, t3_engine** out) {
    return engine_create(cfg, nullptr, false, out);
}

extern "C" int t3_engine_create_on_stream(const t3_config* cfg, void* stream, t3_engine** out) {
    return engine_create(cfg, static_cast<hipStream_t>(stream), true, out);
}

static int engine_create(const t3_config* cfg, hipStream_t caller_stream, bool on_caller_stream, t3_engine** out) {
    T3_REQUIRE(cfg && out, "null argument");
    T3_REQUIRE(cfg->d_model == 1024 && cfg->head_dim == 64 && cfg->n_heads * cfg->head_dim == cfg->d_model,
               "engine is built for d_model 1024 = 16 heads x 64");
    T3_REQUIRE(cfg->d_ff == 4096, "engine is built for d_ff 4096");
    T3_REQUIRE(cfg->n_layers >= 1 && cfg->max_num_seqs >= 1 && cfg->max_model_len >= kCondRows + 3, "bad sizes");
    T3_REQUIRE(cfg->speech_vocab + kOffset <= kSeenStride, "speech vocab too large");
    T3_HIP_CHECK(hipSetDevice(cfg->device));
    // a construction that fails half-way gives back what it had allocated
    struct Destroy { void operator()(t3_engine* p) const { (void)t3_engine_destroy(p); } };
    std::unique_ptr<t3_engine, Destroy> e(new t3_engine());
    e->cfg = *cfg;
    if (e->cfg.max_prefill_rows < e->cfg.max_model_len) e->cfg.max_prefill_rows = e->cfg.max_model_len;
    const t3_config& c = e->cfg;
    T3_REQUIRE(c.cfg_paired >= 0 && c.cfg_paired <= 2, "cfg_paired must be 0, 1 or 2");
    e->rows_per_req = c.cfg_paired ? 2 : 1;
    e->uncond_ghost = c.cfg_paired == 2;
    e->kv_per_req = c.cfg_paired == 1 ? 2 : 1;
    e->n_streams = c.max_num_seqs * e->kv_per_req;
    e->n_rows = c.max_num_seqs * e->rows_per_req;
    e->max_pages = pages_for(c.max_model_len);
    e->head_n = (c.speech_vocab + 15) / 16 * 16;
    if (on_caller_stream) {
        e->stream = caller_stream; e->stream_owned = false;
        if (caller_stream == nullptr) e->cfg.use_graphs = 0;      // the default stream cannot be captured
    }
    else T3_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    T3_HIP_CHECK(hipStreamCreateWithFlags(&e->pstream_own, hipStreamNonBlocking));
    e->pstream = e->pstream_own;
    e->cur = e->stream;
    e->layers.resize(c.n_layers);
    e->slots.resize(c.max_num_seqs);
    const char* dbg = getenv("T3_DEBUG_SYNC");
    e->debug_sync = dbg && dbg[0] == '1';

    // KV pool
    e->n_pages = c.kv_pages > 0 ? c.kv_pages : (int64_t)e->n_streams * e->max_pages;
    T3_REQUIRE(e->n_pages >= (int64_t)e->kv_per_req * e->max_pages, "kv_pages cannot hold one full request");
    // shared prefixes: one page each behind the request pages; more entries than slots, so that one is always free
    T3_REQUIRE(c.prefix_sharing == 0 || c.prefix_sharing == 1, "prefix_sharing must be 0 or 1");
    T3_REQUIRE(c.max_voices >= 0, "max_voices must not be negative");
    e->share_prefix = c.prefix_sharing != 0;
    e->n_voices = !e->share_prefix ? 0 : std::max(c.max_voices, c.max_num_seqs + 1);
    e->voices.resize(e->n_voices);
    const int64_t page_elems = 2LL * c.n_heads * T3_PAGE * c.head_dim;
    e->layer_stride = (e->n_pages + e->n_voices) * page_elems;
    const size_t pool_elems = (size_t)e->layer_stride * c.n_layers;
    if (dmalloc(&e->kv_pool, pool_elems)) return 1;
    T3_HIP_CHECK(hipMemsetAsync(e->kv_pool, 0, pool_elems * sizeof(bf16_t), e->stream));
    e->free_pages.reserve(e->n_pages);
    for (int64_t p = e->n_pages - 1; p >= 0; --p) e->free_pages.push_back((int32_t)p);
    if (e->block_table.alloc((size_t)(e->n_streams + e->n_voices) * e->max_pages)) return 1;
    for (int v = 0; v < e->n_voices; ++v)
        e->block_table.h[(size_t)(e->n_streams + v) * e->max_pages] = (int32_t)(e->n_pages + v);
    T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->bt_prefill_d), sizeof(int32_t) * e->block_table.n));
    for (int q = 0; q < 2; ++q)
        T3_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->bt_prefill_stage[q]), sizeof(int32_t) * e->block_table.n));
    e->bt_cur = e->block_table.d;

    // activations
    const int prefill_rows = c.max_prefill_rows * e->kv_per_req + (e->uncond_ghost ? c.max_num_seqs : 0);
    e->max_rows = (std::max(prefill_rows, e->n_rows) + 127) / 128 * 128;   // whole row tiles for the tile-ordered buffers
    for (int which = 0; which < 2; ++which) {             // decode rows, prompt rows
        const size_t R = (size_t)((which == 0 ? e->n_rows : prefill_rows) + 127) / 128 * 128, RD = (size_t)e->n_rows;
        t3_engine::Bufs& b = which == 0 ? e->bufs : e->pbufs;
        if (dmalloc(&b.res, R * c.d_model) || dmalloc(&b.xn, R * c.d_model) || dmalloc(&b.qkv, R * 3 * c.d_model) ||
            dmalloc(&b.att, R * c.d_model) || dmalloc(&b.hbuf, R * c.d_ff) || dmalloc(&b.logits, RD * e->head_n) ||
            dmalloc(&b.res2, (size_t)16 * c.d_model)) return 1;
        T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&b.att_states),
                               sizeof(float) * GEMM_FUSED_NORM_ROWS * c.n_heads * 8 * 8 * ATT_STATE_FLOATS));
        T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&b.slab), sizeof(float) * ATTN_OPROJ_SPLIT * R * c.d_model));   // 16 head slabs >= 8 split-K slabs
    }
    e->b = &e->bufs;
    if (const char* ug = getenv("T3_USE_GRAPHS")) e->cfg.use_graphs = atoi(ug) != 0;
    if (const char* as = getenv("T3_ASYNC")) e->async_steps = atoi(as) != 0;
    if (const char* ov = getenv("T3_OVERLAP_PREFILL")) e->overlap_prefill = atoi(ov) != 0;
    if (const char* fn = getenv("T3_FUSED_NORM")) e->fused_norm = atoi(fn) != 0;
    if (const char* fo = getenv("T3_FUSED_OPROJ")) e->fused_oproj = atoi(fo) != 0;
    if (const char* sa = getenv("T3_SPLIT_ATTN")) e->split_attn = atoi(sa) != 0;
    if (const char* gm = getenv("T3_GROUP_MIN_ROWS")) e->group_min_rows = std::max(1, atoi(gm));
    if (const char* gr = getenv("T3_GROUP_ROWS")) { const int v = atoi(gr); if (v == 1 || v == 2 || v == 4) e->group_rows = v; }
    for (int q = 0; q < 2; ++q) {
        T3_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&e->bt_stage[q]), sizeof(int32_t) * e->block_table.n));
        T3_HIP_CHECK(hipEventCreateWithFlags(&e->step_done[q], hipEventDisableTiming));
        T3_HIP_CHECK(hipEventCreateWithFlags(&e->fork_ev[q], hipEventDisableTiming));

    }
    T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->cond_buf),
                           sizeof(float) * std::max(c.max_num_seqs, e->n_voices) * kCondRows * c.d_model));
    T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->seen), (size_t)c.max_num_seqs * kSeenStride));
    T3_HIP_CHECK(hipMemsetAsync(e->seen, 0, (size_t)c.max_num_seqs * kSeenStride, e->stream));
    T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->last_token), sizeof(int32_t) * c.max_num_seqs));
    T3_HIP_CHECK(hipMemsetAsync(e->last_token, 0, sizeof(int32_t) * c.max_num_seqs, e->stream));
    T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&e->dbg_logits),
                           sizeof(float) * c.max_num_seqs * (size_t)(c.speech_vocab + kOffset)));
    for (int q = 0; q < 2; ++q) {
        if (meta_alloc(e->dm[q], e->n_rows, c.max_num_seqs, 16)) return 1;
        if (meta_alloc(e->pm[q], prefill_rows, c.max_num_seqs,
                       (size_t)c.max_prefill_rows + (size_t)c.max_model_len + (size_t)c.max_num_seqs * 64)) return 1;
    }
    for (auto& ev : e->ev) T3_HIP_CHECK(hipEventCreate(&ev));
    T3_HIP_CHECK(hipEventCreateWithFlags(&e->pjob_done, hipEventDisableTiming));
    T3_HIP_CHECK(hipStreamSynchronize(e->stream));
    *out = e.release();
    return 0;
}

static int load_tensor(t3_engine* e, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t dtype,
                       hipMemcpyKind kind);

extern "C" int t3_engine_load_tensor(t3_engine* e, const char* name, const void* data, const int64_t* shape,
                                     int32_t ndim, int32_t dtype) {
    return load_tensor(e, name, data, shape, ndim, dtype, hipMemcpyHostToDevice);
}

extern "C" int t3_engine_load_tensor_dev(t3_engine* e, const char* name, const void* data_dev, const int64_t* shape,
                                         int32_t ndim, int32_t dtype) {
    return load_tensor(e, name, data_dev, shape, ndim, dtype, hipMemcpyDeviceToDevice);
}

static int load_tensor(t3_engine* e, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t dtype,
                       hipMemcpyKind kind) {
    T3_REQUIRE(e && name && data && shape, "null argument");
    T3_REQUIRE(!e->finalized, "engine already finalized");
    T3_REQUIRE(dtype == 0, "only bf16 (dtype 0) tensors are accepted");
    T3_REQUIRE(ndim == 1 || ndim == 2, "tensors must be 1-D or 2-D");
    T3_HIP_CHECK(hipSetDevice(e->cfg.device));
    size_t n = 1;
    std::vector<int64_t> sh(shape, shape + ndim);
    for (auto v : sh) { T3_REQUIRE(v > 0, "bad shape"); n *= (size_t)v; }
    bf16_t* p = nullptr;
    if (dmalloc(&p, n)) return 1;
    T3_HIP_CHECK(hipMemcpy(p, data, n * sizeof(bf16_t), kind));    // blocking: the source is free when the call returns
    auto it = e->raw.find(name);
    if (it != e->raw.end()) { (void)hipFree(it->second.p); e->raw.erase(it); }
    e->raw[name] = t3_engine::Raw{p, sh};
    return 0;
}

extern "C" int t3_engine_finalize(t3_engine* e) {
    T3_REQUIRE(e && !e->finalized, "bad engine state");
    T3_HIP_CHECK(hipSetDevice(e->cfg.device));
    const t3_config& c = e->cfg;
    const int D = c.d_model, F = c.d_ff;
    hipStream_t st = e->stream;
    if (find_tensor(e, "text_emb.weight", c.text_vocab, D, &e->text_emb)) return 1;
    if (find_tensor(e, "speech_emb.weight", c.speech_vocab, D, &e->speech_emb)) return 1;
    if (find_tensor(e, "text_pos_emb.emb.weight", c.text_pos_rows, D, &e->text_pos)) return 1;
    if (find_tensor(e, "speech_pos_emb.emb.weight", c.speech_pos_rows, D, &e->speech_pos)) return 1;
    if (find_tensor(e, "tfmr.norm.weight", D, -1, &e->norm_w)) return 1;
    {
        auto it = e->raw.find("rope.cos_sin");
        T3_REQUIRE(it != e->raw.end(), "missing tensor: rope.cos_sin");
        T3_REQUIRE(it->second.shape.size() == 2 && it->second.shape[1] == c.head_dim &&
                   it->second.shape[0] >= c.max_model_len, "rope.cos_sin must be [>= max_model_len, head_dim]");
        e->rope = it->second.p; e->rope_rows = (int)it->second.shape[0];
    }
    std::vector<std::string> consumed;
    // speech head, zero-padded to a multiple of 16 rows, tiled
    {
        bf16_t* hw = nullptr;
        if (find_tensor(e, "speech_head.weight", c.speech_vocab, D, &hw)) return 1;
        bf16_t* padded = nullptr;
        if (dmalloc(&padded, (size_t)e->head_n * D) || dmalloc(&e->head, (size_t)e->head_n * D)) return 1;
        T3_HIP_CHECK(hipMemsetAsync(padded, 0, (size_t)e->head_n * D * sizeof(bf16_t), st));
        T3_HIP_CHECK(hipMemcpyAsync(padded, hw, (size_t)c.speech_vocab * D * sizeof(bf16_t), hipMemcpyDeviceToDevice, st));
        if (launch_repack(padded, e->head, e->head_n, D, st)) return 1;
        T3_HIP_CHECK(hipStreamSynchronize(st));
        (void)hipFree(padded);
        consumed.push_back("speech_head.weight");
    }
    bf16_t* cat = nullptr;
    if (dmalloc(&cat, (size_t)2 * F * D)) return 1;       // staging for fused [q|k|v] and [gate|up]
    for (int li = 0; li < c.n_layers; ++li) {
        Layer& L = e->layers[li];
        const std::string p = "tfmr.layers." + std::to_string(li) + ".";
        bf16_t *q, *k, *v, *o, *g, *u, *d;
        if (find_tensor(e, p + "input_layernorm.weight", D, -1, &L.ln1)) return 1;
        if (find_tensor(e, p + "post_attention_layernorm.weight", D, -1, &L.ln2)) return 1;
        if (find_tensor(e, p + "self_attn.q_proj.weight", D, D, &q) || find_tensor(e, p + "self_attn.k_proj.weight", D, D, &k) ||
            find_tensor(e, p + "self_attn.v_proj.weight", D, D, &v) || find_tensor(e, p + "self_attn.o_proj.weight", D, D, &o) ||
            find_tensor(e, p + "mlp.gate_proj.weight", F, D, &g) || find_tensor(e, p + "mlp.up_proj.weight", F, D, &u) ||
            find_tensor(e, p + "mlp.down_proj.weight", D, F, &d)) return 1;
        // a matrix in its streaming order: bf16 tiles, or (weight_int8) int8 pair tiles + one scale per output column
        const bool w8 = c.weight_int8 != 0;
        auto place = [&](const bf16_t* src, bf16_t** wt, float** sc, int N, int K) -> int {
            if (!w8) {
                if (dmalloc(wt, (size_t)N * K)) return 1;
                return launch_repack(src, *wt, N, K, st);
            }
            T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(wt), (size_t)N * K));
            T3_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(sc), sizeof(float) * N));
            return launch_quantize(src, reinterpret_cast<uint8_t*>(*wt), *sc, N, K, st);
        };
        const size_t dd = (size_t)D * D * sizeof(bf16_t), fd = (size_t)F * D * sizeof(bf16_t);
        T3_HIP_CHECK(hipMemcpyAsync(cat, q, dd, hipMemcpyDeviceToDevice, st));
        T3_HIP_CHECK(hipMemcpyAsync(cat + (size_t)D * D, k, dd, hipMemcpyDeviceToDevice, st));
        T3_HIP_CHECK(hipMemcpyAsync(cat + (size_t)2 * D * D, v, dd, hipMemcpyDeviceToDevice, st));
        if (place(cat, &L.wqkv, &L.sqkv, 3 * D, D)) return 1;
        if (place(o, &L.wo, &L.so, D, D)) return 1;
        T3_HIP_CHECK(hipMemcpyAsync(cat, g, fd, hipMemcpyDeviceToDevice, st));
        T3_HIP_CHECK(hipMemcpyAsync(cat + (size_t)F * D, u, fd, hipMemcpyDeviceToDevice, st));
        if (place(cat, &L.wgu, &L.sgu, 2 * F, D)) return 1;
        if (place(d, &L.wd, &L.sd, D, F)) return 1;
        for (const char* s : {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                              "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight"})
            consumed.push_back(p + s);
    }
    T3_HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(cat);
    for (const auto& nme : consumed) {
        auto it = e->raw.find(nme);
        if (it != e->raw.end()) { (void)hipFree(it->second.p); e->raw.erase(it); }
    }
    e->finalized = true;
    return 0;
}

extern "C" int t3_engine_destroy(t3_engine* e) {
    if (!e) return 0;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream || !e->stream_owned) (void)hipStreamSynchronize(e->stream);
    if (e->pstream_own) (void)hipStreamSynchronize(e->pstream_own);
    for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
    for (auto& kv : e->raw) (void)hipFree(kv.second.p);
    for (auto& L : e->layers) {
        (void)hipFree(L.wqkv); (void)hipFree(L.wo); (void)hipFree(L.wgu); (void)hipFree(L.wd);
        (void)hipFree(L.sqkv); (void)hipFree(L.so); (void)hipFree(L.sgu); (void)hipFree(L.sd);
    }
    (void)hipFree(e->head); (void)hipFree(e->kv_pool);
    for (auto* bp : {&e->bufs, &e->pbufs}) {
        auto& b = *bp;
        (void)hipFree(b.res); (void)hipFree(b.xn); (void)hipFree(b.qkv); (void)hipFree(b.att); (void)hipFree(b.hbuf);
        (void)hipFree(b.logits); (void)hipFree(b.slab); (void)hipFree(b.res2); (void)hipFree(b.att_states);
    }
    (void)hipFree(e->cond_buf); (void)hipFree(e->seen); (void)hipFree(e->last_token); (void)hipFree(e->dbg_logits);
    for (int q = 0; q < 2; ++q) {
        if (e->bt_stage[q]) (void)hipHostFree(e->bt_stage[q]);
        if (e->step_done[q]) (void)hipEventDestroy(e->step_done[q]);
        if (e->fork_ev[q]) (void)hipEventDestroy(e->fork_ev[q]);

    }
    e->block_table.release();
    (void)hipFree(e->bt_prefill_d);
    for (int q = 0; q < 2; ++q) if (e->bt_prefill_stage[q]) (void)hipHostFree(e->bt_prefill_stage[q]);
    for (auto* m : {&e->dm[0], &e->dm[1], &e->pm[0], &e->pm[1]}) {
        m->blk.release();
        if (m->out_tokens) (void)hipHostFree(m->out_tokens);
    }
    for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->pjob_done) (void)hipEventDestroy(e->pjob_done);
    for (auto& ev : e->kpool) (void)hipEventDestroy(ev);
    if (e->stream && e->stream_owned) (void)hipStreamDestroy(e->stream);
    if (e->pstream_own) (void)hipStreamDestroy(e->pstream_own);
    delete e;
    return 0;
}

extern "C" int t3_engine_add_request(t3_engine* e, int64_t req_id, const int32_t* text_ids, int32_t n_text,
                                     const float* cond, const t3_sampling* sp) {
    T3_REQUIRE(e && text_ids && cond && sp, "null argument");
    T3_REQUIRE(e->finalized, "engine not finalized");
    const t3_config& c = e->cfg;
    T3_REQUIRE(n_text >= 1, "empty prompt");
    T3_REQUIRE(n_text <= c.text_pos_rows, "prompt longer than the text position table");
    const int P = kCondRows + n_text + 1;
    if (P > c.max_model_len) {
        t3_set_error("prompt of " + std::to_string(P) + " tokens exceeds max_model_len=" + std::to_string(c.max_model_len));
        return 2;
    }
    T3_REQUIRE(P <= c.max_prefill_rows, "prompt longer than max_prefill_rows");
    for (int i = 0; i < n_text; ++i)
        T3_REQUIRE(text_ids[i] >= 0 && text_ids[i] < c.text_vocab, "text id outside the text vocabulary");
    T3_REQUIRE(sp->max_tokens >= 1 && sp->repetition_penalty > 0.f && sp->n_extra_stop >= 0 && sp->n_extra_stop <= 4,
               "bad sampling parameters");
    for (auto& w : e->waiting) T3_REQUIRE(w->id != req_id, "duplicate request id");
    for (auto& s : e->slots) T3_REQUIRE(!s || s->id != req_id, "duplicate request id");
    std::unique_ptr<Request> r(new Request());
    r->id = req_id;
    r->text.assign(text_ids, text_ids + n_text);
    r->cond.assign(cond, cond + (size_t)kCondRows * c.d_model);
    r->sp = *sp;
    r->prompt_len = P;
    r->limit = std::max(1, std::min(sp->max_tokens, c.max_model_len - P));
    r->reserve_pages = pages_for(std::min(P + sp->max_tokens, c.max_model_len));
    if (e->schedule_policy != 0) {
        // longest first: a queue that is known up front drains with fewer half-empty steps at its end.  Stable
        // for equal sizes; which requests share a step never changes a request's tokens.  Policy 2 ranks by the
        // prompt alone (what a caller knows: longer text, longer speech); policy 1 also reads max_tokens, which is
        // a ranking by true length only when the caller sets max_tokens per request to the length it expects.
        auto work = [&](const Request& q) { return e->schedule_policy == 2 ? q.prompt_len : q.prompt_len + q.limit; };
        const int mine = work(*r);
        auto it = e->waiting.begin();
        while (it != e->waiting.end() && work(**it) >= mine) ++it;
        e->waiting.insert(it, std::move(r));
    } else {
        e->waiting.push_back(std::move(r));
    }
    return 0;
}

extern "C" int t3_engine_set_schedule_policy(t3_engine* e, int32_t policy) {
    T3_REQUIRE(e, "null argument");
    T3_REQUIRE(policy >= 0 && policy <= 2,
               "policy must be 0 (first come first served), 1 (longest prompt + max_tokens first) or 2 (longest prompt first)");
    e->schedule_policy = policy;
    return 0;
}

extern "C" int t3_engine_set_prefill_chunk(t3_engine* e, int32_t positions) {
    T3_REQUIRE(e && e->finalized, "engine not finalized");
    T3_REQUIRE(positions >= 0, "the prefill budget must be 0 (whole prompts) or a number of positions");
    for (auto& s : e->slots) T3_REQUIRE(!s, "the prefill budget can only change while no request is admitted");
    e->prefill_chunk = positions;
    return 0;
}

extern "C" int t3_engine_abort(t3_engine* e, int64_t req_id) {
    T3_REQUIRE(e, "null argument");
    for (auto it = e->waiting.begin(); it != e->waiting.end(); ++it)
        if ((*it)->id == req_id) { e->waiting.erase(it); return 0; }
    for (auto& s : e->slots)
        if (s && s->id == req_id && !s->zombie) {
            if (s->n_launched > s->n_out || s->chunks_out > 0) { s->zombie = true; s->aborted = true; }   // a launched step still uses its pages
            else { release_request(e, *s); s.reset(); }
            return 0;
        }
    t3_set_error("abort: unknown request id");
    return 1;
}

extern "C" int t3_engine_has_unfinished(t3_engine* e) {
    if (!e) return 0;
    if (!e->waiting.empty() || e->inflight.valid || e->pjob.valid) return 1;
    for (auto& s : e->slots) if (s) return 1;
    return 0;
}

extern "C" int t3_engine_get_stats(t3_engine* e, t3_step_stats* out) {
    T3_REQUIRE(e && out, "null argument");
    *out = e->stats;
    return 0;
}

extern "C" int t3_engine_set_timing(t3_engine* e, int32_t on) {
    T3_REQUIRE(e, "null argument");
    e->timing = on != 0;
    return 0;
}

extern "C" int t3_engine_set_kernel_timing(t3_engine* e, int32_t on) {
    T3_REQUIRE(e, "null argument");
    e->ktiming = on != 0;
    if (on) {
        std::memset(e->kms, 0, sizeof(e->kms));
        std::memset(e->klaunches, 0, sizeof(e->klaunches));
        std::memset(e->kbytes, 0, sizeof(e->kbytes));
    }
    return 0;
}

extern "C" int t3_kernel_class_count(void) { return K_NCLASS; }
extern "C" const char* t3_kernel_class_name(int32_t cls) { return cls >= 0 && cls < K_NCLASS ? kClassName[cls] : ""; }

extern "C" int t3_engine_get_kernel_times(t3_engine* e, int32_t phase, double* ms, int64_t* launches, int64_t* bytes,
                                          int32_t n) {
    T3_REQUIRE(e && ms && launches && bytes, "null argument");
    T3_REQUIRE((phase == 0 || phase == 1) && n >= K_NCLASS, "bad phase or array too small");
    for (int i = 0; i < K_NCLASS; ++i) { ms[i] = e->kms[phase][i]; launches[i] = e->klaunches[phase][i]; bytes[i] = e->kbytes[phase][i]; }
    return 0;
}

extern "C" int t3_engine_set_keep_logits(t3_engine* e, int32_t on) {
    T3_REQUIRE(e, "null argument");
    e->keep_logits = on != 0;
    return 0;
}

extern "C" int t3_engine_last_logits(t3_engine* e, int64_t req_id, float* out) {
    T3_REQUIRE(e && out, "null argument");
    T3_REQUIRE(e->keep_logits, "logit taps are off (t3_engine_set_keep_logits)");
    for (auto& s : e->slots)
        if (s && s->id == req_id && !s->zombie) {
            const size_t n = (size_t)e->cfg.speech_vocab + kOffset;
            T3_HIP_CHECK(hipMemcpy(out, e->dbg_logits + (size_t)s->slot * n, n * sizeof(float), hipMemcpyDeviceToHost));
            return 0;
        }
    t3_set_error("last_logits: request is not running");
    return 1;
}

namespace {

// One computed token of request r: count it, apply the finish rules, free the request's slot and pages when it ends.
void account_token(t3_engine* e, Request* r, int32_t tok, int64_t* out_req_ids, int32_t* out_tokens, int32_t* out_finished,
                   int& k) {
    if (r->zombie) {                                          // its token is past a stop or an abort: drop it
        release_request(e, *r); e->slots[r->slot].reset();
        return;
    }
    r->n_out++;
    int fin = 0;
    if (tok < 0) { fin = 3; r->poisoned = true; }             // the sampler found no finite logit
    else if (tok == r->sp.stop_token) fin = 1;
    else for (int j = 0; j < r->sp.n_extra_stop; ++j) if (tok == r->sp.extra_stop_tokens[j]) fin = 1;
    if (!fin && r->n_out >= r->limit) fin = 2;
    out_req_ids[k] = r->id; out_tokens[k] = tok; out_finished[k] = fin; ++k;
    if (!fin) return;
    if (r->n_launched > r->n_out) r->zombie = true;           // the next step already carries it
    else { release_request(e, *r); e->slots[r->slot].reset(); }
}

// The job that carried a chunk of these prompts has completed: they may be scheduled again, and one that was aborted
// meanwhile gives its slot and pages back now.
void settle_partial(t3_engine* e, std::vector<Request*>& partial) {
    for (Request* r : partial) {
        r->prefilling = false;
        if (--r->chunks_out == 0 && r->zombie) { release_request(e, *r); e->slots[r->slot].reset(); }
    }
    partial.clear();
}

// Wait for the outstanding prefill job and report the first token of each of its requests; they may decode from now on.
int report_prefill_job(t3_engine* e, int64_t* out_req_ids, int32_t* out_tokens, int32_t* out_finished, int32_t cap,
                       int32_t* n_out) {
    t3_engine::PrefillJob& j = e->pjob;
    if (!j.valid) return 0;
    T3_REQUIRE(*n_out + (int)j.reqs.size() <= cap, "output arrays too small");
    T3_HIP_CHECK(hipEventSynchronize(e->pjob_done));
    j.valid = false;
    int k = *n_out;
    const int32_t* pt = e->pm[j.parity].out_tokens;
    for (size_t i = 0; i < j.reqs.size(); ++i) {
        j.reqs[i]->prefilling = false;
        account_token(e, j.reqs[i], pt[i], out_req_ids, out_tokens, out_finished, k);
    }
    j.reqs.clear();
    settle_partial(e, j.partial);
    *n_out = k;
    return 0;
}

// Wait for the launched step and report it: one (request, token, finish reason) triple per row it carried.
int report_inflight(t3_engine* e, int64_t* out_req_ids, int32_t* out_tokens, int32_t* out_finished, int32_t cap,
                    int32_t* n_out) {
    t3_engine::InFlight& f = e->inflight;
    if (!f.valid) return 0;
    T3_REQUIRE(*n_out + (int)(f.admitted.size() + f.running.size()) <= cap, "output arrays too small");
    T3_HIP_CHECK(hipEventSynchronize(e->step_done[f.parity]));
    f.valid = false;
    if (e->ktiming && ktime_collect(e)) return 1;
    if (e->timing) {
        if (!f.admitted.empty()) T3_HIP_CHECK(hipEventElapsedTime(&e->stats.prefill_gpu_ms, e->ev[0], e->ev[1]));
        if (!f.running.empty()) T3_HIP_CHECK(hipEventElapsedTime(&e->stats.decode_gpu_ms, e->ev[2], e->ev[3]));
    }
    int k = *n_out;
    const int32_t* pt = e->pm[f.parity].out_tokens;
    const int32_t* dt = e->dm[f.parity].out_tokens;
    for (size_t i = 0; i < f.admitted.size(); ++i) account_token(e, f.admitted[i], pt[i], out_req_ids, out_tokens, out_finished, k);
    for (size_t i = 0; i < f.running.size(); ++i) account_token(e, f.running[i], dt[i], out_req_ids, out_tokens, out_finished, k);
    settle_partial(e, f.partial);
    *n_out = k;
    return 0;
}

}  // namespace

extern "C" int t3_engine_step(t3_engine* e, int64_t* out_req_ids, int32_t* out_tokens, int32_t* out_finished,
                              int32_t cap, int32_t* n_out) {
    T3_REQUIRE(e && out_req_ids && out_tokens && out_finished && n_out, "null argument");
    T3_REQUIRE(e->finalized, "engine not finalized");
    T3_HIP_CHECK(hipSetDevice(e->cfg.device));
    const t3_config& c = e->cfg;
    *n_out = 0;
    std::memset(&e->stats, 0, sizeof(e->stats));
    const bool sync_mode = !e->async_steps || e->timing || e->ktiming || e->keep_logits || e->debug_sync;
    // a synchronous step must not be planned on stale knowledge: report the launched one first
    if (sync_mode && report_inflight(e, out_req_ids, out_tokens, out_finished, cap, n_out)) return 1;

    // A prefill job on the second stream: report it once it has completed (or when the GPU has nothing else to do);
    // a switch to a synchronous mode finishes it first.
    const bool overlap = !sync_mode && e->overlap_prefill;
    if (e->pjob.valid) {
        bool report = !overlap;
        if (!report) {
            const hipError_t qs = hipEventQuery(e->pjob_done);
            if (qs != hipSuccess && qs != hipErrorNotReady) T3_HIP_CHECK(qs);
            if (qs == hipErrorNotReady) (void)hipGetLastError();   // "not ready" is an answer, not an error to find after a launch
            report = qs == hipSuccess;
            if (!report) {                                    // nothing decoding and nothing in flight: wait for it
                bool busy = e->inflight.valid;
                for (auto& s : e->slots) busy = busy || (s && !s->prefilling && !s->zombie && s->n_launched < s->limit);
                report = !busy;
            }
        }
        if (report && report_prefill_job(e, out_req_ids, out_tokens, out_finished, cap, n_out)) return 1;
    }

    // requests with tokens still to launch decode; new ones prefill
    std::vector<Request*> running, admitted;
    for (auto& s : e->slots)
        if (s && !s->zombie && !s->prefilling && s->pf_pos >= s->prompt_len && s->n_launched < s->limit) running.push_back(s.get());
    std::sort(running.begin(), running.end(), [](const Request* a, const Request* b) {
        const int ca = a->prompt_len + a->n_launched, cb = b->prompt_len + b->n_launched;
        return ca != cb ? ca > cb : a->slot < b->slot;
    });
    // The grouped attention form gives one workgroup four consecutive rows: pair the longest request with the shortest, the
    // second longest with the second shortest, ... so that every group holds about the same number of positions (a row's
    // tokens do not depend on its place in the batch).
    if (e->fused_oproj && (int)running.size() * e->rows_per_req >= e->group_min_rows) {
        std::vector<Request*> paired;
        paired.reserve(running.size());
        for (size_t lo = 0, hi = running.size(); lo < hi;) {
            paired.push_back(running[lo++]);
            if (lo < hi) paired.push_back(running[--hi]);
        }
        running.swap(paired);
    }
    // What this step's prefill job carries.  Whole prompts: new requests while the row budget holds their prompts.  Chunked
    // (prefill_chunk > 0, the token budget of vLLM's scheduler): prompts already under way come first, then new requests,
    // and the prompt that does not fit is cut at the budget; it goes on in the next job.
    const bool chunked = e->prefill_chunk > 0;
    const int first_pos = e->share_prefix ? kCondRows : 0;
    // two limits: the token budget (positions per stream a chunked job may carry) and the room of the job's buffers
    // (max_prefill_rows positions per stream; the 34 rows of a voice that a request's first chunk builds count against it,
    // and whole prompts count their 34 prefix positions although a shared prefix leaves them out of the rows)
    int chunk_left = chunked ? e->prefill_chunk : c.max_prefill_rows;
    int rows_left = c.max_prefill_rows;
    std::vector<PrefillPart> parts;
    std::vector<Request*> partial;
    if (chunked && !e->pjob.valid)
        for (auto& s : e->slots) {
            Request* r = s.get();
            if (!r || r->zombie || r->prefilling || r->pf_pos >= r->prompt_len) continue;
            const int take = std::min(r->prompt_len - r->pf_pos, std::min(chunk_left, rows_left));
            if (take <= 0) break;
            chunk_left -= take; rows_left -= take;
            parts.push_back({r, r->pf_pos, r->pf_pos + take});
        }
    while (!e->waiting.empty() && !e->pjob.valid) {
        Request& r = *e->waiting.front();
        int slot = -1;
        for (int i = 0; i < c.max_num_seqs; ++i) if (!e->slots[i]) { slot = i; break; }
        if (slot < 0) break;
        const int64_t need = (int64_t)r.reserve_pages * e->kv_per_req;
        if ((int64_t)e->free_pages.size() - e->reserved_unallocated < need) break;
        if (chunked ? (chunk_left <= 0 || rows_left <= first_pos) : r.prompt_len > rows_left) break;
        e->reserved_unallocated += need;
        r.slot = slot;
        r.pf_pos = first_pos;
        if (e->share_prefix && acquire_voice(e, r)) return 1;
        int take = r.prompt_len - first_pos;
        if (chunked) {
            if (e->share_prefix && r.builds_voice) rows_left -= kCondRows;
            take = std::min(take, std::min(chunk_left, rows_left));       // >= 1: rows_left was above 34
            chunk_left -= take; rows_left -= take;
        } else {
            rows_left -= r.prompt_len;
        }
        parts.push_back({&r, first_pos, first_pos + take});
        e->slots[slot] = std::move(e->waiting.front());
        e->waiting.pop_front();
    }
    for (const PrefillPart& pt : parts) if (pt.p1 < pt.r->prompt_len) partial.push_back(pt.r);

    const int q = e->parity;
    e->pstream = overlap ? e->pstream_own : e->stream;     // no job is outstanding when this changes
    if (!parts.empty()) {
        if (e->timing) T3_HIP_CHECK(hipEventRecord(e->ev[0], e->stream));
        if (run_prefill(e, parts, admitted)) return 1;     // admitted: the requests whose prompt ends in this job
        if (e->timing) T3_HIP_CHECK(hipEventRecord(e->ev[1], e->stream));
        for (Request* r : admitted) r->n_launched = 1;
        for (Request* r : partial) r->chunks_out++;
        if (overlap) {                                     // a job of its own: reported by a later call, see above
            T3_HIP_CHECK(hipEventRecord(e->pjob_done, e->pstream));
            for (Request* r : admitted) r->prefilling = true;
            for (Request* r : partial) r->prefilling = true;
            e->pjob.valid = true;
            e->pjob.parity = q;
            e->pjob.reqs = std::move(admitted);
            e->pjob.partial = std::move(partial);
            admitted.clear();
            partial.clear();
        }
    }
    if (!running.empty()) {
        if (e->timing) T3_HIP_CHECK(hipEventRecord(e->ev[2], e->stream));
        if (run_decode(e, running)) return 1;
        if (e->timing) T3_HIP_CHECK(hipEventRecord(e->ev[3], e->stream));
        for (Request* r : running) r->n_launched++;
    }
    const bool launched = !admitted.empty() || !partial.empty() || !running.empty();
    if (launched) T3_HIP_CHECK(hipEventRecord(e->step_done[q], e->stream));

    // report the previous step while this one runs (in the synchronous modes it was reported above)
    if (report_inflight(e, out_req_ids, out_tokens, out_finished, cap, n_out)) return 1;
    if (launched) {
        e->inflight.valid = true;
        e->inflight.parity = q;
        e->inflight.admitted = std::move(admitted);
        e->inflight.partial = std::move(partial);
        e->inflight.running = std::move(running);
        e->parity ^= 1;
        if (sync_mode && report_inflight(e, out_req_ids, out_tokens, out_finished, cap, n_out)) return 1;
    }
    e->stats.prefill_outstanding = e->pjob.valid ? (int32_t)e->pjob.reqs.size() : 0;
    e->stats.free_pages = (int32_t)e->free_pages.size();
    e->stats.waiting = (int32_t)e->waiting.size();
    int run = 0;
    for (auto& s : e->slots) if (s) ++run;
    e->stats.running = run;
    return 0;
}
