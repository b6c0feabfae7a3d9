// Shared device helpers and the internal launcher interface of the T3 engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>

typedef unsigned short bf16_t;   // storage type: raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define T3_PAGE 64               // tokens per KV page
#define T3_WAVE 64
#define T3_MAX_DEVICES 64        // device ordinals the per-device tables of the library cover

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned)x) << 16); }
// round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }   // round through bf16
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

// ---- the residual-add + RMSNorm arithmetic of one thread (8 consecutive columns of a 1024-wide row) ---------
// Shared by the norm kernel and by the projection kernels that do the norm in their prologue for launches of a few
// rows: both must produce the same bits, so both run exactly these operations in exactly this order.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// z = the thread's 8 residual values; folds the split-K partial sums in: z = bf16(bf16(slab[0] + slab[1] + ...) + z)
// (loads and sums are separate so that a caller can put other loads between them)
template <int NSPLIT>
__device__ __forceinline__ void norm_load8(const float* __restrict__ sp, size_t slab_stride, f32x4 (&p)[NSPLIT > 0 ? NSPLIT : 1][2]) {
#pragma unroll
    for (int s = 0; s < NSPLIT; ++s)                       // all partial loads in flight at once
#pragma unroll
        for (int v = 0; v < 2; ++v) p[s][v] = *reinterpret_cast<const f32x4*>(sp + s * slab_stride + 4 * v);
}
// NSPLIT == ATTN_OPROJ_SPLIT (16): the partials are o_proj's per-head sums written by the attention launch; heads 2s and
// 2s+1 together are what split-K slab s of the stand-alone projection holds, so they are added first (and, for int8 weights,
// scaled by the column's weight scale, which that projection applies to its slab) and the eight pair sums then fold like
// eight slabs: the same additions in the same order, hence the same bits as the stand-alone projection's eight slabs.
template <int NSPLIT>
__device__ __forceinline__ void norm_sum8(const f32x4 (&p)[NSPLIT > 0 ? NSPLIT : 1][2], float z[8], const float* __restrict__ pair_scale = nullptr) {
    if constexpr (NSPLIT > 0) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            f32x4 a;
            if constexpr (NSPLIT == 16) {
                f32x4 sc = {1.f, 1.f, 1.f, 1.f};
                if (pair_scale) sc = *reinterpret_cast<const f32x4*>(pair_scale + 4 * v);
                a = p[0][v] + p[1][v];
                if (pair_scale) a *= sc;
#pragma unroll
                for (int s = 1; s < 8; ++s) {
                    f32x4 b = p[2 * s][v] + p[2 * s + 1][v];
                    if (pair_scale) b *= sc;
                    a += b;
                }
            } else {
                a = p[0][v];
#pragma unroll
                for (int s = 1; s < NSPLIT; ++s) a += p[s][v];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) z[4 * v + i] = rbf(rbf(a[i]) + z[4 * v + i]);
        }
    }
}
template <int NSPLIT>
__device__ __forceinline__ void norm_fold8(const float* __restrict__ sp, size_t slab_stride, float z[8], const float* __restrict__ pair_scale = nullptr) {
    f32x4 p[NSPLIT > 0 ? NSPLIT : 1][2];
    norm_load8<NSPLIT>(sp, slab_stride, p);
    norm_sum8<NSPLIT>(p, z, pair_scale);
}
__device__ __forceinline__ float norm_ssq8(const float z[8]) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss = fmaf(z[i], z[i], ss);
    return ss;
}
__device__ __forceinline__ float norm_rstd(float sum_sq, float eps) { return 1.0f / sqrtf(sum_sq / 1024.0f + eps); }
__device__ __forceinline__ u32x4 norm_scale8(const float z[8], float sc, const u32x4 wv) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        o[i] = pack_bf2(rbf(z[2 * i] * sc) * bf_lo(wv[i]), rbf(z[2 * i + 1] * sc) * bf_hi(wv[i]));
    return o;
}
__device__ __forceinline__ void unpack8(const u32x4 v, float z[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { z[2 * i] = bf_lo(v[i]); z[2 * i + 1] = bf_hi(v[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float z[8]) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf2(z[2 * i], z[2 * i + 1]);
    return o;
}

// ---- decode attention: merge of the eight per-wave softmax states of one (row, head, 8-dim column group) ------------
// sp[w * wave_stride + 0..9] = m, l, o[0..7] of wave w.  Waves are folded in order 0..7; returns the 8 normalised outputs
// as bf16.  Shared by the attention kernel (states in LDS) and by the o_proj kernel that follows the few-row split form
// (states in memory): the same operations in the same order, so the same bits.
__device__ __forceinline__ u32x4 att_merge_waves(const float* sp, int wave_stride) {
    float M = sp[0], L = sp[1], O[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) O[j] = sp[2 + j];
    for (int w2 = 1; w2 < 8; ++w2) {
        const float* p = sp + (size_t)w2 * wave_stride;
        const float m2 = p[0], l2 = p[1];
        const float mn = fmaxf(M, m2);
        const float a1 = __expf(M - mn), a2 = __expf(m2 - mn);
        L = fmaf(L, a1, l2 * a2);                     // explicit: the compiler's own contraction may differ between kernels
#pragma unroll
        for (int j = 0; j < 8; ++j) O[j] = fmaf(O[j], a1, p[2 + j] * a2);
        M = mn;
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    u32x4 ov;
#pragma unroll
    for (int k = 0; k < 4; ++k) ov[k] = pack_bf2(O[2 * k] * inv, O[2 * k + 1] * inv);
    return ov;
}

// ---- error plumbing (host) -------------------------------------------------------
void t3_set_error(const std::string& msg);
#define T3_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            t3_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define T3_REQUIRE(cond, msg)                                                                \
    do {                                                                                     \
        if (!(cond)) { t3_set_error(std::string(msg) + " [" #cond "]"); return 1; }          \
    } while (0)

// ---- KV cache geometry -------------------------------------------------------------
// pool[layer][page][kv(0=K,1=V)][head][token(64)][head_dim]   bf16
// layer_stride = n_pages * 2 * H * 64 * hd elements.
// Every kernel of the library is launched through T3_LAUNCH.  When the engine's kernel-timing mode has
// armed a pair of events, the launch binds them to the dispatch itself, so that their elapsed time is
// the kernel's own start-to-end time on the device (what a profiler reports), not a stream bracket.
struct T3LaunchEvents { hipEvent_t start, stop; };
inline thread_local T3LaunchEvents t3_launch_events = {nullptr, nullptr};
#define T3_LAUNCH(kernel, grid, block, lds, st, ...)                                                        \
    do {                                                                                                    \
        if (t3_launch_events.start) {                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, t3_launch_events.start,                     \
                                  t3_launch_events.stop, 0, __VA_ARGS__);                                   \
            t3_launch_events = T3LaunchEvents{nullptr, nullptr};                                            \
        } else {                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                  \
        }                                                                                                   \
    } while (0)

struct KvView {
    bf16_t* pool;
    int64_t layer_stride;        // elements
    const int32_t* block_table;  // [n_streams][max_pages]
    int32_t max_pages;
    int32_t H, hd;
};

// ---- launchers (all asynchronous on `st`) -------------------------------------------
enum GemmEpilogue { EPI_STORE = 0, EPI_RESID = 1, EPI_SILU = 2, EPI_QKV_ROPE = 3, EPI_SLAB = 4, EPI_STORE_GATHER = 5 };
#define GEMM_SLAB_SPLIT 8          // K is cut into 8 block-level ranges in the split-K (slab) form
#define ATTN_GROUP_ROWS 4          // rows one workgroup of the grouped decode attention takes (attn.hip)
#define ATTN_OPROJ_SPLIT 16        // partial slabs of o_proj when it runs inside the decode attention launch: one per head
#define GEMM_FUSED_NORM_ROWS 4     // launches of at most this many rows can take the residual-add + RMSNorm into the projection

struct GemmArgs {
    const bf16_t* A; int lda;          // activations [M, K]
    // Activation tile order (decode buffers): tile (row block of 16, k block of 32) = 1 KiB at ((r/16)*(K/32) + k/32)*512
    // elements, inside it element ((k%32)/8*16 + r%16)*8 + k%8: one 16-byte load per lane is an MFMA operand and a wave
    // reads 1 KiB of contiguous memory.  Rows are padded to a multiple of 16 (padding is never stored from).
    bool a_tiled;                      // A is in activation tile order (lda unused)
    bool c_tiled;                      // EPI_SILU: h is written in activation tile order
    const int32_t* a_rows;             // optional gather of A rows (nullptr = identity)
    const bf16_t* Wt;                  // weights in tile order (bf16 tiles, or int8 pair tiles when wscale is set)
    const float* wscale;               // int8 weights: one scale per output column, nullptr = bf16 weights
    bf16_t* C; int ldc;                // output
    int M, N, K;
    int n_valid;                       // columns >= n_valid are not stored (EPI_STORE)
    // EPI_QKV_ROPE
    const bf16_t* rope; const int32_t* pos; const int32_t* row_stream;
    KvView kv; int layer;
    // EPI_SLAB
    float* slab;                       // [GEMM_SLAB_SPLIT][M][N] fp32 partial sums
    // Norm in the prologue (EPI_SILU / EPI_QKV_ROPE, K = 1024, M <= GEMM_FUSED_NORM_ROWS): when fn_w is set, A is unused and
    // the kernel itself computes res' = bf16(bf16(sum of fn_nsplit slabs) + fn_res_in) (fn_nsplit = 0: res' = fn_res_in)
    // and multiplies with rmsnorm(res') * fn_w, bit for bit what launch_slab_add_rmsnorm would have handed it; workgroup 0
    // stores res' to fn_res_out (a buffer other than fn_res_in: every workgroup reads fn_res_in)
    const float* fn_slab; int fn_nsplit; const bf16_t* fn_res_in; bf16_t* fn_res_out; const bf16_t* fn_w; float fn_eps;
    const float* fn_pair_scale;        // fn_nsplit == ATTN_OPROJ_SPLIT with int8 o_proj weights: its [1024] column scales, else nullptr
    // EPI_SLAB, K = 1024, M <= GEMM_FUSED_NORM_ROWS (o_proj of a few-row decode step): when a_states is set, A is unused and
    // the rows are the attention outputs merged here from the per-wave softmax states that launch_attn_decode_split left
    // (attn.hip), with the arithmetic of the attention kernel's own merge: the same bits as its stored output
    const float* a_states;
#ifdef T3_GEMM_TIMELINE
    long long* ts;                     // tools/gemm_timeline.hip: [block][wave][8] wall-clock stamps
#endif
};
// Stage stamps for the stand-alone timeline tool; nothing in the shipped library.
#ifdef T3_GEMM_TIMELINE
#define T3_STAMP(i)                                                                                       \
    do {                                                                                                  \
        if (lane == 0)                                                                                    \
            a.ts[(((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 128 + w * 8 + (i)] = \
                wall_clock64();                                                                           \
    } while (0)
#else
#define T3_STAMP(i) do { } while (0)
#endif
int launch_gemm(int epi, const GemmArgs& a, hipStream_t st);
int launch_gemm_slab(const GemmArgs& a, hipStream_t st);
// Prompt rows (prefill): LDS-tiled 128 x 128 kernel, epi = EPI_QKV_ROPE, EPI_SILU or EPI_SLAB (split-K 8).
int launch_gemm_prompt(int epi, const GemmArgs& a, hipStream_t st);
// res = bf16(bf16(sum_s slab[s]) + res) when slab != nullptr (nsplit slabs of [rows, d] fp32);
// xn = rmsnorm(res) * w.  One pass, the row stays in registers.
int launch_slab_add_rmsnorm(const float* slab, int nsplit, bf16_t* res, const bf16_t* w, bf16_t* xn, int rows, int d,
                            float eps, hipStream_t st, bool xn_tiled = false, const bf16_t* emb = nullptr,
                            const int32_t* emb_slot = nullptr, const int32_t* last_token = nullptr,
                            const float* pair_scale = nullptr);
int launch_repack(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t st);
// W[N,K] bf16 -> int8 pair tiles (offset binary) + scale[N] = absmax(row) / 127 (1 for an all-zero row)
int launch_quantize(const bf16_t* w, uint8_t* wq, float* scale, int N, int K, hipStream_t st);
int launch_add_rmsnorm(const bf16_t* x, bf16_t* res, const bf16_t* w, bf16_t* xn, int rows, int d,
                       float eps, hipStream_t st);
// row_pfx (optional): per row, the page that holds positions [0, npfx) of its stream (shared prefix), or -1
int launch_attn_decode(const bf16_t* qkv, int ldq, bf16_t* out, int rows, const int32_t* row_stream,
                       const int32_t* row_ctx, const KvView& kv, int layer, hipStream_t st,
                       const int32_t* row_pfx = nullptr, int npfx = 0);
// Few-row steps: one single-wave workgroup per (head, row, wave of the kernel above); each leaves its softmax state
// (m, l, o[8] per 8-dim column group: ATT_STATE_FLOATS floats) in states[row][head][wave][column group] instead of merging in LDS,
// so that one row's K/V stream is pulled by up to eight CUs instead of one.  The projection that follows merges them.
#define ATT_STATE_FLOATS 12
int launch_attn_decode_split(const bf16_t* qkv, int ldq, float* states, int rows, const int32_t* row_stream,
                             const int32_t* row_ctx, const KvView& kv, int layer, hipStream_t st,
                             const int32_t* row_pfx = nullptr, int npfx = 0);
// the same with o_proj in the launch: slab[head][row][0..1024) fp32, to be folded by launch_slab_add_rmsnorm(nsplit = ATTN_OPROJ_SPLIT)
int launch_attn_decode_oproj(const bf16_t* qkv, int ldq, float* slab, int rows, const int32_t* row_stream,
                             const int32_t* row_ctx, const KvView& kv, int layer, const bf16_t* wo, bool int8,
                             int rows_per_group, hipStream_t st, const int32_t* row_pfx = nullptr, int npfx = 0);
// tiles_dev: (sequence index, first query row) per 64-row query tile.  seq_pfx_dev (optional): per sequence, the
// pool page holding keys [0, npfx) when the sequence's rows start at position npfx, or -1 (rows start at 0)
int launch_attn_prefill(const bf16_t* qkv, int ldq, bf16_t* out, const int32_t* seq_start_dev,
                        const int32_t* tiles_dev, int ntiles, int H, int hd, hipStream_t st,
                        const int32_t* seq_pfx_dev = nullptr, int npfx = 0, const KvView* kv = nullptr, int layer = 0,
                        const int32_t* seq_first_dev = nullptr, const int32_t* seq_stream_dev = nullptr);

struct EmbedArgs {
    bf16_t* out; int d;                 // [rows, d]
    int rows;
    const int32_t* kind;                // per row: 0 cond row, 1 text, 2 zero, 3 SOS, 4 decode token
    const int32_t* idx;                 // cond row / text id / - / - / slot
    const int32_t* aux;                 // cond buffer index / text position / - / - / -
    const float* cond;                  // [n_cond_bufs, 34, d] fp32
    const bf16_t* text_emb; const bf16_t* text_pos; const bf16_t* speech_emb; const bf16_t* speech_pos;
    const int32_t* last_token;          // [slots] un-offset speech id per slot
    int sos_id;
};
int launch_embed(const EmbedArgs& a, hipStream_t st);

struct SampleArgs {
    const bf16_t* logits; int ldl; int vocab;      // speech logits rows
    const int32_t* row_c; const int32_t* row_u;    // per request rows (row_u < 0: single stream)
    const int32_t* slot;                           // per request slot (seen / last_token index), nullptr = i
    int n; float cfg_scale;
    const float* rep_penalty;                      // per request
    const float* temperature; const float* top_p; const float* min_p;   // per request (nullptr = greedy)
    const uint64_t* seed; const int32_t* step;     // per request RNG key / counter
    uint8_t* seen; int seen_stride;                // [slots, seen_stride] column space (offset 1000)
    int32_t* last_token;                           // [slots] written with the un-offset id (may be null)
    int32_t* out_tokens;                           // [n] +1000 space
    float* out_logits;                             // optional [slots, vocab+1000] penalised fp32 logits
};
int launch_sample(const SampleArgs& a, hipStream_t st);
int launch_seen_init(uint8_t* seen_row, int seen_stride, const int32_t* ids_dev, int n_ids, hipStream_t st);
