/*
 * t3_hip.h -- C ABI of the MI355X (gfx950) T3 speech-token decode engine.
 *
 * Everything the Python host layer (or any other FFI: cgo, JNI, ctypes) needs is
 * declared here: plain pointers and sizes, int status codes, no C++ or torch types.
 * Status 0 = success; on failure t3_last_error() returns a message for the calling
 * thread.  Pointers named *_dev are device (HIP) addresses, everything else is host
 * memory.  All device work of a call is enqueued on the engine's own HIP stream, or on the caller's when the engine
 * was made by t3_engine_create_on_stream (per-op entry points: on their `stream` argument; 0 = the default stream).
 *
 * The reference has no native code of its own: the arithmetic of this path runs
 * inside the third-party vLLM wheel.  Each entry point therefore cites the Python
 * interface of the reference (paths relative to the reference tree) whose job it
 * takes over.
 */
#ifndef T3_HIP_H
#define T3_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T3_ABI_VERSION 4

typedef struct t3_engine t3_engine;

/* Engine construction parameters.
 * Replaces the keyword arguments of vllm.LLM(...) built at
 * src/chatterbox_vllm/tts.py:130-140 plus the model constants of
 * t3-model/config.json:8-32 and models/t3/modules/t3_config.py:2-21. */
typedef struct t3_config {
    int32_t n_layers, d_model, n_heads, head_dim, d_ff;
    int32_t text_vocab, speech_vocab, text_pos_rows, speech_pos_rows;
    int32_t max_model_len;     /* prompt + generated positions per request            */
    int32_t max_num_seqs;      /* requests decoded concurrently                       */
    int32_t cfg_paired;        /* 0: one stream per request, no CFG.  1: a cond and an uncond stream, each with its
                                  own KV (what models/t3/t3.py:742-778 and the reference's README describe).
                                  2: cond + uncond rows, but the uncond row gets NO attention output and no KV: what
                                  the reference computes if the serving stack attends only the rows its scheduler
                                  counted (the model doubles the rows behind the scheduler's back, t3.py:762-770) */
    int32_t max_prefill_rows;  /* prompt rows (per stream) admitted in one step       */
    int32_t device;            /* HIP device ordinal                                  */
    int32_t use_graphs;        /* 1: replay decode steps from captured HIP graphs     */
    int32_t weight_int8;       /* 1: backbone projections held as int8 + one fp32 scale per output
                                  column (W8A16, quantised at t3_engine_finalize); head stays bf16 */
    float cfg_scale;           /* CHATTERBOX_CFG_SCALE, models/t3/t3.py:343           */
    float rms_eps;
    int64_t kv_pages;          /* KV pool size in 64-token pages; 0 = no oversubscription */
    int32_t prefix_sharing;    /* 1: the K/V of positions 0..33 (the conditioning rows, t3.py:229-240: the same in
                                  the cond and uncond stream and in every request of one voice) are computed once
                                  per distinct conditioning tensor and kept in a page of their own; ids and logits
                                  are bit-identical to 0.  (The reference defeats vLLM's prefix cache with a random
                                  hash, t3.py:286-289.)                                                          */
    int32_t max_voices;        /* distinct conditioning tensors kept; raised to max_num_seqs + 1 if smaller       */
} t3_config;

/* Per-request sampling parameters: the vllm.SamplingParams fields set at
 * src/chatterbox_vllm/tts.py:410-417. */
typedef struct t3_sampling {
    float temperature, top_p, min_p, repetition_penalty;
    int32_t max_tokens;
    int32_t stop_token;        /* in the engine's +1000 id space (7562); -1 = none    */
    int32_t has_seed;
    int32_t n_extra_stop;      /* further stop ids in extra_stop_tokens (0..3): vllm.SamplingParams.stop_token_ids is a list */
    uint64_t seed;
    int32_t extra_stop_tokens[4];
} t3_sampling;

/* Counters of the last t3_engine_step (for benchmarks / tracing). */
typedef struct t3_step_stats {
    int32_t prefill_requests, prefill_rows, decode_requests, decode_rows;
    int64_t decode_ctx_tokens; /* sum over decode streams of context length           */
    int32_t free_pages, waiting, running;
    int32_t prefill_outstanding; /* requests of the prefill job that runs beside the decode steps, 0 = none */
    float decode_gpu_ms;       /* HIP-event time of the decode part, 0 if not timed   */
    float prefill_gpu_ms;
} t3_step_stats;

const char* t3_last_error(void);
int t3_abi_version(void);

/* ---- engine life cycle: stands in for vllm.LLM.__init__ (tts.py:140) and
 *      T3VllmModel.__init__/load_weights (models/t3/t3.py:305-379) ---------------- */
int t3_engine_create(const t3_config* cfg, t3_engine** out);
/* The same engine with the caller's HIP stream (a hipStream_t; 0 = the default stream) as its main stream: decode
 * iterations, uploads and the iteration events are enqueued there, in order with the caller's own work on that stream.
 * The stream stays the caller's (it must outlive the engine; t3_engine_destroy synchronises it and leaves it alone).
 * Prompts still run beside the decode iterations on a second stream of the engine's own, which starts behind what the
 * main stream holds. */
int t3_engine_create_on_stream(const t3_config* cfg, void* stream, t3_engine** out);
/* `name` is the checkpoint tensor name routed by t3.py:347-369 (tfmr.*, text_emb.weight,
 * speech_emb.weight, text_pos_emb.emb.weight, speech_pos_emb.emb.weight,
 * speech_head.weight) or "rope.cos_sin" ([max_pos, head_dim] bf16, cos|sin).
 * `data` is host memory holding bf16 (dtype 0) in row-major `shape`. */
int t3_engine_load_tensor(t3_engine* e, const char* name, const void* data,
                          const int64_t* shape, int32_t ndim, int32_t dtype);
/* The same from DEVICE memory on the engine's device (e.g. the storage of a PyTorch-ROCm tensor): borrowed for the
 * duration of the call, which copies it and returns when the copy is complete. */
int t3_engine_load_tensor_dev(t3_engine* e, const char* name, const void* data_dev,
                              const int64_t* shape, int32_t ndim, int32_t dtype);
/* Repack the loaded matrices into the streaming tile order; must follow the last load. */
int t3_engine_finalize(t3_engine* e);
int t3_engine_destroy(t3_engine* e);

/* ---- request flow: stands in for LLM.generate(request_items, SamplingParams)
 *      (tts.py:365-370, 408-418) and the per-request prompt layout of
 *      T3MultiModalProcessor.apply (t3.py:229-262) --------------------------------- */
/* text_ids: tokenizer output for "[START]...[STOP]"; cond: fp32 [34, d_model] (host). */
int t3_engine_add_request(t3_engine* e, int64_t req_id, const int32_t* text_ids, int32_t n_text,
                          const float* cond, const t3_sampling* sp);
/* One scheduler iteration: admit + prefill new requests, decode one token for every
 * running request.  out_* have room for `cap` entries; *n_out receives the number of
 * (req_id, token, finished) triples reported.  finished: 0 running, 1 stop token,
 * 2 length, 3 error: the request's logits had no finite maximum (NaN weights or conditioning); its token is -1
 * and the request has ended.  Tokens are in the +1000 id space, the stop token is reported.
 * Iterations are pipelined one deep: a call launches the next iteration and reports the
 * previous one, so a call may report nothing (the first after admission) and, when a
 * measuring mode has just been switched on, two iterations' triples in order.  The prompts of
 * newly admitted requests run beside the decode iterations as a job on a second stream; the
 * call that finds the job complete reports the first token of each of its requests (before any
 * later token of theirs), and they decode from the next iteration on.  Give
 * cap >= 2 * max_num_seqs and loop while t3_engine_has_unfinished(). */
int t3_engine_step(t3_engine* e, int64_t* out_req_ids, int32_t* out_tokens, int32_t* out_finished,
                   int32_t cap, int32_t* n_out);
/* Admission order of waiting requests: 0 = first come first served (vLLM's default policy, what the reference
 * runs under, tts.py:130-140); 2 = longest prompt first, which shortens the half-empty tail of a queue that is known
 * up front (one generate() call) using only what every caller knows; 1 = longest first by prompt + max_tokens, the same
 * idea for a caller that sets max_tokens per request to the length it expects.  Applies to requests added afterwards;
 * a request's tokens do not depend on what it shares a step with. */
int t3_engine_set_schedule_policy(t3_engine* e, int32_t policy);
/* Token budget of a prefill job, the `max_num_batched_tokens` of the vLLM scheduler the reference runs under
 * (README.md:145 shows 8192; vLLM cuts the prompt that does not fit the rest of a step's budget and continues it in the
 * next step): positions = 0, the default, sends every prompt through the layers whole (up to max_prefill_rows positions
 * of whole prompts per job); positions > 0 lets a job carry at most that many positions per stream, prompts already
 * under way first, and cuts the prompt that does not fit; its later chunks read the earlier ones' keys from the KV pool.
 * A request's tokens do not depend on where its prompt is cut (the same key tiles, the same values: bit-identical).
 * Only while no request is admitted. */
int t3_engine_set_prefill_chunk(t3_engine* e, int32_t positions);
int t3_engine_abort(t3_engine* e, int64_t req_id);
int t3_engine_has_unfinished(t3_engine* e);
int t3_engine_get_stats(t3_engine* e, t3_step_stats* out);
/* Enable HIP-event timing of the prefill / decode parts of each step (adds a sync). */
int t3_engine_set_timing(t3_engine* e, int32_t on);

/* ---- measurement: per-launch HIP-event timing by kernel class.  While on, decode steps are
 *      launched eagerly (no graph replay) and every kernel launch carries a start and a stop
 *      event bound to its own dispatch (hipExtLaunchKernelGGL) on the engine's stream, so a
 *      launch's time is the kernel's start-to-end time on the device; totals accumulate until
 *      the next t3_engine_set_kernel_timing(e, 1).
 *      No counterpart in the reference (its only timing is wall clock around LLM.generate,
 *      src/chatterbox_vllm/tts.py:407-420). ------------------------------------------------- */
int t3_engine_set_kernel_timing(t3_engine* e, int32_t on);
int t3_kernel_class_count(void);
const char* t3_kernel_class_name(int32_t cls);
/* phase 0 = decode steps, 1 = prefill.  Arrays of n >= t3_kernel_class_count() entries:
 * total milliseconds, launches, and algorithmic bytes (weights or KV a launch must read). */
int t3_engine_get_kernel_times(t3_engine* e, int32_t phase, double* ms, int64_t* launches, int64_t* bytes, int32_t n);

/* ---- conditioning encoder: the reference's T3CondEnc + Perceiver
 *      (src/chatterbox_vllm/models/t3/modules/cond_enc.py:57-123, perceiver.py:116-215) as fp32 HIP kernels.
 *      Called once per voice from get_audio_conditionals (tts.py:215-247) and for the emotion row of
 *      update_exaggeration (tts.py:249-257).  All pointers are host memory, fp32. ----------------------- */
typedef struct t3_condenc t3_condenc;
int t3_condenc_create(int32_t device, t3_condenc** out);
/* `name` is the checkpoint tensor name without the "cond_enc." prefix (spkr_enc.weight, spkr_enc.bias,
 * emotion_adv_fc.weight, perceiver.pre_attention_query, perceiver.attn.{norm,to_q,to_k,to_v,proj_out}.{weight,bias}). */
int t3_condenc_load_tensor(t3_condenc* c, const char* name, const float* data, const int64_t* shape, int32_t ndim);
/* speaker_emb [256], prompt_emb [n_prompt, 1024] (speech_emb[tokens] + speech_pos_emb[:n], tts.py:238),
 * out [34, 1024]: row 0 speaker, rows 1..32 perceiver, row 33 emotion_adv_fc(exaggeration).  Blocking. */
int t3_condenc_encode(t3_condenc* c, const float* speaker_emb, const float* prompt_emb, int32_t n_prompt,
                      float exaggeration, float* out);
int t3_condenc_emotion_row(t3_condenc* c, float exaggeration, float* out);
int t3_condenc_destroy(t3_condenc* c);

/* ---- debugging taps for parity tests ------------------------------------------- */
/* Keep the CFG-combined, penalised fp32 logits of every sampled token (turns graph replay
 * off).  The logits are what models/t3/t3.py:725-739 returns, after the sampler's
 * repetition penalty. */
int t3_engine_set_keep_logits(t3_engine* e, int32_t on);
/* Copies the logits [speech_vocab + 1000] that produced the most recent token of the
 * still-running request `req_id` into host memory `out`. */
int t3_engine_last_logits(t3_engine* e, int64_t req_id, float* out);

/* ---- per-op entry points (parity tests call the kernels one by one).
 *      All tensors are device memory, bf16 unless noted, row-major. ---------------- */

/* Repack W[N,K] (bf16, N%16==0, K%32==0) into the streaming tile order used by t3_op_gemm. */
int t3_op_repack_weight(const void* w_dev, void* wt_dev, int32_t N, int32_t K, void* stream);
/* res = bf16(x + res) when x_dev != NULL (else res is used as is); xn = rmsnorm(res) * w.
 * Equivalent of vLLM's fused_add_rms_norm reached from LlamaDecoderLayer (t3.py:768-773). */
int t3_op_add_rmsnorm(const void* x_dev, void* res_dev, const void* w_dev, void* xn_dev,
                      int32_t rows, int32_t d, float eps, void* stream);
/* out[M,N] = bf16(A[M,K] @ W[N,K]^T), W given in tile order.  epilogue 0: store;
 * 1: out = bf16(bf16(acc) + out) (residual add in place). */
int t3_op_gemm(const void* a_dev, const void* wt_dev, void* out_dev, int32_t M, int32_t N, int32_t K,
               int32_t ldc, int32_t epilogue, void* stream);
/* Split-K form used by the decode path for o_proj / down_proj (N = 1024): K is cut into
 * t3_op_slab_split() ranges, slab[s][M][N] (fp32) receives the partial sums of range s. */
int t3_op_gemm_slab(const void* a_dev, const void* wt_dev, float* slab_dev, int32_t M, int32_t N, int32_t K, void* stream);
int t3_op_slab_split(void);
/* Measurement / test switch, process-wide, no counterpart in the reference: which kernel prompt-row projections use from
 * now on.  0 (default): one workgroup per 128-row tile; 1: one workgroup walks all rows of its weight tile, so every weight
 * byte is fetched once per launch (slower below a few thousand rows).  Same bits either way. */
int t3_op_set_prompt_gemm_form(int32_t walk);
/* res = bf16(bf16(slab[0] + ... + slab[nsplit-1]) + res) (skipped when nsplit == 0), then
 * xn = rmsnorm(res) * w; d must be 1024; nsplit 0 or t3_op_slab_split().  The decode-path partner of t3_op_gemm_slab. */
int t3_op_slab_add_rmsnorm(const float* slab_dev, int32_t nsplit, void* res_dev, const void* w_dev, void* xn_dev,
                           int32_t rows, int32_t d, float eps, void* stream);
/* Same with xn written in activation tile order (t3_gemm_desc), what the decode step's projections read. */
int t3_op_slab_add_rmsnorm_tiled(const float* slab_dev, int32_t nsplit, void* res_dev, const void* w_dev, void* xn_dev,
                                 int32_t rows, int32_t d, float eps, void* stream);
/* h[M,F] = bf16(bf16(silu(g)) * u) with [g|u] = bf16(A[M,K] @ Wgu[2F,K]^T)  (LlamaMLP). */
int t3_op_gemm_silu(const void* a_dev, const void* wt_dev, void* h_dev, int32_t M, int32_t F, int32_t K,
                    void* stream);
/* qkv[M,3*H*hd] = A @ Wqkv^T, q and k rotated (half-split RoPE, table rope_dev [max_pos, hd]
 * cos|sin) at positions pos_dev[M]; k and v rows are also written into the paged KV cache of
 * layer `layer` through stream_dev[M] (KV stream of each row, <0 = skip) and the block table. */
int t3_op_gemm_qkv_rope(const void* a_dev, const void* wt_dev, void* qkv_dev, int32_t M, int32_t H, int32_t hd,
                        int32_t K, const void* rope_dev, const int32_t* pos_dev, const int32_t* stream_dev,
                        void* kv_pool_dev, int64_t layer_stride_elems, int32_t layer,
                        const int32_t* block_table_dev, int32_t max_pages, void* stream);
/* Prompt-row forms of the three fused projections above (same arguments, same arithmetic contract):
 * an LDS-tiled 128 x 128 kernel for the few hundred rows of a prefill, where the weight-streaming
 * kernels would re-read every activation row once per 32 output columns.  The engine sends prompt
 * rows here and decode rows to the forms above, always, so a row's bits do not depend on the launch. */
int t3_op_gemm_prompt_slab(const void* a_dev, const void* wt_dev, float* slab_dev, int32_t M, int32_t N, int32_t K,
                           void* stream);
int t3_op_gemm_prompt_silu(const void* a_dev, const void* wt_dev, void* h_dev, int32_t M, int32_t F, int32_t K,
                           void* stream);
int t3_op_gemm_prompt_qkv_rope(const void* a_dev, const void* wt_dev, void* qkv_dev, int32_t M, int32_t H, int32_t hd,
                               int32_t K, const void* rope_dev, const int32_t* pos_dev, const int32_t* stream_dev,
                               void* kv_pool_dev, int64_t layer_stride_elems, int32_t layer,
                               const int32_t* block_table_dev, int32_t max_pages, void* stream);
/* int8 weights (the engine's weight_int8 mode): W[N,K] bf16 -> wq_dev (N*K bytes, offset-binary int8 in the
 * pair-tile order the kernels stream) and scale_dev[N] = absmax(row)/127; q = rint(w/scale), ties to even.
 * N % 16 == 0, K % 64 == 0.  The reference has no quantised path (weights are bf16 under vLLM, t3.py:305-379). */
int t3_op_quantize_weight(const void* w_dev, void* wq_dev, float* scale_dev, int32_t N, int32_t K, void* stream);
/* Every projection form behind one descriptor: what the entry points above do, plus int8 weights. */
typedef struct t3_gemm_desc {
    const void* a_dev;            /* activations [M, K] bf16, row-major                               */
    const void* w_dev;            /* bf16 tile order, or int8 pair tiles when w_scale_dev != NULL     */
    const float* w_scale_dev;     /* [N] or NULL                                                      */
    void* out_dev;                /* epilogue 0/1: [M, ldc]; 2: h [M, N/2]; 3: qkv [M, N]; 4: unused  */
    float* slab_dev;              /* epilogue 4: [t3_op_slab_split()][M][N] fp32                      */
    int32_t M, N, K, ldc;
    int32_t epilogue;             /* 0 store, 1 residual add, 2 SiLU-mul, 3 q|k|v + RoPE + cache write, 4 split-K slabs */
    int32_t prompt_form;          /* 0: weight-streaming kernel; 1: LDS-tiled prompt kernel (epilogues 2, 3, 4) */
    int32_t H, hd, layer, max_pages;                           /* epilogue 3, as t3_op_gemm_qkv_rope   */
    const void* rope_dev; const int32_t* pos_dev; const int32_t* stream_dev; void* kv_pool_dev;
    int64_t layer_stride_elems; const int32_t* block_table_dev;
    /* Activation tile order, the layout of the decode step's intermediate buffers: rows padded to a multiple
     * of 16; element (r, k) at ((r/16)*(K/32) + k/32)*512 + ((k%32)/8*16 + r%16)*8 + k%8.                    */
    int32_t a_tiled;              /* a_dev is in activation tile order (streaming form only)                  */
    int32_t out_tiled;            /* epilogue 2: h is written in activation tile order (streaming form only)  */
    /* Residual-add + RMSNorm in the prologue (streaming form, epilogues 2 and 3, K = 1024, M <= 4): when fn_norm_w_dev
     * is set, a_dev is ignored and the projection reads rmsnorm(res') * fn_norm_w with
     * res' = bf16(bf16(fn_slab[0] + ... + fn_slab[fn_nsplit-1]) + fn_res_in) (fn_nsplit 0: res' = fn_res_in), the bits
     * t3_op_slab_add_rmsnorm would have produced; res' is stored to fn_res_out_dev (not the buffer of fn_res_in_dev).
     * What the engine runs for decode steps of at most 4 rows.                                                    */
    const float* fn_slab_dev;     /* [fn_nsplit][M][1024] fp32, or NULL                                          */
    const void* fn_res_in_dev;    /* [M, 1024] bf16                                                              */
    void* fn_res_out_dev;         /* [M, 1024] bf16 (fn_nsplit > 0)                                              */
    const void* fn_norm_w_dev;    /* [1024] bf16; NULL = no norm prologue                                        */
    int32_t fn_nsplit;            /* 0, t3_op_slab_split(), or t3_op_oproj_split() (head partials, folded pairwise)  */
    float fn_eps;
    const float* fn_pair_scale_dev; /* fn_nsplit 16 with int8 o_proj weights: their [1024] column scales; else NULL  */
} t3_gemm_desc;
int t3_op_gemm_ex(const t3_gemm_desc* desc, void* stream);
/* Single-query attention over the paged KV cache: row r attends to positions
 * [0, ctx_dev[r]) of KV stream stream_dev[r]; q is read from qkv_dev[r, 0:H*hd]. */
int t3_op_attn_decode(const void* qkv_dev, void* out_dev, int32_t rows, int32_t H, int32_t hd,
                      const int32_t* stream_dev, const int32_t* ctx_dev, const void* kv_pool_dev,
                      int64_t layer_stride_elems, int32_t layer, const int32_t* block_table_dev,
                      int32_t max_pages, void* stream);
/* Attention of (row, head) and, in the same launch, that head's share of o_proj (self_attn.o_proj of the vLLM
 * LlamaAttention reached from t3.py:768-773): what full decode steps run.  slab_dev [t3_op_oproj_split() = 16 heads]
 * [rows][H*hd] fp32 receives bf16(attention output of head h) x Wq[:, h*hd .. (h+1)*hd)^T with Wq = Wo (bf16 tiles of
 * t3_op_repack_weight), or, with wo_int8 set, the UNSCALED int8 values of the pair tiles of t3_op_quantize_weight (the
 * fold applies the scales).  rows_per_group: 1 = one workgroup per (head, row); 4 = one workgroup per (head, 4 consecutive
 * rows), which reads the head's weights once per four rows; both give the same bits.  The fold
 * (t3_op_head_slab_add_rmsnorm, or the norm prologue of t3_op_gemm_ex with fn_nsplit 16) adds heads 2s and 2s+1, scales, and
 * sums the eight pairs in order: the additions t3_op_gemm_slab + t3_op_slab_add_rmsnorm perform, so the bits are the same.
 * prefix_page_dev may be NULL (no shared prefix). */
int t3_op_attn_decode_oproj(const void* qkv_dev, float* slab_dev, int32_t rows, int32_t H, int32_t hd,
                            const int32_t* stream_dev, const int32_t* ctx_dev, const void* kv_pool_dev,
                            int64_t layer_stride_elems, int32_t layer, const int32_t* block_table_dev,
                            int32_t max_pages, const void* wo_dev, int32_t wo_int8, int32_t rows_per_group,
                            const int32_t* prefix_page_dev, int32_t n_prefix, void* stream);
int t3_op_oproj_split(void);

/* Decode attention for steps of 1..4 rows (a single stream, the tail of a queue) followed by o_proj.  One single-wave
 * workgroup per (head, row, eighth of the context) leaves its online-softmax state in states_dev
 * (rows x t3_op_attn_states_floats(H) fp32); the o_proj kernel merges the eight states of a (row, head) itself, with the
 * arithmetic of t3_op_attn_decode's merge, and writes the eight split-K slabs of t3_op_gemm_slab bit for bit
 * (slab_dev [8][rows][H*hd] fp32).  wo_scale_dev NULL for bf16 weights.  Same reference lines as t3_op_attn_decode:
 * vLLM's paged decode attention behind src/chatterbox_vllm/models/t3/t3.py:282-300. */
int t3_op_attn_states_floats(int32_t H);
int t3_op_attn_decode_split_oproj(const void* qkv_dev, float* states_dev, float* slab_dev, int32_t rows, int32_t H,
                                  int32_t hd, const int32_t* stream_dev, const int32_t* ctx_dev, const void* kv_pool_dev,
                                  int64_t layer_stride_elems, int32_t layer, const int32_t* block_table_dev,
                                  int32_t max_pages, const void* wo_dev, const float* wo_scale_dev,
                                  const int32_t* prefix_page_dev, int32_t n_prefix, void* stream);
/* res = bf16(bf16(sum over s of scale * (slab[2s] + slab[2s+1])) + res), xn = rmsnorm(res) * w, for the 16 head partials
 * of t3_op_attn_decode_oproj; wo_scale_dev [d] fp32 or NULL (bf16 weights). */
int t3_op_head_slab_add_rmsnorm(const float* slab_dev, const float* wo_scale_dev, void* res_dev, const void* w_dev,
                                void* xn_dev, int32_t rows, int32_t d, float eps, void* stream);
/* Causal attention inside packed prompt rows: sequence i owns rows [seq_start[i], seq_start[i+1]). */
int t3_op_attn_prefill(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                       const int32_t* seq_start_host, int32_t H, int32_t hd, void* stream);
/* t3_op_attn_prefill for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  It has no counterpart in the reference (whose attention runs inside the vLLM wheel).
 * Same arguments and bit-identical output; the list of query tiles, which t3_op_attn_prefill stages through a buffer
 * it allocates, is built on the device into a 32 KiB array inside the library's code object, so the call allocates
 * nothing, copies nothing and does not synchronise: both kernels are only enqueued on `stream`.
 * Limits: at most 1024 sequences and 4096 query tiles (sum over sequences of ceil(rows / 64)) per call; larger calls
 * are refused with an error.  Every GPU holds one such array for the life of the process, every call on that GPU uses
 * it: calls that overlap on one GPU race.  Make the first call on a device outside any graph capture (it looks the
 * array's address up once per device). */
int t3_op_attn_prefill_static(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                              const int32_t* seq_start_host, int32_t H, int32_t hd, void* stream);
/* The two attention forms with a shared prefix (the engine's prefix_sharing mode; no counterpart in the reference,
 * which keeps vLLM's prefix cache from matching, models/t3/t3.py:286-289).
 * Decode: where prefix_page_dev[r] >= 0, positions [0, n_prefix) of row r are read from that page of the pool
 * instead of page 0 of the row's own stream (1 <= n_prefix <= 64); the bits equal those of a private copy.
 * Prefill: where prefix_page_dev[i] >= 0, the rows of sequence i are positions n_prefix, n_prefix + 1, ... and
 * keys [0, n_prefix) come from that page (layout as written by t3_op_gemm_qkv_rope); -1 = a sequence from position 0. */
int t3_op_attn_decode_prefix(const void* qkv_dev, void* out_dev, int32_t rows, int32_t H, int32_t hd,
                             const int32_t* stream_dev, const int32_t* ctx_dev, const void* kv_pool_dev,
                             int64_t layer_stride_elems, int32_t layer, const int32_t* block_table_dev,
                             int32_t max_pages, const int32_t* prefix_page_dev, int32_t n_prefix, void* stream);
int t3_op_attn_prefill_prefix(const void* qkv_dev, void* out_dev, int32_t n_seqs, const int32_t* seq_start_dev,
                              const int32_t* seq_start_host, int32_t H, int32_t hd, const int32_t* prefix_page_dev,
                              int32_t n_prefix, const void* kv_pool_dev, int64_t layer_stride_elems, int32_t layer,
                              void* stream);
/* logits rows (bf16 [*, ldl]) -> CFG combine, +1000 offset, repetition penalty over seen_dev
 * ([n, seen_stride] bytes, column space), greedy argmax (lowest index wins ties).
 * row_c/row_u: logits row of the cond / uncond stream per request (row_u < 0: no CFG). */
int t3_op_sample_greedy(const void* logits_dev, int32_t ldl, int32_t vocab, const int32_t* row_c_dev,
                        const int32_t* row_u_dev, int32_t n, float cfg_scale, const float* rep_penalty_dev,
                        uint8_t* seen_dev, int32_t seen_stride, int32_t* out_tokens_dev,
                        float* out_logits_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* T3_HIP_H */
