// Row-wise kernels of the T3 engine: residual-add + RMSNorm, embedding assembly, and the
// CFG + repetition-penalty + sampling step.  All are HBM/L2-bound byte movers: one workgroup
// per row, 16-byte loads, wave shuffles for the reductions.
#include "t3_common.h"

// ------------------------------------------------------------------------------------
// res = bf16(x + res) (x optional) ; xn = bf16(bf16(res * rsqrt(mean(res^2) + eps)) * w)
// One workgroup of 128 threads per row, 8 elements per thread per pass.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void add_rmsnorm_kernel(const bf16_t* __restrict__ x, bf16_t* res,
                                                          const bf16_t* __restrict__ w, bf16_t* __restrict__ xn,
                                                          int d, float eps) {
    __shared__ float part[2];
    const size_t row = blockIdx.x;
    const int tid = threadIdx.x;
    float ss = 0.f;
    for (int c = tid * 8; c < d; c += 128 * 8) {
        u32x4 rv = *reinterpret_cast<const u32x4*>(res + row * d + c);
        float z[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { z[2 * i] = bf_lo(rv[i]); z[2 * i + 1] = bf_hi(rv[i]); }
        if (x) {
            const u32x4 xv = *reinterpret_cast<const u32x4*>(x + row * d + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                z[2 * i] = rbf(z[2 * i] + bf_lo(xv[i]));
                z[2 * i + 1] = rbf(z[2 * i + 1] + bf_hi(xv[i]));
            }
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = pack_bf2(z[2 * i], z[2 * i + 1]);
            *reinterpret_cast<u32x4*>(res + row * d + c) = o;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += z[i] * z[i];
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float tot = part[0] + part[1];
    const float sc = 1.0f / sqrtf(tot / (float)d + eps);
    for (int c = tid * 8; c < d; c += 128 * 8) {
        const u32x4 rv = *reinterpret_cast<const u32x4*>(res + row * d + c);   // own writes, program order
        const u32x4 wv = *reinterpret_cast<const u32x4*>(w + c);
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = rbf(bf_lo(rv[i]) * sc) * bf_lo(wv[i]);
            const float b = rbf(bf_hi(rv[i]) * sc) * bf_hi(wv[i]);
            o[i] = pack_bf2(a, b);
        }
        *reinterpret_cast<u32x4*>(xn + row * d + c) = o;
    }
}

// ------------------------------------------------------------------------------------
// Decode form, d = 1024: the row (8 elements per thread) stays in registers between the
// sum of squares and the scaling, and the split-K partial sums of the preceding projection
// are folded in first:  res = bf16(bf16(slab[0] + slab[1] + ... ) + res)  in slab order.
// ------------------------------------------------------------------------------------
#define SLAB_NORM_THREADS 128      // 256 threads (4 elements each) measured the same: the kernel is one memory round trip
template <int NSPLIT>
__global__ __launch_bounds__(SLAB_NORM_THREADS) void slab_add_rmsnorm_kernel(const float* __restrict__ slab,
                                                               size_t slab_stride, bf16_t* __restrict__ res,
                                                               const bf16_t* __restrict__ w, bf16_t* __restrict__ xn,
                                                               float eps, int xn_tiled, const bf16_t* __restrict__ emb,
                                                               const int32_t* __restrict__ emb_slot,
                                                               const int32_t* __restrict__ last_token,
                                                               const float* __restrict__ pair_scale) {
    constexpr int d = 1024;
    __shared__ float part[2];
    const size_t row = blockIdx.x;
    const int tid = threadIdx.x, c = tid * 8;
    float z[8];
    const u32x4 wv = *reinterpret_cast<const u32x4*>(w + c);
    if (NSPLIT == 0 && emb != nullptr) {
        // first norm of a decode step: the row is the embedding of the token its request sampled last
        // (models/t3/t3.py:503-514), fetched here and stored as the residual row, instead of by a launch of its own
        const int id = last_token[emb_slot[row]];
        const u32x4 ev = *reinterpret_cast<const u32x4*>(emb + (size_t)id * d + c);
        *reinterpret_cast<u32x4*>(res + row * d + c) = ev;
        unpack8(ev, z);
    } else {
        unpack8(*reinterpret_cast<const u32x4*>(res + row * d + c), z);
    }
    if constexpr (NSPLIT > 0) {
        norm_fold8<NSPLIT>(slab + row * d + c, slab_stride, z, pair_scale ? pair_scale + c : nullptr);
        *reinterpret_cast<u32x4*>(res + row * d + c) = pack8(z);
    }
    const float ss = wave_sum(norm_ssq8(z));
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float sc = norm_rstd(part[0] + part[1], eps);
    // activation tile order (t3_common.h): this thread's 8 columns are one 16-byte fragment piece
    const size_t off = xn_tiled ? ((row >> 4) * (d >> 5) + (tid >> 2)) * 512 + ((tid & 3) * 16 + (row & 15)) * 8
                                : row * d + c;
    *reinterpret_cast<u32x4*>(xn + off) = norm_scale8(z, sc, wv);
}

int launch_slab_add_rmsnorm(const float* slab, int nsplit, bf16_t* res, const bf16_t* w, bf16_t* xn, int rows, int d,
                            float eps, hipStream_t st, bool xn_tiled, const bf16_t* emb, const int32_t* emb_slot,
                            const int32_t* last_token, const float* pair_scale) {
    T3_REQUIRE(pair_scale == nullptr || nsplit == ATTN_OPROJ_SPLIT, "slab_add_rmsnorm: a pair scale belongs to the 16 head partials");
    T3_REQUIRE(emb == nullptr || (nsplit == 0 && emb_slot && last_token), "slab_add_rmsnorm: the embedding gather belongs to the first norm");
    T3_REQUIRE(rows >= 1 && d == 1024, "slab_add_rmsnorm: d must be 1024");
    static_assert(SLAB_NORM_THREADS == 128, "8 columns per thread, two waves per row");
    T3_REQUIRE(nsplit == 0 || (slab != nullptr && (nsplit == GEMM_SLAB_SPLIT || nsplit == ATTN_OPROJ_SPLIT)),
               "slab_add_rmsnorm: nsplit must be 0, the split-K count (8) or the head count (16)");
    if (nsplit == ATTN_OPROJ_SPLIT)
        T3_LAUNCH(slab_add_rmsnorm_kernel<ATTN_OPROJ_SPLIT>, dim3(rows), dim3(SLAB_NORM_THREADS), 0, st, slab, (size_t)rows * d, res, w, xn,
                  eps, (int)xn_tiled, emb, emb_slot, last_token, pair_scale);
    else if (nsplit == 0)
        T3_LAUNCH(slab_add_rmsnorm_kernel<0>, dim3(rows), dim3(SLAB_NORM_THREADS), 0, st, slab, (size_t)rows * d, res, w, xn, eps,
                  (int)xn_tiled, emb, emb_slot, last_token, pair_scale);
    else
        T3_LAUNCH(slab_add_rmsnorm_kernel<GEMM_SLAB_SPLIT>, dim3(rows), dim3(SLAB_NORM_THREADS), 0, st, slab, (size_t)rows * d, res, w, xn,
                  eps, (int)xn_tiled, emb, emb_slot, last_token, pair_scale);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_add_rmsnorm(const bf16_t* x, bf16_t* res, const bf16_t* w, bf16_t* xn, int rows, int d,
                       float eps, hipStream_t st) {
    T3_REQUIRE(rows >= 1 && d % 8 == 0, "add_rmsnorm: d must be a multiple of 8");
    T3_LAUNCH(add_rmsnorm_kernel, dim3(rows), dim3(128), 0, st, x, res, w, xn, d, eps);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Embedding rows.  kind 0: conditioning row (fp32 -> bf16); 1: text_emb[id] + text_pos[p];
// 2: zeros (uncond text row); 3: speech_emb[SOS] + speech_pos[0]; 4: speech_emb[last_token[slot]].
// ------------------------------------------------------------------------------------
// Each row kind is a straight-line body with its own store: a shared store after a chain of
// branches was miscompiled by hipcc 7.2 (the store's column register was left undefined on one
// path), so no value is carried across the kind dispatch.
__device__ __forceinline__ void embed_sum_row(bf16_t* __restrict__ dst, const bf16_t* __restrict__ e,
                                              const bf16_t* __restrict__ p, int d) {
    for (int c = threadIdx.x * 8; c < d; c += 128 * 8) {
        const u32x4 ev = *reinterpret_cast<const u32x4*>(e + c);
        const u32x4 pv = *reinterpret_cast<const u32x4*>(p + c);
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack_bf2(bf_lo(ev[i]) + bf_lo(pv[i]), bf_hi(ev[i]) + bf_hi(pv[i]));
        *reinterpret_cast<u32x4*>(dst + c) = o;
    }
}

__global__ __launch_bounds__(128) void embed_kernel(EmbedArgs a) {
    const size_t row = blockIdx.x;
    const int kind = __builtin_amdgcn_readfirstlane(a.kind[row]);
    const int idx = __builtin_amdgcn_readfirstlane(a.idx[row]);
    const int aux = __builtin_amdgcn_readfirstlane(a.aux[row]);
    bf16_t* dst = a.out + row * a.d;
    if (kind == 0) {
        const float* src = a.cond + ((size_t)aux * 34 + idx) * a.d;
        for (int c = threadIdx.x * 8; c < a.d; c += 128 * 8) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + c);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(src + c + 4);
            const u32x4 o = {pack_bf2(v0[0], v0[1]), pack_bf2(v0[2], v0[3]), pack_bf2(v1[0], v1[1]), pack_bf2(v1[2], v1[3])};
            *reinterpret_cast<u32x4*>(dst + c) = o;
        }
        return;
    }
    if (kind == 1) {
        embed_sum_row(dst, a.text_emb + (size_t)idx * a.d, a.text_pos + (size_t)aux * a.d, a.d);
        return;
    }
    if (kind == 3) {
        embed_sum_row(dst, a.speech_emb + (size_t)a.sos_id * a.d, a.speech_pos, a.d);
        return;
    }
    if (kind == 4) {
        const bf16_t* src = a.speech_emb + (size_t)a.last_token[idx] * a.d;
        for (int c = threadIdx.x * 8; c < a.d; c += 128 * 8)
            *reinterpret_cast<u32x4*>(dst + c) = *reinterpret_cast<const u32x4*>(src + c);
        return;
    }
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (int c = threadIdx.x * 8; c < a.d; c += 128 * 8) *reinterpret_cast<u32x4*>(dst + c) = zero;
}

int launch_embed(const EmbedArgs& a, hipStream_t st) {
    T3_REQUIRE(a.rows >= 1 && a.d % 8 == 0, "embed: bad shape");
    T3_LAUNCH(embed_kernel, dim3(a.rows), dim3(128), 0, st, a);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Sampler: one workgroup per request.
//   l[v]   = bf16(c + bf16(s * bf16(c - u)))          CFG, three bf16 roundings (or c alone)
//   column = v + 1000; repetition penalty on seen columns: l > 0 ? l / p : l * p     (fp32)
//   greedy : argmax, lowest index wins ties
//   sampled: l / T -> min_p mask -> top_p nucleus -> argmax(prob / Exp(1)) with a
//            counter-based generator keyed by (seed, step, column)
// The chosen column is marked in `seen` and the un-offset id stored for the next step.
// ------------------------------------------------------------------------------------
#define SAMPLE_THREADS 1024     // 8 columns per thread: the whole speech vocabulary in one pass
#define SAMPLE_MAX_VOCAB 8448

__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// uniform in (0,1] from a (seed, step, column) key
__device__ __forceinline__ float key_uniform(uint64_t seed, unsigned step, unsigned col) {
    unsigned h = mix32((unsigned)seed ^ 0x9e3779b9U);
    h = mix32(h ^ (unsigned)(seed >> 32));
    h = mix32(h ^ (step * 0x85ebca6bU + 0xc2b2ae35U));
    h = mix32(h ^ (col * 0x27d4eb2fU + 0x165667b1U));
    return ((float)(h >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

struct MaxIdx { float v; int i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ MaxIdx block_argmax(MaxIdx m, MaxIdx* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MaxIdx t; t.v = __shfl_xor(m.v, o); t.i = __shfl_xor(m.i, o);
        m = better(m, t);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    MaxIdx r = sh[0];
    for (int k = 1; k < SAMPLE_THREADS / 64; ++k) r = better(r, sh[k]);
    return r;
}
__device__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int k = 0; k < SAMPLE_THREADS / 64; ++k) r += sh[k];
    return r;
}

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_kernel(SampleArgs a) {
    __shared__ float sl[SAMPLE_MAX_VOCAB];
    __shared__ MaxIdx shm[SAMPLE_THREADS / 64];
    __shared__ float shf[SAMPLE_THREADS / 64];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int rc = a.row_c[i], ru = a.row_u[i];
    const int slot = a.slot ? a.slot[i] : i;
    uint8_t* seen = a.seen + (size_t)slot * a.seen_stride;
    const float pen = a.rep_penalty[i];
    const float T = a.temperature ? a.temperature[i] : 0.f;          // fetched up front, not behind the argmax
    const bf16_t* lc = a.logits + (size_t)rc * a.ldl;
    const bf16_t* lu = ru >= 0 ? a.logits + (size_t)ru * a.ldl : nullptr;
    float* ol = a.out_logits ? a.out_logits + (size_t)slot * (a.vocab + 1000) : nullptr;
    if (ol) for (int v = tid; v < 1000; v += SAMPLE_THREADS) ol[v] = -INFINITY;

    MaxIdx best; best.v = -INFINITY; best.i = 0x7fffffff;
    // 8 consecutive columns per thread and pass: 16-byte logit loads, one 8-byte load of seen flags
    // (rows are 16-byte aligned: ldl % 8 == 0, seen stride % 8 == 0, and 1000 % 8 == 0)
    for (int v0 = tid * 8; v0 < a.vocab; v0 += SAMPLE_THREADS * 8) {
        float c[8], u[8];
        unsigned char sf[8];
        if (v0 + 8 <= a.vocab) {
            const u32x4 cv = *reinterpret_cast<const u32x4*>(lc + v0);
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[2 * k] = bf_lo(cv[k]); c[2 * k + 1] = bf_hi(cv[k]); }
            if (lu) {
                const u32x4 uv = *reinterpret_cast<const u32x4*>(lu + v0);
#pragma unroll
                for (int k = 0; k < 4; ++k) { u[2 * k] = bf_lo(uv[k]); u[2 * k + 1] = bf_hi(uv[k]); }
            }
            const u32x2 sv = *reinterpret_cast<const u32x2*>(seen + v0 + 1000);
#pragma unroll
            for (int k = 0; k < 8; ++k) sf[k] = (unsigned char)((sv[k >> 2] >> (8 * (k & 3))) & 0xff);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int v = v0 + k < a.vocab ? v0 + k : a.vocab - 1;
                c[k] = bf2f(lc[v]); u[k] = lu ? bf2f(lu[v]) : 0.f; sf[k] = seen[v + 1000];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = v0 + k;
            if (v < a.vocab) {
                float l = c[k];
                if (lu) l = rbf(c[k] + rbf(a.cfg_scale * rbf(c[k] - u[k])));
                if (pen != 1.0f && sf[k]) l = l > 0.f ? l / pen : l * pen;
                sl[v] = l;
                if (ol) ol[v + 1000] = l;
                if (l > best.v) { best.v = l; best.i = v; }
            }
        }
    }
    best = block_argmax(best, shm);
    int tok = best.i;

    if (T >= 1e-5f) {
        // logits / T ; everything below is relative to the maximum so no overflow
        const float lmax = best.v / T;
        const float minp = a.min_p ? a.min_p[i] : 0.f;
        const float thr = minp > 0.f ? lmax + logf(minp) : -INFINITY;   // p_v < min_p * p_max
        float z = 0.f;
        for (int v = tid; v < a.vocab; v += SAMPLE_THREADS) {
            float l = sl[v] / T;
            float e = (l < thr) ? 0.f : expf(l - lmax);
            sl[v] = e;                                                 // unnormalised probability
            z += e;
        }
        z = block_sum(z, shf);
        const float topp = a.top_p ? a.top_p[i] : 1.f;
        float cut = 0.f;                                                // keep e >= cut
        if (topp < 1.f) {
            // smallest-mass nucleus: largest cut with mass{e >= cut} >= top_p * z  (bisection on e)
            float lo = 0.f, hi = 1.f;                                   // e in (0, 1]
            for (int it = 0; it < 30; ++it) {
                const float mid = 0.5f * (lo + hi);
                float mass = 0.f;
                for (int v = tid; v < a.vocab; v += SAMPLE_THREADS) mass += sl[v] >= mid ? sl[v] : 0.f;
                mass = block_sum(mass, shf);
                if (mass >= topp * z) lo = mid; else hi = mid;
            }
            cut = lo;
        }
        const uint64_t seed = a.seed ? a.seed[i] : 0;
        const unsigned step = a.step ? (unsigned)a.step[i] : 0u;
        MaxIdx r; r.v = -INFINITY; r.i = 0x7fffffff;
        for (int v = tid; v < a.vocab; v += SAMPLE_THREADS) {
            const float e = sl[v];
            if (e > 0.f && e >= cut) {
                const float q = -logf(key_uniform(seed, step, (unsigned)v));   // Exp(1)
                const float sc = e / fmaxf(q, 1e-30f);
                if (sc > r.v) { r.v = sc; r.i = v; }
            }
        }
        r = block_argmax(r, shm);
        tok = r.i;
    }
    if (tid == 0) {
        // A row without a finite maximum (NaN or all -inf logits: a bad checkpoint, a non-finite conditioning row) has no
        // argmax: report -1 for the caller to turn into an error, and leave id 0 behind so that the next step's
        // embedding gather and the seen row stay inside their tables.
        const bool ok = (unsigned)tok < (unsigned)a.vocab;
        if (!ok) tok = 0;
        a.out_tokens[i] = ok ? tok + 1000 : -1;
        seen[tok + 1000] = 1;
        if (a.last_token) a.last_token[slot] = tok;
    }
}

int launch_sample(const SampleArgs& a, hipStream_t st) {
    T3_REQUIRE(a.n >= 1 && a.vocab <= SAMPLE_MAX_VOCAB, "sample: vocab too large");
    T3_REQUIRE(a.seen_stride >= a.vocab + 1000 && a.seen_stride % 8 == 0, "sample: seen rows too short or unaligned");
    T3_REQUIRE(a.ldl % 8 == 0, "sample: logits row stride must be a multiple of 8");
    T3_LAUNCH(sample_kernel, dim3(a.n), dim3(SAMPLE_THREADS), 0, st, a);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// seen row := {ids}
__global__ __launch_bounds__(256) void seen_init_kernel(uint8_t* seen, int stride, const int32_t* ids, int n) {
    for (int c = threadIdx.x * 4; c < stride; c += 256 * 4) *reinterpret_cast<unsigned*>(seen + c) = 0u;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const int id = ids[k];
        if (id >= 0 && id < stride) seen[id] = 1;
    }
}

int launch_seen_init(uint8_t* seen_row, int seen_stride, const int32_t* ids_dev, int n_ids, hipStream_t st) {
    T3_REQUIRE(seen_stride % 4 == 0, "seen_init: stride must be a multiple of 4");
    T3_LAUNCH(seen_init_kernel, dim3(1), dim3(256), 0, st, seen_row, seen_stride, ids_dev, n_ids);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}
