// Attention kernels of the T3 engine (gfx950).  16 heads x 64 dims, no head sharing
// (n_kv_heads == n_heads), so every query row streams its own K and V exactly once:
// HBM-bound vector work, not MFMA work.
//
// KV cache: pool[layer][page][K|V][head][token 0..63][64 dims], bf16.  One page holds 64
// positions of one KV stream; a (page, head) slab of K or V is 8 KiB of contiguous memory.
#include "t3_common.h"

#define ATT_NW 8          // waves per workgroup in the decode kernel

// ------------------------------------------------------------------------------------
// Decode: one workgroup per (head, row).  The context is cut into 64-token tiles (= pages);
// wave w takes tiles w, w+8, ...  Inside a tile a lane owns 8 tokens (it*8 + lane/8) and
// 8 of the 64 dims (lane%8): each of its 16 loads (8 K, 8 V) is a 16-byte piece of a 1 KiB
// contiguous wave-wide read.  q.k is finished by a 3-step butterfly over the 8 lanes that
// share a token, so those lanes also share the softmax weight for the V accumulation.
// Every lane group keeps an online-softmax state (m, l, o[8]); states are merged across
// the 8 groups of a wave by butterflies and across waves through LDS in wave order.  The
// tile -> wave -> merge order depends only on the row's own context length, never on the
// batch, so results are bit-identical for any batch composition.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(ATT_NW * 64) void attn_decode_kernel(
        const bf16_t* __restrict__ qkv, int ldq, bf16_t* __restrict__ out, int ldo,
        const int32_t* __restrict__ row_stream, const int32_t* __restrict__ row_ctx, KvView kv, int layer) {
    __shared__ float part[ATT_NW][8][12];
    const int h = blockIdx.x, r = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 7, tg = lane >> 3;
    const int st = row_stream[r];
    const int ctx = row_ctx[r];
    const int hd = 64;

    float q[8];
    {
        const u32x4 qv = *reinterpret_cast<const u32x4*>(qkv + (size_t)r * ldq + h * hd + c * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) { q[2 * i] = bf_lo(qv[i]); q[2 * i + 1] = bf_hi(qv[i]); }
    }
    float m = -1e30f, l = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;

    const int ntiles = st >= 0 ? (ctx + 63) >> 6 : 0;
    const size_t slab = (size_t)T3_PAGE * hd;                      // one (page, kv, head) slab
    const bf16_t* lbase = kv.pool + (size_t)layer * kv.layer_stride;
    for (int tile = w; tile < ntiles; tile += ATT_NW) {
        const int page = kv.block_table[(size_t)st * kv.max_pages + tile];
        const bf16_t* kb = lbase + ((size_t)page * 2 * kv.H + h) * slab + lane * 8;
        const bf16_t* vb = kb + (size_t)kv.H * slab;
        u32x4 kr[8], vr[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) kr[it] = *reinterpret_cast<const u32x4*>(kb + it * 512);
#pragma unroll
        for (int it = 0; it < 8; ++it) vr[it] = *reinterpret_cast<const u32x4*>(vb + it * 512);
        float s[8];
        float mt = m;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d = fmaf(q[2 * i], bf_lo(kr[it][i]), d);
                d = fmaf(q[2 * i + 1], bf_hi(kr[it][i]), d);
            }
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            s[it] = d * 0.125f;
            const bool valid = tile * 64 + it * 8 + tg < ctx;
            s[it] = valid ? s[it] : -1e30f;
            mt = fmaxf(mt, s[it]);
        }
        const float alpha = __expf(m - mt);
        l *= alpha;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= alpha;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const bool valid = tile * 64 + it * 8 + tg < ctx;
            const float p = valid ? __expf(s[it] - mt) : 0.f;
            l += p;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[2 * i] = fmaf(p, bf_lo(vr[it][i]), o[2 * i]);
                o[2 * i + 1] = fmaf(p, bf_hi(vr[it][i]), o[2 * i + 1]);
            }
        }
        m = mt;
    }
    // merge the 8 token groups of the wave
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float a1 = __expf(m - mn), a2 = __expf(m2 - mn);
        l = l * a1 + l2 * a2;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = o[j] * a1 + __shfl_xor(o[j], off) * a2;
        m = mn;
    }
    if (tg == 0) {
        part[w][c][0] = m; part[w][c][1] = l;
#pragma unroll
        for (int j = 0; j < 8; ++j) part[w][c][2 + j] = o[j];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        float M = part[0][c][0], L = part[0][c][1], O[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) O[j] = part[0][c][2 + j];
        for (int w2 = 1; w2 < ATT_NW; ++w2) {
            const float m2 = part[w2][c][0], l2 = part[w2][c][1];
            const float mn = fmaxf(M, m2);
            const float a1 = __expf(M - mn), a2 = __expf(m2 - mn);
            L = L * a1 + l2 * a2;
#pragma unroll
            for (int j = 0; j < 8; ++j) O[j] = O[j] * a1 + part[w2][c][2 + j] * a2;
            M = mn;
        }
        const float inv = L > 0.f ? 1.0f / L : 0.f;
        u32x4 ov;
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = pack_bf2(O[2 * i] * inv, O[2 * i + 1] * inv);
        *reinterpret_cast<u32x4*>(out + (size_t)r * ldo + h * hd + c * 8) = ov;
    }
}

int launch_attn_decode(const bf16_t* qkv, int ldq, bf16_t* out, int rows, const int32_t* row_stream,
                       const int32_t* row_ctx, const KvView& kv, int layer, hipStream_t st) {
    T3_REQUIRE(kv.hd == 64, "attn_decode: head_dim must be 64");
    T3_REQUIRE(rows >= 1 && ldq % 8 == 0, "attn_decode: bad shape");
    hipLaunchKernelGGL(attn_decode_kernel, dim3(kv.H, rows), dim3(ATT_NW * 64), 0, st, qkv, ldq, out,
                       kv.H * kv.hd, row_stream, row_ctx, kv, layer);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Prefill: causal attention inside packed prompt rows, read from the step's q|k|v buffer
// (row-major [rows, 3*H*64]).  One wave per (sequence, 64-query tile, head); a lane owns
// one query row (q and the output row live in registers), keys are walked in order and the
// key/value row of the step is the same for all lanes, so it arrives through the scalar
// path.  fp32 online softmax, four keys per update.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void attn_prefill_kernel(const bf16_t* __restrict__ qkv, int ldq,
                                                          bf16_t* __restrict__ out, int ldo,
                                                          const int32_t* __restrict__ seq_start,
                                                          const int32_t* __restrict__ tiles, int H) {
    const int hd = 64;
    const int seq = tiles[2 * blockIdx.x], q0 = tiles[2 * blockIdx.x + 1];
    const int h = blockIdx.y;
    const int s0 = seq_start[seq], P = seq_start[seq + 1] - s0;
    const int lane = threadIdx.x;
    const int row = q0 + lane;
    const int lrow = row < P ? row : P - 1;
    const int D = H * hd;

    float q[64], o[64];
    {
        const bf16_t* qp = qkv + (size_t)(s0 + lrow) * ldq + h * hd;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(qp + j * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) { q[j * 8 + 2 * i] = bf_lo(v[i]) * 0.125f; q[j * 8 + 2 * i + 1] = bf_hi(v[i]) * 0.125f; }
        }
    }
#pragma unroll
    for (int j = 0; j < 64; ++j) o[j] = 0.f;
    float m = -1e30f, l = 0.f;
    const int nkeys = (q0 + 64 < P) ? q0 + 64 : P;
    const bf16_t* kbase = qkv + (size_t)s0 * ldq + D + h * hd;
    const bf16_t* vbase = kbase + D;
    for (int key = 0; key < nkeys; key += 4) {
        float s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = key + u < nkeys ? key + u : nkeys - 1;
            const unsigned* kp = reinterpret_cast<const unsigned*>(kbase + (size_t)kk * ldq);
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const unsigned wv = kp[j];
                d = fmaf(q[2 * j], bf_lo(wv), d);
                d = fmaf(q[2 * j + 1], bf_hi(wv), d);
            }
            s[u] = (key + u <= lrow && key + u < nkeys) ? d : -1e30f;
        }
        const float mt = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), m);
        const float alpha = __expf(m - mt);
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = s[u] > -1e29f ? __expf(s[u] - mt) : 0.f;
        l = l * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
#pragma unroll
        for (int j = 0; j < 64; ++j) o[j] *= alpha;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = key + u < nkeys ? key + u : nkeys - 1;
            const unsigned* vp = reinterpret_cast<const unsigned*>(vbase + (size_t)kk * ldq);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const unsigned wv = vp[j];
                o[2 * j] = fmaf(p[u], bf_lo(wv), o[2 * j]);
                o[2 * j + 1] = fmaf(p[u], bf_hi(wv), o[2 * j + 1]);
            }
        }
        m = mt;
    }
    if (row < P) {
        const float inv = 1.0f / l;
        bf16_t* op = out + (size_t)(s0 + row) * ldo + h * hd;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = pack_bf2(o[j * 8 + 2 * i] * inv, o[j * 8 + 2 * i + 1] * inv);
            *reinterpret_cast<u32x4*>(op + j * 8) = v;
        }
    }
}

int launch_attn_prefill(const bf16_t* qkv, int ldq, bf16_t* out, const int32_t* seq_start_dev,
                        const int32_t* tiles_dev, int ntiles, int H, int hd, hipStream_t st) {
    T3_REQUIRE(hd == 64 && ntiles >= 1, "attn_prefill: head_dim must be 64");
    hipLaunchKernelGGL(attn_prefill_kernel, dim3(ntiles, H), dim3(64), 0, st, qkv, ldq, out, H * hd, seq_start_dev,
                       tiles_dev, H);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}
