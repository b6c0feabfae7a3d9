// Attention kernels of the T3 engine (gfx950).  16 heads x 64 dims, no head sharing
// (n_kv_heads == n_heads), so every query row streams its own K and V exactly once:
// HBM-bound vector work, not MFMA work.
//
// KV cache: pool[layer][page][K|V][head][token 0..63][64 dims], bf16.  One page holds 64
// positions of one KV stream; a (page, head) slab of K or V is 8 KiB of contiguous memory.
#include "t3_common.h"

// K and V are read once per step and never again before ~1 GB of other traffic: nontemporal loads
// keep them from displacing the step's activations in L2 (measured: 12 % less time per launch).
#ifndef ATT_LOAD
#define ATT_LOAD(p) __builtin_nontemporal_load(p)
#endif
#ifndef ATT_NW
#define ATT_NW 8          // waves per workgroup in the decode kernel
#endif

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
//
// Output projection in the same launch (OPROJ; the engine keeps it OFF, T3_FUSED_OPROJ=1 turns it on): the 64 values a
// workgroup ends with are exactly the K range of o_proj that belongs to its head, so the workgroup can multiply them
// (rounded to bf16, as the attention output is) with Wo[:, 64h .. 64h+64) itself and write the fp32 partial sums to
// slab[h][row][0..1024): o_proj as a split-K over the 16 heads, one dependent launch fewer per layer.  The fold adds
// heads 2s and 2s+1 and then the eight pair sums in order: the additions of the stand-alone projection's eight split-K
// slabs, so both forms give the same bits (int8 column scales are applied by the fold).  The weight tiles (128 KiB per
// head, the tile order of gemm.hip; all rows of a head run on one XCD) feed 16 MFMAs per wave with the group's rows in
// the first columns of the B operand.  Measured on the benchmark job (62 rows): the attention launch is bound by what a
// CU's vector memory path moves, and every extra byte through it, L2 hits included, costs what a K/V byte costs:
// one workgroup per (head, row) re-reads 127 MB of Wo per launch, 21.2 -> 26.6 us; four rows per workgroup (one
// workgroup per CU, two tiles in flight per wave) 26.5 us, two rows 25.4 us, against 21.6 + 4.4 us for the two launches;
// with one row (single stream) 16 workgroups cannot pull Wo at the chip's rate: 1287 against 1333 tokens/s.
//
// Shared prefix: when row_pfx[r] >= 0, positions [0, npfx) of row r live in page row_pfx[r]
// (one copy for every stream that starts with the same conditioning rows) and the row's own
// page 0 holds only positions [npfx, 64).  The tile boundaries stay where they are, each lane
// just picks the page its token is in, so the result has the same bits as with private
// copies, and all rows of a step read the same few KiB, which then come from L2.
// ------------------------------------------------------------------------------------
// 8 offset-binary bytes (two dwords) -> 8 bf16 values u - 128 (int8 weights, as in gemm.hip)
__device__ __forceinline__ bf16x8 att_dequant8(unsigned lo, unsigned hi) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (__bf16)((float)((lo >> (8 * j)) & 0xffu) - 128.0f);
        r[4 + j] = (__bf16)((float)((hi >> (8 * j)) & 0xffu) - 128.0f);
    }
    return r;
}

// One 64-token tile of one (row, head) in a wave's registers: 8 K and 8 V pieces of 16 bytes per lane.
struct AttTile { u32x4 k[8], v[8]; };

// Requests the tile's 16 loads.  Lanes whose token lies in the shared prefix read the prefix page (tile 0 only): an
// element offset between the two pages, zero everywhere else.  With ATT_TRIM, tokens at or beyond `ctx` (the tail of a
// row's last tile) are not fetched: their registers are zero, the fold gives them weight 0 whatever they hold, and a launch
// moves the bytes of the positions that exist instead of whole tiles (7 % fewer at a mean context of 443).  Measured: 2.4 %
// faster at 222 rows (byte-bound), 1.4 % SLOWER at 62 rows (the launch is ramp and tail there, and the predicated loads
// cost issue slots): off by default.
#ifndef ATT_TRIM
#define ATT_TRIM 0
#endif
__device__ __forceinline__ void att_load_tile(AttTile& t, const bf16_t* __restrict__ lbase, const KvView& kv, int st, int tile,
                                              int h, int lane, int pfx, int npfx, int ctx) {
    const int tg = lane >> 3;
    const size_t slab = (size_t)T3_PAGE * 64;                      // one (page, kv, head) slab
    const int page = kv.block_table[(size_t)st * kv.max_pages + tile];
    const bf16_t* kb = lbase + ((size_t)page * 2 * kv.H + h) * slab + lane * 8;
    const bool shared = tile == 0 && pfx >= 0;
    const long long to_pfx = shared ? ((long long)pfx - page) * (long long)(2 * kv.H) * (long long)slab : 0;
    const int lim = shared ? npfx : 0;
    const size_t vofs = (size_t)kv.H * slab;
    const int nv = ATT_TRIM ? ctx - tile * 64 : 64;                 // tokens of this tile that exist
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const bf16_t* p = kb + it * 512 + (it * 8 + tg < lim ? to_pfx : 0);
        t.k[it] = u32x4{0u, 0u, 0u, 0u};
        if (it * 8 + tg < nv) t.k[it] = ATT_LOAD(reinterpret_cast<const u32x4*>(p));
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const bf16_t* p = kb + vofs + it * 512 + (it * 8 + tg < lim ? to_pfx : 0);
        t.v[it] = u32x4{0u, 0u, 0u, 0u};
        if (it * 8 + tg < nv) t.v[it] = ATT_LOAD(reinterpret_cast<const u32x4*>(p));
    }
}

// Folds the tile into the lane group's online-softmax state (m, l, o[8]).
__device__ __forceinline__ void att_fold_tile(const AttTile& t, const float (&q)[8], int tile, int ctx, int tg, float& m, float& l,
                                              float (&o)[8]) {
    float s[8];
    float mt = m;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d = fmaf(q[2 * i], bf_lo(t.k[it][i]), d);
            d = fmaf(q[2 * i + 1], bf_hi(t.k[it][i]), d);
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
            o[2 * i] = fmaf(p, bf_lo(t.v[it][i]), o[2 * i]);
            o[2 * i + 1] = fmaf(p, bf_hi(t.v[it][i]), o[2 * i + 1]);
        }
    }
    m = mt;
}

// Merges the 8 token groups of the wave (butterflies over lane bits 3..5).
__device__ __forceinline__ void att_merge_groups(float& m, float& l, float (&o)[8]) {
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float a1 = __expf(m - mn), a2 = __expf(m2 - mn);
        l = fmaf(l, a1, l2 * a2);                     // explicit, so that every instantiation rounds alike
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(o[j], a1, __shfl_xor(o[j], off) * a2);
        m = mn;
    }
}

// RPW rows per workgroup.  RPW = 1: workgroup = (head, row), two workgroups per CU (second launch bound: 4 waves per SIMD, at
// most 128 VGPRs).  RPW = 4 (full decode steps, with OPROJ): workgroup = (head, group of 4 rows); wave w takes tile w (w + 8,
// ...) of EVERY row of the group, one row after the other, exactly as it would in four workgroups of its own, so a row's bits
// do not depend on the grouping; the first tile of the next row is requested before the current row's tile is folded (two tile
// buffers, so every wave keeps two tiles in flight), one workgroup per CU (256 VGPRs), and the head's 128 KiB of Wo are read
// once per four rows.  The host orders the rows so that the groups hold about the same number of positions.
template <int RPW, bool OPROJ, bool W8>
__global__ __launch_bounds__(ATT_NW * 64, (RPW > 2 || (OPROJ && RPW == 1)) ? 2 : 4) void attn_decode_kernel(
        const bf16_t* __restrict__ qkv, int ldq, bf16_t* __restrict__ out, int ldo, int rows,
        const int32_t* __restrict__ row_stream, const int32_t* __restrict__ row_ctx, KvView kv, int layer,
        const int32_t* __restrict__ row_pfx, int npfx,
        const bf16_t* __restrict__ wo, float* __restrict__ oslab
#ifdef T3_ATTN_TIMELINE
        , long long* ts
#endif
        ) {
#ifdef T3_ATTN_TIMELINE
#define ATT_STAMP(i) do { if ((threadIdx.x & 63) == 0) ts[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ATT_NW + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif
    ATT_STAMP(0);
    __shared__ float part[RPW][ATT_NW][8][ATT_STATE_FLOATS];
    __shared__ __attribute__((aligned(16))) bf16_t o_sh[RPW][64];
    const int h = blockIdx.x, r0 = blockIdx.y * RPW;
    const int nr = rows - r0 < RPW ? rows - r0 : RPW;              // rows of this group
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 7, tg = lane >> 3;
    const int hd = 64;
    const bf16_t* lbase = kv.pool + (size_t)layer * kv.layer_stride;

    int st[RPW], ctx[RPW], pfx[RPW], nt[RPW];
    float q[RPW][8], m[RPW], l[RPW], o[RPW][8];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = r0 + (i < nr ? i : 0);
        st[i] = row_stream[r];
        ctx[i] = row_ctx[r];
        pfx[i] = row_pfx ? row_pfx[r] : -1;
        nt[i] = (i < nr && st[i] >= 0) ? (ctx[i] + 63) >> 6 : 0;
        const u32x4 qv = *reinterpret_cast<const u32x4*>(qkv + (size_t)r * ldq + h * hd + c * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) { q[i][2 * k] = bf_lo(qv[k]); q[i][2 * k + 1] = bf_hi(qv[k]); }
        m[i] = -1e30f; l[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[i][k] = 0.f;
    }

    // OPROJ, one row per workgroup (launches of a few rows): the weight tiles are requested before the K/V stream
    constexpr int OP_NB = 1024 / 16 / ATT_NW;              // column blocks per wave
    static_assert(!OPROJ || (ATT_NW == 8 && OP_NB == 8), "the output projection is laid out for 8 waves x 8 column blocks");
    bf16x8 wt[OPROJ && !W8 ? OP_NB : 1][2];
    u32x4 wq[OPROJ && W8 ? OP_NB : 1];
    auto load_wo = [&]() {
#pragma unroll
        for (int i = 0; i < OP_NB; ++i) {
            const int nb = w * OP_NB + i;
            if constexpr (W8) {
                wq[i] = *(reinterpret_cast<const u32x4*>(wo) + ((size_t)nb * 16 + h) * 64 + lane);
            } else {
                const bf16x8* wp = reinterpret_cast<const bf16x8*>(wo) + ((size_t)nb * 32 + 2 * h) * 64 + lane;
                wt[i][0] = wp[0];
                wt[i][1] = wp[64];
            }
        }
    };
    if constexpr (OPROJ && RPW == 1) load_wo();

    constexpr bool DBUF = RPW > 2;                         // two tile buffers: the next row's first tile is requested ahead
    AttTile buf[DBUF ? 2 : 1];
    ATT_STAMP(1);
    if (w < nt[0]) att_load_tile(buf[0], lbase, kv, st[0], w, h, lane, pfx[0], npfx, ctx[0]);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        if constexpr (DBUF) {
            if (i + 1 < RPW) {
                if (w < nt[i + 1 < RPW ? i + 1 : 0])
                    att_load_tile(buf[(i + 1) & 1], lbase, kv, st[i + 1 < RPW ? i + 1 : 0], w, h, lane, pfx[i + 1 < RPW ? i + 1 : 0], npfx, ctx[i + 1 < RPW ? i + 1 : 0]);
            }
        }
        AttTile& cur = buf[DBUF ? (i & 1) : 0];
        if constexpr (!DBUF) { if (i > 0 && w < nt[i]) att_load_tile(cur, lbase, kv, st[i], w, h, lane, pfx[i], npfx, ctx[i]); }
        if (w < nt[i]) att_fold_tile(cur, q[i], w, ctx[i], tg, m[i], l[i], o[i]);
        for (int tile = w + ATT_NW; tile < nt[i]; tile += ATT_NW) {      // contexts beyond 512 positions
            att_load_tile(cur, lbase, kv, st[i], tile, h, lane, pfx[i], npfx, ctx[i]);
            att_fold_tile(cur, q[i], tile, ctx[i], tg, m[i], l[i], o[i]);
        }
    }
    ATT_STAMP(2);
    // OPROJ, grouped rows: this wave's 16 weight tiles of Wo (output columns 128w .. 128w+128 x the head's 64 inputs),
    // requested now (the tile buffers are free), used after the merge.  bf16: tile (nb, kb) of 1 KiB, kb = 2h, 2h+1; int8:
    // one pair tile per nb holds both.  All rows of a head run on one XCD, so after the first group they come from its L2.
    if constexpr (OPROJ && RPW > 1) load_wo();
    ATT_STAMP(3);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        att_merge_groups(m[i], l[i], o[i]);
        if (tg == 0) {
            part[i][w][c][0] = m[i]; part[i][w][c][1] = l[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) part[i][w][c][2 + j] = o[i][j];
        }
    }
    __syncthreads();
    if (threadIdx.x < 8 * RPW) {
        const int i = threadIdx.x >> 3;                     // row of the group; c = threadIdx.x & 7
        const u32x4 ov = att_merge_waves(&part[i][0][c][0], 8 * ATT_STATE_FLOATS);
        if constexpr (OPROJ) *reinterpret_cast<u32x4*>(&o_sh[i][c * 8]) = ov;
        else if (i < nr) *reinterpret_cast<u32x4*>(out + (size_t)(r0 + i) * ldo + h * hd + c * 8) = ov;
    }
    if constexpr (OPROJ) {
        __syncthreads();
        // B operand of D = Wtile(16 n x 32 k) * B(32 k x 16 columns): row i of the group sits in column i, the rest is zero
        const int col = lane & 15;
        const bool live = col < nr;
        bf16x8 bfrag[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            u32x4 v = *reinterpret_cast<const u32x4*>(&o_sh[live ? col : 0][32 * kt + 8 * (lane >> 4)]);
            if (!live) v = u32x4{0u, 0u, 0u, 0u};
            bfrag[kt] = __builtin_bit_cast(bf16x8, v);
        }
        float* srow = oslab + ((size_t)h * rows + r0 + (live ? col : 0)) * 1024 + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < OP_NB; ++i) {
            const int nb = w * OP_NB + i;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (W8) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_dequant8(wq[i][0], wq[i][1]), bfrag[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_dequant8(wq[i][2], wq[i][3]), bfrag[1], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[i][0], bfrag[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[i][1], bfrag[1], acc, 0, 0, 0);
            }
            if (live) *reinterpret_cast<f32x4*>(srow + nb * 16) = acc;
        }
    }
    ATT_STAMP(4);
}

// Few-row form: grid (head, row, wave): the work of ONE wave of attn_decode_kernel<1, ...> per single-wave workgroup, its
// merged lane-group state stored to states[((row * H + head) * ATT_NW + wave) * 8 + c][ATT_STATE_FLOATS].
__global__ __launch_bounds__(64) void attn_decode_split_kernel(
        const bf16_t* __restrict__ qkv, int ldq, float* __restrict__ states,
        const int32_t* __restrict__ row_stream, const int32_t* __restrict__ row_ctx, KvView kv, int layer,
        const int32_t* __restrict__ row_pfx, int npfx) {
    const int h = blockIdx.x, r = blockIdx.y, w = blockIdx.z;
    const int lane = threadIdx.x;
    const int c = lane & 7, tg = lane >> 3;
    const int st = row_stream[r];
    const int ctx = row_ctx[r];
    const int pfx = row_pfx ? row_pfx[r] : -1;
    const bf16_t* lbase = kv.pool + (size_t)layer * kv.layer_stride;
    float q[8], m = -1e30f, l = 0.f, o[8];
    {
        const u32x4 qv = *reinterpret_cast<const u32x4*>(qkv + (size_t)r * ldq + h * 64 + c * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) { q[2 * k] = bf_lo(qv[k]); q[2 * k + 1] = bf_hi(qv[k]); }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0.f;
    const int nt = st >= 0 ? (ctx + 63) >> 6 : 0;
    AttTile t;
    for (int tile = w; tile < nt; tile += ATT_NW) {
        att_load_tile(t, lbase, kv, st, tile, h, lane, pfx, npfx, ctx);
        att_fold_tile(t, q, tile, ctx, tg, m, l, o);
    }
    att_merge_groups(m, l, o);
    if (tg == 0) {
        float* sp = states + ((((size_t)r * kv.H + h) * ATT_NW + w) * 8 + c) * ATT_STATE_FLOATS;
        *reinterpret_cast<f32x4*>(sp) = f32x4{m, l, o[0], o[1]};
        *reinterpret_cast<f32x4*>(sp + 4) = f32x4{o[2], o[3], o[4], o[5]};
        *reinterpret_cast<f32x4*>(sp + 8) = f32x4{o[6], o[7], 0.f, 0.f};
    }
}

#ifndef T3_ATTN_TIMELINE
int launch_attn_decode_split(const bf16_t* qkv, int ldq, float* states, int rows, const int32_t* row_stream,
                             const int32_t* row_ctx, const KvView& kv, int layer, hipStream_t st,
                             const int32_t* row_pfx, int npfx) {
    T3_REQUIRE(kv.hd == 64, "attn_decode_split: head_dim must be 64");
    T3_REQUIRE(rows >= 1 && ldq % 8 == 0 && states, "attn_decode_split: bad arguments");
    T3_REQUIRE(row_pfx == nullptr || (npfx >= 1 && npfx <= T3_PAGE), "attn_decode: the shared prefix must fit one page");
    T3_LAUNCH(attn_decode_split_kernel, dim3(kv.H, rows, ATT_NW), dim3(64), 0, st, qkv, ldq, states, row_stream, row_ctx, kv,
              layer, row_pfx, npfx);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attn_decode(const bf16_t* qkv, int ldq, bf16_t* out, int rows, const int32_t* row_stream,
                       const int32_t* row_ctx, const KvView& kv, int layer, hipStream_t st,
                       const int32_t* row_pfx, int npfx) {
    T3_REQUIRE(kv.hd == 64, "attn_decode: head_dim must be 64");
    T3_REQUIRE(rows >= 1 && ldq % 8 == 0, "attn_decode: bad shape");
    T3_REQUIRE(row_pfx == nullptr || (npfx >= 1 && npfx <= T3_PAGE), "attn_decode: the shared prefix must fit one page");
    T3_LAUNCH((attn_decode_kernel<1, false, false>), dim3(kv.H, rows), dim3(ATT_NW * 64), 0, st, qkv, ldq, out,
              kv.H * kv.hd, rows, row_stream, row_ctx, kv, layer, row_pfx, npfx, static_cast<const bf16_t*>(nullptr),
              static_cast<float*>(nullptr));
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// Attention + output projection: slab[h][row][0..1024) = bf16(attention of head h) x Wq[:, 64h..64h+64)^T, fp32, where Wq is
// Wo in the tile order of gemm.hip (bf16 tiles), or its unscaled int8 values (pair tiles; `int8` set: the fold applies the
// column scales).  rows_per_group: 1 (workgroup = (head, row)) or ATTN_GROUP_ROWS (workgroup = (head, 4 consecutive rows)).
int launch_attn_decode_oproj(const bf16_t* qkv, int ldq, float* slab, int rows, const int32_t* row_stream,
                             const int32_t* row_ctx, const KvView& kv, int layer, const bf16_t* wo, bool int8,
                             int rows_per_group, hipStream_t st, const int32_t* row_pfx, int npfx) {
    T3_REQUIRE(kv.hd == 64 && kv.H == 16, "attn_decode_oproj: built for 16 heads x 64");
    T3_REQUIRE(rows >= 1 && ldq % 8 == 0 && slab && wo, "attn_decode_oproj: bad arguments");
    T3_REQUIRE(rows_per_group == 1 || rows_per_group == 2 || rows_per_group == 4, "attn_decode_oproj: rows per group must be 1, 2 or 4");
    T3_REQUIRE(row_pfx == nullptr || (npfx >= 1 && npfx <= T3_PAGE), "attn_decode: the shared prefix must fit one page");
    const int groups = (rows + rows_per_group - 1) / rows_per_group;
#define T3_ATT_OPROJ(RPW, W8)                                                                                              \
    T3_LAUNCH((attn_decode_kernel<RPW, true, W8>), dim3(kv.H, groups), dim3(ATT_NW * 64), 0, st, qkv, ldq,                \
              static_cast<bf16_t*>(nullptr), 0, rows, row_stream, row_ctx, kv, layer, row_pfx, npfx, wo, slab)
    if (rows_per_group == 1) { if (int8) T3_ATT_OPROJ(1, true); else T3_ATT_OPROJ(1, false); }
    else if (rows_per_group == 2) { if (int8) T3_ATT_OPROJ(2, true); else T3_ATT_OPROJ(2, false); }
    else { if (int8) T3_ATT_OPROJ(4, true); else T3_ATT_OPROJ(4, false); }
#undef T3_ATT_OPROJ
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}
#endif

// ------------------------------------------------------------------------------------
// Prefill: causal flash attention inside packed prompt rows, read from the step's q|k|v buffer
// (row-major [rows, 3*H*64]).  One workgroup = (sequence, 64-query tile, head), 4 waves of 16
// query rows each; 64-key K and V tiles are staged through LDS (K row-major, V transposed).
//
// Both products run on v_mfma_f32_16x16x32_bf16 with the QUERY on the lane:
//   S^T[key][q] = K * Q^T     A = K rows from LDS, B = Q held in registers for the whole kernel;
//                             lane (q = l&15, g = l>>4) gets keys 16t + 4g + r of sub-tile t,
//                             so the softmax max/sum of a query is 16 registers + 2 shuffles.
//   O^T[d][q]   = V^T * P^T   the exponentiated S^T registers are the B operand as they stand
//                             (k-slot (g, j) <-> key 16*t(j>>2) + 4g + (j&3), the same permutation
//                             is applied to the V^T fragment), no LDS round trip for P.
// P is fed as hi + lo bf16 parts (two MFMAs), so the weights keep ~16 significant bits and the
// result matches an fp32-softmax checker to bf16 rounding.  fp32 online softmax across key tiles.
//
// Shared prefix: a sequence with seq_pfx[seq] >= 0 brings rows for positions npfx.. only; keys
// [0, npfx) are read from page seq_pfx[seq] of the KV pool (written by an earlier step, or by
// the QKV projection of this very step when the prefix rows ride along as a sequence of their
// own).  Key tiles stay aligned to absolute positions, and a query's result does not depend on
// which queries share its tile, so the bits equal those of a prompt that carries its own prefix.
// ------------------------------------------------------------------------------------
#define PF_LD 72   // LDS row stride in elements (64 + 8 pad: 144 B keeps 16-B alignment, spreads banks)

// Chunked prompts (POOL): a sequence's rows cover positions seq_first[seq] .. only, the rest of its prompt went through an
// earlier launch.  Every key is then read from the KV pool through the sequence's block-table row (this launch's own keys
// were written there by the QKV projection before it): the same bf16 values in the same tiles, so a prompt's bits do not
// depend on how it is cut.
template <bool POOL>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const bf16_t* __restrict__ qkv, int ldq,
                                                           bf16_t* __restrict__ out, int ldo,
                                                           const int32_t* __restrict__ seq_start,
                                                           const int32_t* __restrict__ tiles, int H,
                                                           const int32_t* __restrict__ seq_pfx, int npfx,
                                                           const bf16_t* __restrict__ pool_layer,
                                                           const int32_t* __restrict__ seq_first,
                                                           const int32_t* __restrict__ seq_stream,
                                                           const int32_t* __restrict__ block_table, int max_pages) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * PF_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[64 * PF_LD];
    const int hd = 64, D = H * hd;
    const int seq = tiles[2 * blockIdx.x];
    const int h = blockIdx.y;
    // positions: the sequence's rows are positions base .. P-1; q0, qrow, key indices are positions
    const int pfx = seq_pfx ? seq_pfx[seq] : -1;
    const int base = POOL ? seq_first[seq] : (pfx >= 0 ? npfx : 0);
    const int32_t* const bt = POOL ? block_table + (size_t)seq_stream[seq] * max_pages : nullptr;
    const int s0 = seq_start[seq] - base, P = seq_start[seq + 1] - s0;     // row of position p = s0 + p
    const int q0 = base + tiles[2 * blockIdx.x + 1];
    const bf16_t* pk = pool_layer + ((size_t)(pfx >= 0 ? pfx : 0) * 2 * H + h) * (T3_PAGE * 64);   // prefix K slab of this head
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int qrow = q0 + w * 16 + c;
    const int lq = qrow < P ? qrow : P - 1;

    bf16x8 qf[2];
    {
        const bf16_t* qp = qkv + (size_t)(s0 + lq) * ldq + h * hd + 8 * g;
        qf[0] = *reinterpret_cast<const bf16x8*>(qp);
        qf[1] = *reinterpret_cast<const bf16x8*>(qp + 32);
    }
    float m = -1e30f, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int last_q = (q0 + 64 < P ? q0 + 64 : P) - 1;           // last query row of the workgroup
    const int nkt = last_q / 64 + 1;
    const int wave_last_q = q0 + w * 16 + 15;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, key = idx >> 3, ch = idx & 7;
            int gk = kt * 64 + key;
            gk = gk < P ? gk : P - 1;
            const bool in_pfx = pfx >= 0 && gk < npfx;
            const bf16_t* kp;
            if constexpr (POOL) {
                kp = in_pfx ? pk + gk * 64 + ch * 8
                            : pool_layer + ((size_t)bt[gk >> 6] * 2 * H + h) * (T3_PAGE * 64) + (gk & 63) * 64 + ch * 8;
            } else {
                kp = in_pfx ? pk + gk * 64 + ch * 8 : qkv + (size_t)(s0 + gk) * ldq + D + h * hd + ch * 8;
            }
            const u32x4 kv = *reinterpret_cast<const u32x4*>(kp);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(kp + ((POOL || in_pfx) ? (size_t)H * (T3_PAGE * 64) : (size_t)D));
            *reinterpret_cast<u32x4*>(&Ks[key * PF_LD + ch * 8]) = kv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(ch * 8 + 2 * e) * PF_LD + key] = (bf16_t)(vv[e] & 0xffffu);
                Vt[(ch * 8 + 2 * e + 1) * PF_LD + key] = (bf16_t)(vv[e] >> 16);
            }
        }
        __syncthreads();
        if (kt * 64 > wave_last_q) continue;                        // wave-uniform: tile is entirely in the future
        f32x4 st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            st[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(16 * t + c) * PF_LD + 32 * s + 8 * g]);
                st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], st[t], 0, 0, 0);
            }
        }
        float mt = -1e30f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 64 + 16 * t + 4 * g + r;
                const float sv = (key <= qrow && key < P) ? st[t][r] * 0.125f : -1e30f;
                st[t][r] = sv;
                mt = fmaxf(mt, sv);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt);
        const float alpha = __expf(m - mn);
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = st[t][r] > -1e29f ? __expf(st[t][r] - mn) : 0.f;
                st[t][r] = pv;
                ls += pv;
            }
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        l = l * alpha + ls;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = st[2 * pr + (j >> 2)][j & 3];
                const __bf16 hi = (__bf16)pv;
                ph[j] = hi;
                pl[j] = (__bf16)(pv - (float)hi);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16_t* vp = &Vt[(16 * dt + c) * PF_LD + 4 * g];
                const u32x2 v0 = *reinterpret_cast<const u32x2*>(vp + 16 * (2 * pr));
                const u32x2 v1 = *reinterpret_cast<const u32x2*>(vp + 16 * (2 * pr + 1));
                const u32x4 vcat = {v0[0], v0[1], v1[0], v1[1]};
                const bf16x8 vf = __builtin_bit_cast(bf16x8, vcat);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, ph, o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl, o[dt], 0, 0, 0);
            }
        }
    }
    if (qrow < P) {
        const float inv = 1.0f / l;
        bf16_t* op = out + (size_t)(s0 + qrow) * ldo + h * hd + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const u32x2 ov = {pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv)};
            *reinterpret_cast<u32x2*>(op + 16 * dt) = ov;
        }
    }
}

int launch_attn_prefill(const bf16_t* qkv, int ldq, bf16_t* out, const int32_t* seq_start_dev,
                        const int32_t* tiles_dev, int ntiles, int H, int hd, hipStream_t st,
                        const int32_t* seq_pfx_dev, int npfx, const KvView* kv, int layer,
                        const int32_t* seq_first_dev, const int32_t* seq_stream_dev) {
    T3_REQUIRE(hd == 64 && ntiles >= 1, "attn_prefill: head_dim must be 64");
    T3_REQUIRE(seq_pfx_dev == nullptr || (kv && kv->H == H && kv->hd == hd && npfx >= 1 && npfx <= T3_PAGE),
               "attn_prefill: a shared prefix needs the KV pool and must fit one page");
    T3_REQUIRE(!seq_first_dev == !seq_stream_dev && (!seq_first_dev || (kv && kv->H == H && kv->hd == hd && kv->block_table)),
               "attn_prefill: chunked prompts need first positions, streams and the KV pool");
    const bf16_t* pool_layer = (seq_pfx_dev || seq_first_dev) ? kv->pool + (size_t)layer * kv->layer_stride : nullptr;
    if (seq_first_dev)
        T3_LAUNCH(attn_prefill_kernel<true>, dim3(ntiles, H), dim3(256), 0, st, qkv, ldq, out, H * hd, seq_start_dev, tiles_dev, H,
                  seq_pfx_dev, npfx, pool_layer, seq_first_dev, seq_stream_dev, kv->block_table, kv->max_pages);
    else
    T3_LAUNCH(attn_prefill_kernel<false>, dim3(ntiles, H), dim3(256), 0, st, qkv, ldq, out, H * hd, seq_start_dev,
                       tiles_dev, H, seq_pfx_dev, npfx, pool_layer, static_cast<const int32_t*>(nullptr),
                       static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), 0);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}
