// Split-bf16 forms of the flow-matching decoder's matrix kernels on gfx950 (see cfm_split.h for the arithmetic).
// Reference for what is computed: the diffusers default attention processor inside BasicTransformerBlock
// (src/chatterbox_vllm/models/s3gen/matcha/transformer.py:243-316), softmax(q k^T / 8) v over all frames of an utterance.
#include "cfm_split.h"
#include "../../include/s3cfm_hip.h"

#include <cmath>
#include <cstdlib>
#include <functional>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHeads = 8, kInner = 512;
constexpr float kQScale = 0.125f * 1.44269504088896340736f;     // 1 / sqrt(64) and log2(e): softmax in base 2

// eight values of one lane -> the lane's 16 bytes of each of the three operand planes
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned a, b, c;
        split_pair(x[2 * q], x[2 * q + 1], a, b, c);
        p0[q] = a; p1[q] = b; p2[q] = c;
    }
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
// the six partial products of two split operands: A planes a[0..2], B planes b[0..2]
#define MFMA_SPLIT6(a, b, c)                                                            \
    do {                                                                                \
        c = MFMA_BF16((a)[0], (b)[2], c);                                               \
        c = MFMA_BF16((a)[2], (b)[0], c);                                               \
        c = MFMA_BF16((a)[1], (b)[1], c);                                               \
        c = MFMA_BF16((a)[0], (b)[1], c);                                               \
        c = MFMA_BF16((a)[1], (b)[0], c);                                               \
        c = MFMA_BF16((a)[0], (b)[0], c);                                               \
    } while (0)

// One wave per 1 KiB operand block of each of Q, K and V: workgroup = (frame tile, head, row), wave w = depth step s of Q and
// K and (t, db) = (w / 2, w % 2) of V.
__global__ __launch_bounds__(256) void split_qkv_kernel(const float* __restrict__ qk, size_t qk_bs, const float* __restrict__ vt, size_t vt_bs,
                                                        int vt_ld, int ld, unsigned short* __restrict__ qp, unsigned short* __restrict__ kp,
                                                        unsigned short* __restrict__ vp) {
    const int tile = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const size_t plane = (size_t)(ld / 32) * 2048;
    const size_t bh = ((size_t)b * kHeads + h) * 3;
    const size_t blk = ((size_t)tile * 4 + w) * 512 + (size_t)lane * 8;
    float x[8];
    u32x4 p0, p1, p2;
    const float* q = qk + b * qk_bs + (size_t)(h * 64 + 16 * w + 8 * half) * ld + tile * 32 + col;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = q[(size_t)j * ld] * kQScale;
    split8(x, p0, p1, p2);
    *reinterpret_cast<u32x4*>(qp + (bh + 0) * plane + blk) = p0;
    *reinterpret_cast<u32x4*>(qp + (bh + 1) * plane + blk) = p1;
    *reinterpret_cast<u32x4*>(qp + (bh + 2) * plane + blk) = p2;
    const float* k = q + (size_t)kInner * ld;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = k[(size_t)j * ld];
    split8(x, p0, p1, p2);
    *reinterpret_cast<u32x4*>(kp + (bh + 0) * plane + blk) = p0;
    *reinterpret_cast<u32x4*>(kp + (bh + 1) * plane + blk) = p1;
    *reinterpret_cast<u32x4*>(kp + (bh + 2) * plane + blk) = p2;
    const int t = w >> 1, db = w & 1;
    const float* v = vt + b * vt_bs + (size_t)(tile * 32 + 16 * t + 4 * half) * vt_ld + h * 64 + 32 * db + col;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[(size_t)(8 * (j >> 2) + (j & 3)) * vt_ld];
    split8(x, p0, p1, p2);
    *reinterpret_cast<u32x4*>(vp + (bh + 0) * plane + blk) = p0;
    *reinterpret_cast<u32x4*>(vp + (bh + 1) * plane + blk) = p1;
    *reinterpret_cast<u32x4*>(vp + (bh + 2) * plane + blk) = p2;
}

// both halves of a wave get the maximum (the sum) of a value over the two lanes n and n + 32: one lane-swap instruction
__device__ __forceinline__ float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Self-attention of one transformer block on split operands.  The work split is attn_kernel's (cfm_decoder.hip): one
// workgroup per (QT tiles of 32 queries, head, row), its four waves take every fourth 32-key tile with an online softmax
// each and merge through LDS in wave order; scores transposed (S^T[key][query]) so that a lane owns one query column and
// the probabilities sit where the second product wants its B operand.  Per key tile and query tile: 24 + 24 matrix
// instructions of 32 cycles against 32 + 32 of 64, so the vector work of a tile (softmax, splitting the probabilities:
// about 190 instructions per query tile) is as long as its matrix work and must run under it: the loop is software-pipelined,
// each half of a tile's softmax sits in one basic block with matrix instructions that do not depend on it (the scores of the
// wave's NEXT key tile, the second product of the CURRENT one), without branches, so that the compiler interleaves the two
// streams.  The key mask is only ever needed on a wave's last tile, which is peeled off the loop.
template <int QT>
__global__ __launch_bounds__(256, QT == 1 ? 2 : 1) void attn_split_kernel(const unsigned short* __restrict__ lead_qp, const unsigned short* __restrict__ lead_kp,
                                                         const unsigned short* __restrict__ lead_vp, int lead_ld, int lead_L, int lead_rows,
                                                         SplitAttnArgs a_) {
    __shared__ float so[4 * 32 * 64];
    __shared__ float sm[4][64], sl[4][64];
    const int ld = lead_ld, L = lead_L, rows = lead_rows;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, col = lane & 31;
    // workgroup i runs on XCD i % 8: each XCD gets its own share of the (row, head) pairs, whose K and V then stay in its L2
    const int npairs = rows * kHeads, p = blockIdx.x % npairs, qt = blockIdx.x / npairs;
    const int pair = (p & 7) * rows + (p >> 3);
    const int b = pair >> 3, h = pair & 7;
    const int Lk = a_.Lk[b >> 1];
    float* const op = a_.o + b * a_.o_bs + (size_t)(h * 64) * ld;
    if (qt * QT * 32 >= Lk) {
        // queries that are all padding: zeros keep what follows finite without the work
        for (int i = tid; i < 64 * QT * 32; i += 256) {
            const int qn = qt * QT * 32 + i % (QT * 32);
            if (qn < L) op[(size_t)(i / (QT * 32)) * ld + qn] = 0.f;
        }
        return;
    }
    const int ltiles = ld / 32;
    const size_t plane = (size_t)ltiles * 2048;
    const size_t bh = ((size_t)b * kHeads + h) * 3;
    const u32x4* const qb = reinterpret_cast<const u32x4*>(lead_qp + bh * plane) + lane;
    const u32x4* const kb = reinterpret_cast<const u32x4*>(lead_kp + bh * plane) + lane;
    const u32x4* const vb = reinterpret_cast<const u32x4*>(lead_vp + bh * plane) + lane;
    const size_t pl16 = plane / 8;                             // plane stride in 16-byte units
    u32x4 Q[QT][4][3];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int tile = qt * QT + u < ltiles ? qt * QT + u : ltiles - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) Q[u][s][pl] = qb[pl * pl16 + (size_t)(tile * 4 + s) * 64];
    }
    f32x16 o0[QT], o1[QT], sc[QT], sn[QT];
    float m[QT], l[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[u][i] = 0.f; o1[u][i] = 0.f; sc[u][i] = 0.f; sn[u][i] = 0.f; }
        m[u] = -INFINITY;
        l[u] = 0.f;
    }
    const int ntiles = (Lk + 31) / 32;
    u32x4 K[4][3], V[4][3], P[QT][2][3];
    auto load_k = [&](int tt) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) K[s][pl] = kb[pl * pl16 + (size_t)(tt * 4 + s) * 64];
    };
    auto load_v = [&](int tt) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) V[s][pl] = vb[pl * pl16 + (size_t)(tt * 4 + s) * 64];
    };
    // scores of the tile whose K is in registers: accumulator r of a lane is key 8 (r / 4) + 4 (lane / 32) + r % 4 of the
    // tile, query lane % 32; in units of ln 2 (the factor sits in Q)
    auto scores = [&](f32x16 (&dst)[QT]) {
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[u][i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < QT; ++u) MFMA_SPLIT6(K[s], Q[u][s], dst[u]);
    };
    // The online softmax of a key tile is cut in two, so that each half can sit among the matrix instructions of one of the
    // two products (no branches in either):
    //   stats:  tile maximum, the softmax reference to use, probabilities (in place of the scores), their sum;
    //   finish: running sum, the probabilities as operand planes (accumulators 8 t2 .. 8 t2 + 7 of a lane are the eight
    //           keys of depth step t2 in the order V was written in).
    // The softmax reference m of a query follows its running maximum in jumps: it moves only when a tile's maximum lies more
    // than 16 units of ln 2 above it (probabilities then reach 2^16 at most, far from any fp32 limit; relative precision does
    // not depend on the reference).  After a wave's first tile that is rare, so the rescale of the output accumulators sits
    // under a wave-uniform branch that is almost never taken, ahead of the block, and the accumulators otherwise stay in the
    // registers the matrix instructions keep them in (with an unconditional rescale they were moved to the vector registers
    // and back every tile: 64 multiplies and 128 moves per tile among 590 vector instructions).
    float m_new[QT], psum[QT];
    auto stats = [&](int u, f32x16 (&s)[QT]) {
        float mx = s[u][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
        mx = half_max(mx);                                     // finite: every tile holds a key below Lk
        m_new[u] = mx > m[u] + 16.0f ? mx : m[u];
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[u][r] = __builtin_amdgcn_exp2f(s[u][r] - m_new[u]); ps += s[u][r]; }
        psum[u] = half_sum(ps);
    };
    auto move_reference = [&]() {
        bool moved = false;
#pragma unroll
        for (int u = 0; u < QT; ++u) moved |= m_new[u] != m[u];
        if (__any(moved)) {
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                const float alpha = __builtin_amdgcn_exp2f(m[u] - m_new[u]);     // 1 where the reference stays, 0 on the first tile
                l[u] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[u][i] *= alpha; o1[u][i] *= alpha; }
                m[u] = m_new[u];
            }
        }
    };
    auto finish = [&](int u, const f32x16 (&s)[QT]) {
        l[u] += psum[u];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned a, bb, c;
                split_pair(s[u][8 * t2 + 2 * q], s[u][8 * t2 + 2 * q + 1], a, bb, c);
                P[u][t2][0][q] = a; P[u][t2][1][q] = bb; P[u][t2][2][q] = c;
            }
    };
    auto second_product = [&]() {
        // consecutive matrix instructions go to different accumulators
        constexpr int ta[6] = {0, 2, 1, 0, 1, 0}, tb[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    o0[u] = MFMA_BF16(V[2 * t2][ta[k]], P[u][t2][tb[k]], o0[u]);
                    o1[u] = MFMA_BF16(V[2 * t2 + 1][ta[k]], P[u][t2][tb[k]], o1[u]);
                }
    };
    int t = w;
    auto mask_tail = [&](f32x16 (&s)[QT], int tile) {          // only a wave's last tile can hold keys past Lk
        const int key0 = tile * 32;
        if (key0 + 32 > Lk) {
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + 8 * (r >> 2) + 4 * half + (r & 3) >= Lk) s[u][r] = -INFINITY;
        }
    };
    if (t < ntiles) {
        load_k(t);
        load_v(t);
        scores(sc);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 4 >= ntiles) {                                 // the wave's only tile
            mask_tail(sc, t);
#pragma unroll
            for (int u = 0; u < QT; ++u) stats(u, sc);
            move_reference();
#pragma unroll
            for (int u = 0; u < QT; ++u) finish(u, sc);
            second_product();
        } else {
            load_k(t + 4);
#pragma unroll
            for (int u = 0; u < QT; ++u) stats(u, sc);
            // Steady state, two basic blocks per key tile, each with half of the softmax among one product's matrix
            // instructions: [scores of tile t + 4 | finish of tile t] and [second product of tile t | stats of tile t + 4].
            for (; t + 8 < ntiles; t += 4) {
                move_reference();
                __builtin_amdgcn_sched_barrier(0);
                scores(sn);
#pragma unroll
                for (int u = 0; u < QT; ++u) finish(u, sc);
                __builtin_amdgcn_sched_barrier(0);
                load_k(t + 8);
                __builtin_amdgcn_sched_barrier(0);
                second_product();
#pragma unroll
                for (int u = 0; u < QT; ++u) stats(u, sn);
                __builtin_amdgcn_sched_barrier(0);
                load_v(t + 4);
#pragma unroll
                for (int u = 0; u < QT; ++u) sc[u] = sn[u];
            }
            // tile t + 4 is the wave's last one
            move_reference();
            __builtin_amdgcn_sched_barrier(0);
            scores(sn);
#pragma unroll
            for (int u = 0; u < QT; ++u) finish(u, sc);
            __builtin_amdgcn_sched_barrier(0);
            second_product();
            __builtin_amdgcn_sched_barrier(0);
            load_v(t + 4);
            mask_tail(sn, t + 4);
#pragma unroll
            for (int u = 0; u < QT; ++u) stats(u, sn);
            move_reference();
#pragma unroll
            for (int u = 0; u < QT; ++u) finish(u, sn);
            second_product();
        }
    }
    // merge the four waves' partial results, one query tile at a time through the same 32 KiB
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        if (u > 0) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            so[(w * 32 + r) * 64 + lane] = o0[u][r];
            so[(w * 32 + 16 + r) * 64 + lane] = o1[u][r];
        }
        sm[w][lane] = m[u];
        sl[w][lane] = l[u];
        __syncthreads();
        const float ms = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
        float f[4], lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = __builtin_amdgcn_exp2f(sm[i][lane] - ms); lsum += f[i] * sl[i][lane]; }   // a wave without tiles: 2^-inf = 0
        const float inv = 1.0f / lsum;
        const int qn = (qt * QT + u) * 32 + col;
        if (qn < L) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = w * 8 + i;                       // this wave finishes a quarter of the 64 output channels
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) v += so[(j * 32 + r) * 64 + lane] * f[j];
                const int d = (r >> 4) * 32 + 8 * ((r & 15) >> 2) + 4 * half + (r & 3);
                op[(size_t)d * ld + qn] = v * inv;
            }
        }
    }
}

}  // namespace

int launch_split_qkv(const float* qk, size_t qk_bs, const float* vt, size_t vt_bs, int vt_ld, int ld, int rows, unsigned short* qp,
                     unsigned short* kp, unsigned short* vp, hipStream_t st) {
    S3_REQUIRE(qk && vt && qp && kp && vp && ld % 32 == 0 && rows >= 1, "split of q, k, v: bad arguments");
    hipLaunchKernelGGL(split_qkv_kernel, dim3(ld / 32, kHeads, rows), dim3(256), 0, st, qk, qk_bs, vt, vt_bs, vt_ld, ld, qp, kp, vp);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_attn_split(const SplitAttnArgs& a, hipStream_t st) {
    S3_REQUIRE(a.qp && a.kp && a.vp && a.o && a.ld % 32 == 0 && a.L >= 1 && a.L <= a.ld && a.rows >= 1 && a.rows <= 2 * kSplitMaxUtt,
               "split attention: bad arguments");
    static const int force = getenv("S3CFM_ATTN_QT") ? atoi(getenv("S3CFM_ATTN_QT")) : 0;
    // query tiles per wave: the count whose workgroups pack best into the 256 CUs (launch_attn of cfm_decoder.hip)
    const int tiles = (a.L + 31) / 32;
    int best = 1;
    double best_cost = 1e30;
    for (int qt = 1; qt <= 2; ++qt) {
        const long long wgs = (long long)((tiles + qt - 1) / qt) * a.rows * kHeads;
        const double per_tile = qt == 1 ? 1.1 : 1.0;
        const double cost = (double)((wgs + 255) / 256) * qt * per_tile;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = qt; }
    }
    if (force >= 1 && force <= 2) best = force;
    const dim3 grid((tiles + best - 1) / best * a.rows * kHeads);
    if (best == 1) hipLaunchKernelGGL(attn_split_kernel<1>, grid, dim3(256), 0, st, a.qp, a.kp, a.vp, a.ld, a.L, a.rows, a);
    else hipLaunchKernelGGL(attn_split_kernel<2>, grid, dim3(256), 0, st, a.qp, a.kp, a.vp, a.ld, a.L, a.rows, a);
    S3_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- single operations, for tests and timing --------------------------------------------------------------------------

namespace {
struct DevBytes {
    void* p = nullptr;
    ~DevBytes() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { S3_HIP_CHECK(hipMalloc(&p, n)); S3_HIP_CHECK(hipMemset(p, 0, n)); return 0; }
};
int time_launches(int iters, hipStream_t st, float* avg_us, const std::function<int()>& launch) {
    hipEvent_t e0, e1;
    S3_HIP_CHECK(hipEventCreate(&e0));
    S3_HIP_CHECK(hipEventCreate(&e1));
    int rc = launch();                                           // warm-up, and the result the caller reads
    S3_HIP_CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters && !rc; ++i) rc = launch();
    S3_HIP_CHECK(hipEventRecord(e1, st));
    S3_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    S3_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (avg_us) *avg_us = iters > 0 ? ms * 1e3f / iters : 0.f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}
}  // namespace

extern "C" int s3cfm_op_attention_split(const float* qk_dev, const float* vt_dev, int32_t n_utt, const int32_t* keys, int32_t L, int32_t ld,
                                        float* o_dev, int32_t iters, float* attn_avg_us, float* split_avg_us, void* stream) {
    S3_REQUIRE(qk_dev && vt_dev && o_dev && L >= 1 && ld >= (L + 31) / 32 * 32 && ld % 32 == 0 && iters >= 0,
               "bad arguments (ld must be whole 32-frame tiles)");
    S3_REQUIRE(n_utt >= 1 && n_utt <= kSplitMaxUtt, "bad number of utterances");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = 2 * n_utt;
    const size_t bytes = (size_t)rows * 3 * kInner * ld * sizeof(unsigned short);
    DevBytes qp, kp, vp;
    if (qp.alloc(bytes) || kp.alloc(bytes) || vp.alloc(bytes)) return 1;
    SplitAttnArgs at{static_cast<unsigned short*>(qp.p), static_cast<unsigned short*>(kp.p), static_cast<unsigned short*>(vp.p), o_dev,
                     (size_t)kInner * ld, ld, L, rows, {}};
    for (int u = 0; u < n_utt; ++u) {
        at.Lk[u] = keys ? keys[u] : L;
        S3_REQUIRE(at.Lk[u] >= 1 && at.Lk[u] <= L, "an utterance's keys must lie in [1, L]");
    }
    const int vt_ld = kInner + 64;
    int rc = time_launches(iters, st, split_avg_us, [&]() {
        return launch_split_qkv(qk_dev, (size_t)2 * kInner * ld, vt_dev, (size_t)ld * vt_ld, vt_ld, ld, rows, at.qp ? static_cast<unsigned short*>(qp.p) : nullptr,
                                static_cast<unsigned short*>(kp.p), static_cast<unsigned short*>(vp.p), st);
    });
    if (!rc) rc = time_launches(iters, st, attn_avg_us, [&]() { return launch_attn_split(at, st); });
    S3_HIP_CHECK(hipStreamSynchronize(st));
    return rc;
}
