// Prompt-row GEMM for the T3 backbone on gfx950:  C[M,N] = A[M,K] * W[N,K]^T for the rows of a prefill
// (M = a few hundred).  The weight-streaming kernel of gemm.hip makes every workgroup read all rows
// of A for 32 output columns; at these M that is 250+ MB of L1 fill per launch.  Here a workgroup
// owns a 128 x 64 output tile and walks K in steps of 64 through LDS.
//
//   256 threads = 4 waves stacked along the rows; a wave accumulates 32 rows x 64 columns = 2 x 4 MFMA tiles
//   (v_mfma_f32_16x16x32_bf16, W tile as the A operand so a lane ends up with 4 consecutive columns
//   of one row, as in gemm.hip).  These launches hold only a few GFLOP, so what counts is how many CUs
//   get a tile and how short a tile's k loop is: 64-column tiles give a prompt step of ~500 rows 240
//   (QKV) to 640 (gate/up) workgroups, up to three per CU, where 128 x 128 tiles left half the chip idle.
//   LDS per stage: A 128 rows x 64 k, row-major with a 144-byte row stride (conflict-free 16-byte
//   reads of 16 rows); W 4 n-blocks x 2 k-tiles in the global tile order (a lane's fragment is its
//   own 16 bytes of a 1 KiB tile, read by all four waves).  Two LDS stages, fed from a ring of three
//   register sets: the global loads of the next three k-steps are in flight while the current one is multiplied.
//   Workgroup -> tile: linear id b runs on XCD b % 8; each XCD takes a contiguous run of tiles in
//   (split, column tile, row tile) order, so the row tiles that share a weight tile share an L2.
//
// K is accumulated in k order inside one accumulator (per split), whatever M is: a row's result does
// not depend on the other rows of the launch.  It does differ in rounding from gemm.hip's 8-wave
// K split, which is why the engine sends prompt rows here and decode rows there, always.
//
// int8 weights (GemmArgs::wscale set): the W block of a k-step is 4 pair tiles of 1 KiB (gemm.hip), staged as
// bytes; a lane reads its 16 bytes once per k-step and turns them into the two bf16 fragments in registers; the
// per-column scale multiplies the fp32 sums in the epilogue.
//
// Epilogues: EPI_QKV_ROPE (a tile's 64 columns are one head: RoPE pairs (c, c+32) sit in one lane;
// q|k|v row store + KV-cache scatter), EPI_SILU (a tile takes 32 gate and the matching 32 up
// columns), EPI_SLAB (split-K 8 fp32 partials for o_proj / down_proj, summed by the norm kernel).
#include "t3_common.h"

#include <type_traits>

namespace {

constexpr int PBM = 128, PBN = 64, PBK = 64;
constexpr int PA_STRIDE = PBK + 8;                   // bf16 elements per LDS row of A (144 bytes)
constexpr int PA_ELEMS = PBM * PA_STRIDE;            // 9216
constexpr int PW_ELEMS = 4 * 2 * 512;                // 4 n-blocks x 2 k-tiles x 512 elements
constexpr int PSTAGE = PA_ELEMS + PW_ELEMS;          // 13312 elements = 26624 bytes
constexpr int PD = 3;                                // register sets of the operand prefetch ring (the k loop is written for 3)

__device__ __forceinline__ bf16x8 prompt_dequant8(unsigned lo, unsigned hi) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (__bf16)((float)((lo >> (8 * j)) & 0xffu) - 128.0f);
        r[4 + j] = (__bf16)((float)((hi >> (8 * j)) & 0xffu) - 128.0f);
    }
    return r;
}

template <int V> using IC = std::integral_constant<int, V>;

}  // namespace

template <int EPI, bool W8>
__global__ __launch_bounds__(256) void gemm_prompt_kernel(
        // leading plain arguments travel in preloaded SGPRs (see gemm.hip): the tile map and the fields the first addresses need
        int m_tiles, int n_tiles, int ksplit, int per_xcd, const bf16_t* lead_A, const bf16_t* lead_Wt, int lead_M, int lead_N,
        int lead_K, int lead_lda, int lead_H, GemmArgs a_) {
    GemmArgs a = a_;
    a.A = lead_A; a.Wt = lead_Wt; a.M = lead_M; a.N = lead_N; a.K = lead_K; a.lda = lead_lda; a.kv.H = lead_H;
    extern __shared__ __attribute__((aligned(16))) bf16_t plds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int total = m_tiles * n_tiles * ksplit;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= total) return;
    const int m_tile = t % m_tiles;
    const int rest = t / m_tiles;
    const int n_tile = rest % n_tiles;
    const int z = rest / n_tiles;
    const int m0 = m_tile * PBM;
    const int ktiles = a.K >> 5;
    const int kspan = a.K / ksplit;                   // K range of this workgroup
    const int k0 = z * kspan;
    const int nks = kspan / PBK;

    // the 4 n-blocks (16 weight rows each) of this tile
    int H = 0, region = 0, head = 0;
    if constexpr (EPI == EPI_QKV_ROPE) {
        H = a.kv.H;
        region = n_tile / H;
        head = n_tile - region * H;
    }
    auto nb_of = [&](int s) -> int {
        if constexpr (EPI == EPI_SILU)                // slots 0,1: gate; 2,3: the matching up rows (F/16 further)
            return n_tile * 2 + (s & 1) + (s >> 1) * (a.N >> 5);
        else if constexpr (EPI == EPI_QKV_ROPE)       // a tile = one head: 4 n-blocks of 16 columns
            return region * 4 * H + head * 4 + s;
        else
            return n_tile * 4 + s;
    };

    // global -> register staging: 4 x 16 B of A and 2 (int8: 1) x 16 B of W per thread and k-step
    const bf16_t* ag[4];
    int a_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 32 + (tid >> 3), chunk = tid & 7;
        int r = m0 + row;
        r = r < a.M ? r : a.M - 1;
        ag[i] = a.A + (size_t)r * a.lda + k0 + chunk * 8;
        a_lds[i] = row * PA_STRIDE + chunk * 8;
    }
    constexpr int NWU = W8 ? 1 : 2;                   // 16-byte units of W per thread and k-step
    const bf16_t* wg[NWU];
    int w_lds[NWU];
#pragma unroll
    for (int i = 0; i < NWU; ++i) {
        const int idx = i * 256 + tid;                // 16-byte unit of the stage's W block (512 bf16 / 256 int8 units)
        if constexpr (W8) {
            const int slot = idx >> 6, within = idx & 63;       // one 1 KiB pair tile per n-block
            wg[i] = a.Wt + (((size_t)nb_of(slot) * (ktiles >> 1) + (k0 >> 6)) * 64 + within) * 8;
        } else {
            const int slot = idx >> 7, within = idx & 127;
            wg[i] = a.Wt + ((size_t)nb_of(slot) * ktiles + (k0 >> 5)) * 512 + within * 8;
        }
        w_lds[i] = PA_ELEMS + idx * 8;
    }
    constexpr int W_STEP = W8 ? 512 : 1024;           // bf16_t units the W pointer advances per k-step

    f32x4 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Prefetch ring: the operands of k-steps ks+1 .. ks+PD-1 are in flight in registers while step ks is multiplied.
    // Set s holds step (ks % PD == s); LDS keeps two stages.
    u32x4 ra[PD][4], rw[PD][NWU];
    auto issue = [&](auto S_, int step) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[S][i] = *reinterpret_cast<const u32x4*>(ag[i] + step * PBK);
#pragma unroll
        for (int i = 0; i < NWU; ++i) rw[S][i] = *reinterpret_cast<const u32x4*>(wg[i] + (size_t)step * W_STEP);
    };
    auto stage = [&](auto S_, int step) {
        constexpr int S = decltype(S_)::value;
        bf16_t* dst = plds + (step & 1) * PSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(dst + a_lds[i]) = ra[S][i];
#pragma unroll
        for (int i = 0; i < NWU; ++i) *reinterpret_cast<u32x4*>(dst + w_lds[i]) = rw[S][i];
    };
    const int a_frag = (w * 32 + (lane & 15)) * PA_STRIDE + 8 * (lane >> 4);
    const int w_frag = PA_ELEMS + lane * 8;
    auto multiply = [&](int step) {
        const bf16_t* cur = plds + (step & 1) * PSTAGE;
        u32x4 wq[4];
        if constexpr (W8) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wq[ni] = *reinterpret_cast<const u32x4*>(cur + w_frag + ni * 512);
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            bf16x8 af[2], wf[4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(cur + a_frag + mi * 16 * PA_STRIDE + kt * 32);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if constexpr (W8) wf[ni] = prompt_dequant8(wq[ni][2 * kt], wq[ni][2 * kt + 1]);
                else wf[ni] = *reinterpret_cast<const bf16x8*>(cur + w_frag + ni * 1024 + kt * 512);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
    };
    // One k-step with its register set known at compile time: refill the set that step ks just vacated with step
    // ks + PD, multiply step ks from LDS, move step ks + 1 from its registers into the other LDS stage.  The main
    // loop has no branch between a load and the wait for it (loads past the last step re-read the last step), so
    // the compiler can count the loads still in flight instead of draining them all; the last <= PD steps issue
    // nothing new and run in a tail.
    auto kstep = [&](auto S_, auto Tail_, int ks) {
        constexpr int S = decltype(S_)::value;
        constexpr bool TAIL = decltype(Tail_)::value;
        if constexpr (!TAIL) {
            issue(IC<S>{}, ks + PD < nks ? ks + PD : nks - 1);
            multiply(ks);
            stage(IC<(S + 1) % PD>{}, ks + 1);
        } else {
            multiply(ks);
            if (ks + 1 < nks) stage(IC<(S + 1) % PD>{}, ks + 1);
        }
        __syncthreads();
    };
    using Main = std::false_type;
    using Tail = std::true_type;
    issue(IC<0>{}, 0);
    issue(IC<1>{}, 1 < nks ? 1 : nks - 1);
    issue(IC<2>{}, 2 < nks ? 2 : nks - 1);
    stage(IC<0>{}, 0);
    __syncthreads();
    int ks = 0;
    for (; ks + PD < nks; ks += PD) {
        kstep(IC<0>{}, Main{}, ks);
        kstep(IC<1>{}, Main{}, ks + 1);
        kstep(IC<2>{}, Main{}, ks + 2);
    }
    kstep(IC<0>{}, Tail{}, ks);
    if (ks + 1 < nks) kstep(IC<1>{}, Tail{}, ks + 1);
    if (ks + 2 < nks) kstep(IC<2>{}, Tail{}, ks + 2);

    if constexpr (W8) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.wscale + nb_of(ni) * 16 + 4 * (lane >> 4));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][ni] *= sc;
        }
    }

    // ---- epilogue: lane holds columns 4*(lane>>4)..+3 of n-block `ni`, row (lane & 15) of row tile `mi`
    const int nl = 4 * (lane >> 4);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = m0 + w * 32 + mi * 16 + (lane & 15);
        if (r >= a.M) continue;
        if constexpr (EPI == EPI_SLAB) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                *reinterpret_cast<f32x4*>(a.slab + ((size_t)z * a.M + r) * a.N + nb_of(ni) * 16 + nl) = acc[mi][ni];
        } else if constexpr (EPI == EPI_SILU) {
            const int F = a.N >> 1;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                float hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float g = rbf(acc[mi][ni][i]), u = rbf(acc[mi][ni + 2][i]);
                    const float sg = rbf(g / (1.0f + expf(-g)));
                    hv[i] = sg * u;
                }
                const u32x2 o = {pack_bf2(hv[0], hv[1]), pack_bf2(hv[2], hv[3])};
                *reinterpret_cast<u32x2*>(a.C + (size_t)r * F + (n_tile * 2 + ni) * 16 + nl) = o;
            }
        } else {   // EPI_QKV_ROPE: this tile's 64 columns are head `head` of q, k or v
            const int hd = 64, D = H * hd;
            const int pos = a.pos[r];
            bf16_t* qrow = a.C + (size_t)r * a.ldc + region * D + head * hd;
            bf16_t* kvrow = nullptr;
            if (region >= 1) {
                const int st = a.row_stream[r];
                if (st >= 0) {
                    const int page = a.kv.block_table[(size_t)st * a.kv.max_pages + (pos >> 6)];
                    kvrow = a.kv.pool + (size_t)a.layer * a.kv.layer_stride +
                            ((size_t)page * 2 + (region - 1)) * ((size_t)H * T3_PAGE * hd) +
                            ((size_t)head * T3_PAGE + (pos & 63)) * hd;
                }
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int c0 = ni * 16 + nl, c1 = c0 + 32;
                float x[4], y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { x[i] = rbf(acc[mi][ni][i]); y[i] = rbf(acc[mi][ni + 2][i]); }
                if (region < 2) {
                    const bf16_t* tb = a.rope + (size_t)pos * hd;
                    const u32x2 cs = *reinterpret_cast<const u32x2*>(tb + c0);
                    const u32x2 sn = *reinterpret_cast<const u32x2*>(tb + 32 + c0);
                    const float c[4] = {bf_lo(cs[0]), bf_hi(cs[0]), bf_lo(cs[1]), bf_hi(cs[1])};
                    const float sv[4] = {bf_lo(sn[0]), bf_hi(sn[0]), bf_lo(sn[1]), bf_hi(sn[1])};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xr = x[i] * c[i] - y[i] * sv[i];
                        const float yr = y[i] * c[i] + x[i] * sv[i];
                        x[i] = xr; y[i] = yr;
                    }
                }
                const u32x2 ox = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
                const u32x2 oy = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
                *reinterpret_cast<u32x2*>(qrow + c0) = ox;
                *reinterpret_cast<u32x2*>(qrow + c1) = oy;
                if (kvrow) {
                    *reinterpret_cast<u32x2*>(kvrow + c0) = ox;
                    *reinterpret_cast<u32x2*>(kvrow + c1) = oy;
                }
            }
        }
    }
}

template <int EPI, bool W8>
static int launch_prompt_w(const GemmArgs& a, int n_tiles, int ksplit, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)PSTAGE * sizeof(bf16_t);
    const int m_tiles = (a.M + PBM - 1) / PBM;
    const int total = m_tiles * n_tiles * ksplit;
    const int per_xcd = (total + 7) / 8;
    T3_LAUNCH((gemm_prompt_kernel<EPI, W8>), dim3(per_xcd * 8), dim3(256), lds, st, m_tiles, n_tiles, ksplit, per_xcd, a.A, a.Wt,
              a.M, a.N, a.K, a.lda, a.kv.H, a);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}
template <int EPI>
static int launch_prompt(const GemmArgs& a, int n_tiles, int ksplit, hipStream_t st) {
    return a.wscale ? launch_prompt_w<EPI, true>(a, n_tiles, ksplit, st) : launch_prompt_w<EPI, false>(a, n_tiles, ksplit, st);
}

int launch_gemm_prompt(int epi, const GemmArgs& a, hipStream_t st) {
    T3_REQUIRE(a.M >= 1 && a.lda % 8 == 0, "gemm_prompt: bad rows or lda");
    T3_REQUIRE(a.K % (PBK * GEMM_SLAB_SPLIT) == 0, "gemm_prompt: K must be a multiple of 512");
    switch (epi) {
        case EPI_SILU:
            T3_REQUIRE(a.N % 64 == 0, "gemm_prompt silu: N = 2F with F % 32 == 0");
            return launch_prompt<EPI_SILU>(a, a.N / 64, 1, st);    // a tile = 32 gate + the matching 32 up columns
        case EPI_QKV_ROPE:
            T3_REQUIRE(a.kv.hd == 64 && a.N == 3 * a.kv.H * 64 && a.ldc % 4 == 0,
                       "gemm_prompt qkv: needs head_dim 64, N = 3*H*64");
            return launch_prompt<EPI_QKV_ROPE>(a, a.N / PBN, 1, st);
        case EPI_SLAB:
            T3_REQUIRE(a.slab != nullptr && a.N % PBN == 0, "gemm_prompt slab: N must be a multiple of 64");
            return launch_prompt<EPI_SLAB>(a, a.N / PBN, GEMM_SLAB_SPLIT, st);
    }
    t3_set_error("gemm_prompt: unknown epilogue");
    return 1;
}
