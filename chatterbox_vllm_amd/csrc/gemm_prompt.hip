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

#include <cstdlib>
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

// Epilogue of one 16-row x 64-column accumulator tile: lane holds columns 4*(lane>>4)..+3 of n-block `ni` (acc[ni]) of row r.
template <int EPI, class NbOf>
__device__ __forceinline__ void prompt_store_tile(const GemmArgs& a, const f32x4 (&acc)[4], int r, int z, int n_tile, int region,
                                                  int head, int H, NbOf nb_of, int lane) {
    const int nl = 4 * (lane >> 4);
    if (r >= a.M) return;
    if constexpr (EPI == EPI_SLAB) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            *reinterpret_cast<f32x4*>(a.slab + ((size_t)z * a.M + r) * a.N + nb_of(ni) * 16 + nl) = acc[ni];
    } else if constexpr (EPI == EPI_SILU) {
        const int F = a.N >> 1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = rbf(acc[ni][i]), u = rbf(acc[ni + 2][i]);
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
            for (int i = 0; i < 4; ++i) { x[i] = rbf(acc[ni][i]); y[i] = rbf(acc[ni + 2][i]); }
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
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
        prompt_store_tile<EPI>(a, acc[mi], m0 + w * 32 + mi * 16 + (lane & 15), z, n_tile, region, head, H, nb_of, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Row-walking form: every weight byte is fetched ONCE per launch (per 512 rows).
//
// gemm_prompt_kernel gives every 128-row tile of a 64-column weight tile a workgroup of its own; the five or six of them
// ask for the same weight lines at the same time, the L2 does not merge those misses, and the launch fetches its weights
// four times over (counters: gate_up 66.6 MB for 16.8 MB of weights).  Here ONE workgroup owns the 64 columns (and its K
// range) and walks ALL rows with them: 4 waves x 128 rows, a wave's 8 x 4 accumulator tiles (128 VGPRs) stay in registers
// for the whole K loop.  K advances in steps of 32.  The W block of a step (4 n-blocks x 1 KiB in the global tile order)
// goes through a two-stage LDS ring and is read by all four waves.  The A rows belong to one wave each, so staging them in
// LDS would only add a write and a read per byte (that version was LDS-bound at 1.1 us per step): a wave loads its 8 MFMA
// B fragments of a step straight from memory (lane = row, 16 bytes = 8 k) three steps ahead, in registers.  The
// accumulation order of an output is k order inside one accumulator, as in gemm_prompt_kernel: the same bits.
// Fewer, longer workgroups (gate_up: 128 instead of 640) leave CUs to the decode stream that runs beside a prefill job;
// the launch itself takes 1.5 to 3 times as long as gemm_prompt_kernel's at a few hundred rows (equal at 4096), so the
// engine does not use this form by default (see g_prompt_walk below).
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int WBM = 512;                             // rows per workgroup: 4 waves x 8 row tiles of 16
constexpr int WW_ELEMS = 4 * 512;                    // 4 n-blocks x one 16 x 32 tile = 4 KiB
constexpr int WAD = 3;                               // k-steps of A fragments in flight per wave
}  // namespace

template <int EPI, bool W8>
__global__ __launch_bounds__(256, 1) void gemm_prompt_walk_kernel(
        int m_groups, int n_tiles, int ksplit, const bf16_t* lead_A, const bf16_t* lead_Wt, int lead_M, int lead_N,
        int lead_K, int lead_lda, int lead_H, GemmArgs a_) {
    GemmArgs a = a_;
    a.A = lead_A; a.Wt = lead_Wt; a.M = lead_M; a.N = lead_N; a.K = lead_K; a.lda = lead_lda; a.kv.H = lead_H;
    __shared__ __attribute__((aligned(16))) bf16_t wlds[2][WW_ELEMS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int t = blockIdx.x;
    const int m_group = t % m_groups;
    const int rest = t / m_groups;
    const int n_tile = rest % n_tiles;
    const int z = rest / n_tiles;
    const int g0 = m_group * WBM;
    const int ktiles = a.K >> 5;
    const int kspan = a.K / ksplit;
    const int k0 = z * kspan;
    const int nks = kspan >> 5;                       // k-steps of 32; a multiple of 3 * 4 is not required, see the loop

    int H = 0, region = 0, head = 0;
    if constexpr (EPI == EPI_QKV_ROPE) {
        H = a.kv.H;
        region = n_tile / H;
        head = n_tile - region * H;
    }
    auto nb_of = [&](int s) -> int {
        if constexpr (EPI == EPI_SILU) return n_tile * 2 + (s & 1) + (s >> 1) * (a.N >> 5);
        else if constexpr (EPI == EPI_QKV_ROPE) return region * 4 * H + head * 4 + s;
        else return n_tile * 4 + s;
    };

    // A fragment of row tile mi: lane (row = lane & 15, k group = lane >> 4) holds 8 consecutive k of its row
    const bf16_t* ag[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        int r = g0 + w * 128 + mi * 16 + (lane & 15);
        r = r < a.M ? r : a.M - 1;
        ag[mi] = a.A + (size_t)r * a.lda + k0 + 8 * (lane >> 4);
    }
    // W: one 16-byte unit per thread and k-step (bf16), or per pair of k-steps (int8 pair tiles)
    const int w_slot = tid >> 6;
    const bf16_t* wg = W8 ? a.Wt + (((size_t)nb_of(w_slot) * (ktiles >> 1) + (k0 >> 6)) * 64 + lane) * 8
                          : a.Wt + ((size_t)nb_of(w_slot) * ktiles + (k0 >> 5)) * 512 + lane * 8;

    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[WAD][8], rw[4];
    auto issue_a = [&](auto S_, int step) {
        constexpr int S = decltype(S_)::value;
        step = step < nks ? step : nks - 1;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) ra[S][mi] = *reinterpret_cast<const u32x4*>(ag[mi] + step * 32);
    };
    auto issue_w = [&](auto S_, int step) {
        constexpr int S = decltype(S_)::value;
        step = step < nks ? step : nks - 1;
        if constexpr (W8) rw[S] = *reinterpret_cast<const u32x4*>(wg + (size_t)(step >> 1) * 512);
        else rw[S] = *reinterpret_cast<const u32x4*>(wg + (size_t)step * 512);
    };
    auto stage_w = [&](auto R_, int step) {
        constexpr int R = decltype(R_)::value;
        *reinterpret_cast<u32x4*>(&wlds[step & 1][tid * 8]) = rw[R];
    };
    auto multiply = [&](auto S_, int step) {
        constexpr int S = decltype(S_)::value;
        bf16x8 wf[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(&wlds[step & 1][ni * 512 + lane * 8]);
            if constexpr (W8) wf[ni] = (step & 1) ? prompt_dequant8(q[2], q[3]) : prompt_dequant8(q[0], q[1]);
            else wf[ni] = __builtin_bit_cast(bf16x8, q);
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, ra[S][mi]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af, acc[mi][ni], 0, 0, 0);
        }
    };
    // One k-step with static register indices: the W block of step s + 1 goes to the other LDS stage, step s is multiplied,
    // the A set and the W slot that step s vacated are refilled (steps s + WAD and s + 4).  12 steps per trip of the
    // unrolled loop (lcm of the two ring lengths); steps past the end re-read the last step and multiply nothing.
    auto kstep = [&](auto SA_, auto SW_, int s) {
        constexpr int SA = decltype(SA_)::value, SW = decltype(SW_)::value;
        if (s >= nks) return;
        if (s + 1 < nks) stage_w(IC<(SW + 1) % 4>{}, s + 1);
        multiply(IC<SA>{}, s);
        issue_a(IC<SA>{}, s + WAD);
        issue_w(IC<SW>{}, s + 4);
        __syncthreads();
    };
    issue_w(IC<0>{}, 0); issue_w(IC<1>{}, 1); issue_w(IC<2>{}, 2); issue_w(IC<3>{}, 3);
    issue_a(IC<0>{}, 0); issue_a(IC<1>{}, 1); issue_a(IC<2>{}, 2);
    stage_w(IC<0>{}, 0);
    __syncthreads();
    for (int s = 0; s < nks; s += 12) {
        kstep(IC<0>{}, IC<0>{}, s);      kstep(IC<1>{}, IC<1>{}, s + 1);  kstep(IC<2>{}, IC<2>{}, s + 2);
        kstep(IC<0>{}, IC<3>{}, s + 3);  kstep(IC<1>{}, IC<0>{}, s + 4);  kstep(IC<2>{}, IC<1>{}, s + 5);
        kstep(IC<0>{}, IC<2>{}, s + 6);  kstep(IC<1>{}, IC<3>{}, s + 7);  kstep(IC<2>{}, IC<0>{}, s + 8);
        kstep(IC<0>{}, IC<1>{}, s + 9);  kstep(IC<1>{}, IC<2>{}, s + 10); kstep(IC<2>{}, IC<3>{}, s + 11);
    }

    if constexpr (W8) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.wscale + nb_of(ni) * 16 + 4 * (lane >> 4));
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[mi][ni] *= sc;
        }
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
        prompt_store_tile<EPI>(a, acc[mi], g0 + w * 128 + mi * 16 + (lane & 15), z, n_tile, region, head, H, nb_of, lane);
}

// Which form prompt-row launches take: 0 = one workgroup per 128-row tile (default), 1 = the row-walking kernel.  Both give
// the same bits; the switch exists for their A/B (T3_PROMPT_WALK=1 in the environment, or t3_op_set_prompt_gemm_form).
// Measured (1 x MI355X): the row-walking form fetches its weights once instead of four times, but a workgroup's K loop is
// bound by what one CU's vector memory path pulls (about 32 KB of A rows per microsecond), and with one workgroup per weight
// tile only 48 to 128 CUs pull: gate_up at 612 rows 39.8 us against 24.0, the benchmark job 19.3k against 19.6k tokens/s.
static int g_prompt_walk = [] { const char* v = getenv("T3_PROMPT_WALK"); return (v && v[0] == '1') ? 1 : 0; }();
void set_prompt_gemm_form(int walk) { g_prompt_walk = walk ? 1 : 0; }
static bool prompt_walk_enabled() { return g_prompt_walk != 0; }

template <int EPI, bool W8>
static int launch_prompt_w(const GemmArgs& a, int n_tiles, int ksplit, hipStream_t st) {
    if (prompt_walk_enabled() && (a.K / ksplit) % 64 == 0) {
        const int m_groups = (a.M + WBM - 1) / WBM;
        T3_LAUNCH((gemm_prompt_walk_kernel<EPI, W8>), dim3(m_groups * n_tiles * ksplit), dim3(256), 0, st, m_groups, n_tiles,
                  ksplit, a.A, a.Wt, a.M, a.N, a.K, a.lda, a.kv.H, a);
        T3_HIP_CHECK(hipGetLastError());
        return 0;
    }
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
