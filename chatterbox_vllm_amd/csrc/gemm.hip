// Weight-streaming GEMM for the T3 backbone on gfx950:  C[M,N] = A[M,K] * W[N,K]^T.
//
// Decode steps have few rows (M = 1 .. 2*max_num_seqs), so every projection is bound by
// streaming W from HBM exactly once.  W is therefore stored pre-tiled in the order one
// MFMA operand fragment is consumed: tile (nb, kb) = rows nb*16..+16, cols kb*32..+32 of
// W, 1 KiB, element (lane*8 + j) = W[nb*16 + (lane&15)][kb*32 + 8*(lane>>4) + j].  One
// global_load_dwordx4 per lane fetches a whole tile as 1 KiB of contiguous memory and the
// registers are the v_mfma_f32_16x16x32_bf16 operand as loaded - no LDS, no shuffles.
//
// A workgroup owns NB n-blocks (16 output columns each) over the whole K: its KW waves each
// take KT consecutive k-tiles (KW*KT*32 == K), accumulate NB x MT 16x16 tiles (MT = row tiles
// of 16) and the partial tiles are summed through LDS in wave order 0..KW-1.  That order, and
// the k order inside a wave, do not depend on M or on which rows share the launch, so a row's
// result is bit-identical for every batch composition.  Prompt rows (M > 128) run the same
// kernel over row groups of 128 (blockIdx.y); W then comes from L2 / Infinity Cache.
//
// The MFMA is issued as D = Wtile(16 n x 32 k) * Atile^T(32 k x 16 rows): lane l ends up with
// 4 consecutive output columns n = 4*(l>>4)+0..3 of row (l&15), i.e. one 8-byte bf16 store.
//
// Epilogues (one rounding to bf16 after the fp32 accumulation, then):
//   EPI_STORE     C = bf16(acc)
//   EPI_RESID     C = bf16(bf16(acc) + C)                        (o_proj / down_proj + residual)
//   EPI_SILU      h = bf16(bf16(silu(g)) * u), g/u = n-blocks b and b + F/16   (gate_up_proj)
//   EPI_QKV_ROPE  half-split RoPE on q,k (n-blocks j and j+32 columns of one head), store the
//                 rotated q|k|v row and scatter k, v into the paged KV cache.
//
// int8 weights (W8 = true, GemmArgs::wscale != nullptr): W8A16 with one fp32 scale per output
// column, W[n][k] = wscale[n] * (u8[n][k] - 128).  Two k-tiles share a 1 KiB tile (tile (nb, kp):
// byte lane*16 + t*8 + j = u8[nb*16 + (lane&15)][kp*64 + t*32 + 8*(lane>>4) + j]), so a lane's
// 16-byte load still is 1 KiB of contiguous memory per wave; the bytes become exact bf16 integers
// in registers, the MFMA runs as before and the scale multiplies the fp32 sum before the one rounding.
#include "t3_common.h"

// 8 offset-binary bytes (two dwords) -> 8 bf16 values u - 128 (exact: |u - 128| <= 128)
__device__ __forceinline__ bf16x8 dequant8(unsigned lo, unsigned hi) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[j] = (__bf16)((float)((lo >> (8 * j)) & 0xffu) - 128.0f);
        r[4 + j] = (__bf16)((float)((hi >> (8 * j)) & 0xffu) - 128.0f);
    }
    return r;
}

// FN >= 0: residual-add + RMSNorm in the prologue (launches of at most GEMM_FUSED_NORM_ROWS rows, one row tile): wave pair
// (2r, 2r+1) does for row r what slab_add_rmsnorm_kernel<FN> does with its 128 threads, the normalised row tile goes to LDS
// in activation tile order and the MFMA operands come from there.  FN is the number of split-K slabs to fold in.
// AM: the activation rows are attention outputs that this kernel merges itself from the per-wave softmax states of the
// few-row split attention (EPI_SLAB, K = 1024, one row tile, KT = 2: the wave's two k-tiles are one head).
template <int NB, int MT, int KW, int KT, int EPI, bool W8, int FN = -1, bool AM = false>
__global__ __launch_bounds__(KW * 64) void gemm_stream_kernel(
        // The leading plain arguments reach the waves in SGPRs set up at wave launch (the library is built with kernel-argument
        // preloading; a struct argument does not qualify, its fields are fetched by scalar loads the first addresses would wait
        // for): the fields of GemmArgs that those addresses are made of, passed a second time ahead of the struct.
        const bf16_t* lead_A, const bf16_t* lead_Wt, const int32_t* lead_pos, const int32_t* lead_row_stream, int lead_M, int lead_N,
        int lead_K, int lead_lda, int lead_H, int lead_flags, GemmArgs a_) {
    GemmArgs a = a_;
    a.A = lead_A; a.Wt = lead_Wt; a.pos = lead_pos; a.row_stream = lead_row_stream; a.M = lead_M; a.N = lead_N; a.K = lead_K;
    a.lda = lead_lda; a.kv.H = lead_H; a.a_tiled = (lead_flags & 1) != 0; a.c_tiled = (lead_flags & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) float red_raw[];
    f32x4* red = reinterpret_cast<f32x4*>(red_raw);
    __shared__ __attribute__((aligned(16))) bf16_t fn_tile[FN >= 0 ? 16 * 1024 : 8];   // one row tile, all of K
    __shared__ float fn_part[GEMM_FUSED_NORM_ROWS][2];
    constexpr int NT = NB * MT;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int m0 = blockIdx.y * (MT * 16);
    const int ktiles = a.K >> 5;

    int nbi[NB];
    int H = 0, region = 0, head = 0, half = 0;
    if constexpr (EPI == EPI_SILU) {
        nbi[0] = b;
        nbi[1] = b + (a.N >> 5);                  // up rows start at F = N/2 -> F/16 n-blocks
    } else if constexpr (EPI == EPI_QKV_ROPE) {
        H = a.kv.H;
        region = b / (2 * H);
        const int i = b - region * 2 * H;
        head = i >> 1;
        half = i & 1;
        if (region < 2) { nbi[0] = region * 4 * H + head * 4 + half; nbi[1] = nbi[0] + 2; }
        else            { nbi[0] = 2 * 4 * H + head * 4 + half * 2;  nbi[1] = nbi[0] + 1; }
    } else if constexpr (EPI == EPI_SLAB) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) nbi[nb] = b * NB + nb;
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) nbi[nb] = b * NB + nb;
    }

    f32x4 acc[NB][MT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nb][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    T3_STAMP(0);
    const int kt0 = (blockIdx.z * KW + w) * KT;       // blockIdx.z > 0 only for split-K slabs
    // activation row addresses first: the row gather (head GEMM only) is a dependent load, and a
    // wait for it placed after the weight loads would also wait for the whole HBM stream
    const bf16_t* arow[MT];
    const int a_step = a.a_tiled ? 512 : 32;         // elements from one k-tile's fragment to the next
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if constexpr (AM) { arow[mt] = nullptr; continue; }
        if (a.a_tiled) {                              // wave-uniform: activation tile order, 1 KiB per fragment
            int tile = (m0 >> 4) + mt;
            const int last = (a.M - 1) >> 4;
            tile = tile < last ? tile : last;
            arow[mt] = a.A + (((size_t)tile * ktiles + kt0) * 64 + lane) * 8;
        } else {
            int r = m0 + mt * 16 + (lane & 15);
            r = r < a.M ? r : a.M - 1;
            if constexpr (EPI == EPI_STORE_GATHER) r = a.a_rows[r];
            arow[mt] = a.A + (size_t)r * a.lda + kt0 * 32 + 8 * (lane >> 4);
        }
    }
    // QKV epilogue operands (position, rope row, KV page) hang off two levels of dependent loads;
    // fetched here, ahead of and underneath the weight stream, they cost nothing at the end.
    int pre_pos = 0, pre_st = -1;
    if constexpr (EPI == EPI_QKV_ROPE) {
        static_assert(MT <= KW, "each wave finishes at most one row tile");
        if (w < MT) {
            int r = m0 + w * 16 + (lane & 15);
            r = r < a.M ? r : a.M - 1;
            pre_pos = a.pos[r];
            if (region >= 1) pre_st = a.row_stream[r];
        }
    }
    // FN: this thread's share of the norm.  Its loads go out ahead of the weight stream (loads complete in order: a wait
    // for a load issued behind the weights would wait for the weights too); the arithmetic runs underneath that stream.
    float fz[8];
    u32x4 fwv = {0u, 0u, 0u, 0u}, frv = {0u, 0u, 0u, 0u};
    f32x4 fp[FN > 0 ? FN : 1][2];
    const int frow = w >> 1, ft = (w & 1) * 64 + lane;    // row and thread-of-128 of the norm kernel this thread stands for
    if constexpr (FN >= 0) {
        static_assert(FN < 0 || (MT == 1 && KW == 8 && KW * KT * 32 == 1024), "the fused norm is built for one row tile, K = 1024");
        if (frow < a.M) {
            frv = *reinterpret_cast<const u32x4*>(a.fn_res_in + (size_t)frow * 1024 + ft * 8);
            fwv = *reinterpret_cast<const u32x4*>(a.fn_w + ft * 8);
            norm_load8<(FN > 0 ? FN : 0)>(a.fn_slab + (size_t)frow * 1024 + ft * 8, (size_t)a.M * 1024, fp);
        }
    }
    // AM: lane (row = lane & 15, g = lane >> 4) needs dims 8g..8g+7 of k-tile kt of head kt0 / 2: column group c = 4 kt + g of
    // that head's attention output, merged from the eight wave states exactly as the attention kernel merges them
    bf16x8 am[AM ? KT : 1];
    if constexpr (AM) {
        static_assert(!AM || (MT == 1 && KT == 2 && EPI == EPI_SLAB), "the state merge feeds the o_proj slab form");
        const int row = lane & 15, head = kt0 >> 1;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row < a.M)
                v = att_merge_waves(a.a_states + ((((size_t)row * 16 + head) * 8) * 8 + 4 * kt + (lane >> 4)) * ATT_STATE_FLOATS,
                                    8 * ATT_STATE_FLOATS);
            am[kt] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // weight fragments: the HBM stream
    constexpr int NWF = W8 ? 1 : KT, NWQ = W8 ? KT / 2 : 1;
    bf16x8 wf[NB][NWF];
    u32x4 wq[NB][NWQ];
    if constexpr (W8) {
        static_assert(!W8 || KT % 2 == 0, "int8 tiles hold two k-tiles");
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const u32x4* wp = reinterpret_cast<const u32x4*>(a.Wt) + ((size_t)nbi[nb] * (ktiles >> 1) + (kt0 >> 1)) * 64 + lane;
#pragma unroll
            for (int kp = 0; kp < NWQ; ++kp) wq[nb][kp] = __builtin_nontemporal_load(wp + kp * 64);
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.Wt) + ((size_t)nbi[nb] * ktiles + kt0) * 64 + lane;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) wf[nb][kt] = __builtin_nontemporal_load(wp + kt * 64);
        }
    }
    u32x2 pre_cs = {0u, 0u}, pre_sn = {0u, 0u};
    int pre_page = 0;
    if constexpr (EPI == EPI_QKV_ROPE) {
        if (w < MT) {
            if (region < 2) {
                const bf16_t* tb = a.rope + (size_t)pre_pos * a.kv.hd + half * 16 + 4 * (lane >> 4);
                pre_cs = *reinterpret_cast<const u32x2*>(tb);
                pre_sn = *reinterpret_cast<const u32x2*>(tb + 32);
            }
            if (region >= 1 && pre_st >= 0)
                pre_page = a.kv.block_table[(size_t)pre_st * a.kv.max_pages + (pre_pos >> 6)];
        }
    }
    if constexpr (FN >= 0) {
        float ss = 0.f;
        if (frow < a.M) {
            unpack8(frv, fz);
            norm_sum8<(FN > 0 ? FN : 0)>(fp, fz, a.fn_pair_scale ? a.fn_pair_scale + ft * 8 : nullptr);
            if constexpr (FN > 0) {
                if (blockIdx.x == 0 && blockIdx.y == 0)
                    *reinterpret_cast<u32x4*>(a.fn_res_out + (size_t)frow * 1024 + ft * 8) = pack8(fz);
            }
            ss = wave_sum(norm_ssq8(fz));
            if (lane == 0) fn_part[frow][w & 1] = ss;
        }
        __syncthreads();
        if (frow < a.M) {
            const float sc = norm_rstd(fn_part[frow][0] + fn_part[frow][1], a.fn_eps);
            *reinterpret_cast<u32x4*>(fn_tile + (ft >> 2) * 512 + ((ft & 3) * 16 + frow) * 8) = norm_scale8(fz, sc, fwv);
        }
        __syncthreads();
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        bf16x8 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (FN >= 0) af[mt] = *reinterpret_cast<const bf16x8*>(fn_tile + ((kt0 + kt) * 64 + lane) * 8);
            else if constexpr (AM) af[mt] = am[kt];
            else af[mt] = *reinterpret_cast<const bf16x8*>(arow[mt] + kt * a_step);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            bf16x8 wv;
            if constexpr (W8) wv = dequant8(wq[nb][kt >> 1][2 * (kt & 1)], wq[nb][kt >> 1][2 * (kt & 1) + 1]);
            else wv = wf[nb][kt];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[nb][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, af[mt], acc[nb][mt], 0, 0, 0);
        }
    }
    f32x4 wsc[NB];
    if constexpr (W8) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            wsc[nb] = *reinterpret_cast<const f32x4*>(a.wscale + nbi[nb] * 16 + 4 * (lane >> 4));
    }

    T3_STAMP(1);
#ifdef T3_GEMM_TIMELINE
    if (acc[0][0][0] == 1.2345e-30f) a.ts[0] = 0;     // pins stamp 2 behind the arrival of the MFMA results
#endif
    T3_STAMP(2);
    // cross-wave K reduction through LDS, fixed order
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) red[(w * NT + nb * MT + mt) * 64 + lane] = acc[nb][mt];
    __syncthreads();
    T3_STAMP(3);

    for (int mt = w; mt < MT; mt += KW) {
        f32x4 s[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s[nb] = red[(nb * MT + mt) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < KW; ++w2) s[nb] += red[(w2 * NT + nb * MT + mt) * 64 + lane];
            if constexpr (W8) s[nb] *= wsc[nb];
        }
        const int r = m0 + mt * 16 + (lane & 15);
        if (r >= a.M) continue;
        const int nl = 4 * (lane >> 4);

        if constexpr (EPI == EPI_SLAB) {
            // fp32 partial sums of this block's K range: slab[blockIdx.z][row][col]
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                *reinterpret_cast<f32x4*>(a.slab + ((size_t)blockIdx.z * a.M + r) * a.N + nbi[nb] * 16 + nl) = s[nb];
        } else if constexpr (EPI == EPI_STORE || EPI == EPI_STORE_GATHER || EPI == EPI_RESID) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = nbi[nb] * 16 + nl;
                bf16_t* cp = a.C + (size_t)r * a.ldc + n;
                float v[4] = {rbf(s[nb][0]), rbf(s[nb][1]), rbf(s[nb][2]), rbf(s[nb][3])};
                const bool vec = ((a.ldc & 3) == 0) && (n + 3 < a.n_valid);
                if constexpr (EPI == EPI_RESID) {
                    if (vec) {
                        u32x2 o = *reinterpret_cast<const u32x2*>(cp);
                        v[0] += bf_lo(o[0]); v[1] += bf_hi(o[0]); v[2] += bf_lo(o[1]); v[3] += bf_hi(o[1]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (n + i < a.n_valid) v[i] += bf2f(cp[i]);
                    }
                }
                if (vec) {
                    u32x2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(cp) = o;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (n + i < a.n_valid) cp[i] = f2bf(v[i]);
                }
            }
        } else if constexpr (EPI == EPI_SILU) {
            const int F = a.N >> 1;
            float hv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = rbf(s[0][i]), u = rbf(s[1][i]);
                const float sg = rbf(g / (1.0f + expf(-g)));
                hv[i] = sg * u;
            }
            u32x2 o = {pack_bf2(hv[0], hv[1]), pack_bf2(hv[2], hv[3])};
            const int n = b * 16 + nl;
            const size_t off = a.c_tiled ? ((size_t)(r >> 4) * (F >> 5) + (n >> 5)) * 512 + (((n >> 3) & 3) * 16 + (r & 15)) * 8 + (n & 7)
                                         : (size_t)r * F + n;
            *reinterpret_cast<u32x2*>(a.C + off) = o;
        } else {   // EPI_QKV_ROPE
            const int hd = a.kv.hd;                       // 64
            const int D = H * hd;
            float x[4], y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { x[i] = rbf(s[0][i]); y[i] = rbf(s[1][i]); }
            const int pos = pre_pos;
            int c0, c1;                                   // column inside the head of x[0], y[0]
            if (region < 2) {
                c0 = half * 16 + nl; c1 = c0 + 32;
                const u32x2 cs = pre_cs, sn = pre_sn;
                const float c[4] = {bf_lo(cs[0]), bf_hi(cs[0]), bf_lo(cs[1]), bf_hi(cs[1])};
                const float sv[4] = {bf_lo(sn[0]), bf_hi(sn[0]), bf_lo(sn[1]), bf_hi(sn[1])};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xr = x[i] * c[i] - y[i] * sv[i];
                    const float yr = y[i] * c[i] + x[i] * sv[i];
                    x[i] = xr; y[i] = yr;
                }
            } else {
                c0 = half * 32 + nl; c1 = c0 + 16;
            }
            const u32x2 ox = {pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
            const u32x2 oy = {pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
            bf16_t* qrow = a.C + (size_t)r * a.ldc + region * D + head * hd;
            *reinterpret_cast<u32x2*>(qrow + c0) = ox;
            *reinterpret_cast<u32x2*>(qrow + c1) = oy;
            if (region >= 1) {
                if (pre_st >= 0) {
                    const int page = pre_page;
                    bf16_t* base = a.kv.pool + (size_t)a.layer * a.kv.layer_stride +
                                   ((size_t)page * 2 + (region - 1)) * ((size_t)H * T3_PAGE * hd) +
                                   ((size_t)head * T3_PAGE + (pos & 63)) * hd;
                    *reinterpret_cast<u32x2*>(base + c0) = ox;
                    *reinterpret_cast<u32x2*>(base + c1) = oy;
                }
            }
        }
    }
    T3_STAMP(4);
}

// Tile repack: one thread per 8 consecutive k of one W row.
__global__ void repack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wt, int N, int K) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k8 = K >> 3;
    if (t >= (size_t)N * k8) return;
    const int n = (int)(t / k8), kc = (int)(t % k8);
    const int nb = n >> 4, nr = n & 15, kb = kc >> 2, q = kc & 3;
    const u32x4 v = *reinterpret_cast<const u32x4*>(w + (size_t)n * K + kc * 8);
    const size_t dst = (((size_t)nb * (K >> 5) + kb) * 64 + (q * 16 + nr)) * 8;
    *reinterpret_cast<u32x4*>(wt + dst) = v;
}

int launch_repack(const bf16_t* w, bf16_t* wt, int N, int K, hipStream_t st) {
    T3_REQUIRE(N % 16 == 0 && K % 32 == 0, "repack: N must be a multiple of 16 and K of 32");
    const size_t n = (size_t)N * (K / 8);
    T3_LAUNCH(repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, wt, N, K);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// int8 quantisation of one weight row per workgroup: scale = absmax / 127, q = rint(w / scale) (ties to
// even, as torch.round), stored offset-binary in the pair-tile order the W8 kernels read.
__global__ __launch_bounds__(256) void quantize_kernel(const bf16_t* __restrict__ w, uint8_t* __restrict__ wq,
                                                       float* __restrict__ scale, int K) {
    __shared__ float part[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const bf16_t* row = w + (size_t)n * K;
    float amax = 0.f;
    for (int k = tid * 8; k < K; k += 256 * 8) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(row + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(bf_lo(v[i])), fabsf(bf_hi(v[i]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((tid & 63) == 0) part[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    const float sc = amax > 0.f ? amax / 127.0f : 1.0f;
    if (tid == 0) scale[n] = sc;
    const int nb = n >> 4, nr = n & 15, kpairs = K >> 6;
    for (int k8 = tid; k8 < (K >> 3); k8 += 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(row + k8 * 8);
        unsigned b[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (j & 1) ? bf_hi(v[j >> 1]) : bf_lo(v[j >> 1]);
            float q = rintf(x / sc);
            q = fminf(fmaxf(q, -127.0f), 127.0f);
            b[j >> 2] |= (unsigned)((int)q + 128) << (8 * (j & 3));
        }
        const int kp = k8 >> 3, t = (k8 >> 2) & 1, lane = (k8 & 3) * 16 + nr;
        const u32x2 o = {b[0], b[1]};
        *reinterpret_cast<u32x2*>(wq + (((size_t)nb * kpairs + kp) * 64 + lane) * 16 + t * 8) = o;
    }
}

int launch_quantize(const bf16_t* w, uint8_t* wq, float* scale, int N, int K, hipStream_t st) {
    T3_REQUIRE(N % 16 == 0 && K % 64 == 0, "quantize: N must be a multiple of 16 and K of 64");
    T3_LAUNCH(quantize_kernel, dim3(N), dim3(256), 0, st, w, wq, scale, K);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int NB, int MT, int KW, int KT, int EPI, bool W8, int FN = -1, bool AM = false>
static int launch_w(const GemmArgs& a, int nblocks, hipStream_t st, int ksplit) {
    constexpr size_t lds = (size_t)KW * NB * MT * 64 * sizeof(f32x4);
    // the attribute belongs to the kernel's copy on ONE device: set it once per device this variant is launched on
    static bool attr_set[T3_MAX_DEVICES] = {};
    int dev = 0;
    T3_HIP_CHECK(hipGetDevice(&dev));
    T3_REQUIRE(dev >= 0 && dev < T3_MAX_DEVICES, "gemm: device ordinal out of range");
    if (!attr_set[dev]) {
        T3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<NB, MT, KW, KT, EPI, W8, FN, AM>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    const int groups = (a.M + MT * 16 - 1) / (MT * 16);
    T3_LAUNCH((gemm_stream_kernel<NB, MT, KW, KT, EPI, W8, FN, AM>), dim3(nblocks, groups, ksplit), dim3(KW * 64), lds, st,
              a.A, a.Wt, a.pos, a.row_stream, a.M, a.N, a.K, a.lda, a.kv.H, (a.a_tiled ? 1 : 0) | (a.c_tiled ? 2 : 0), a);
    T3_HIP_CHECK(hipGetLastError());
    return 0;
}

// bf16 or int8 weights by GemmArgs::wscale; KT odd has no int8 form (its callers never pass a scale)
template <int NB, int MT, int KW, int KT, int EPI>
static int launch_one(const GemmArgs& a, int nblocks, hipStream_t st, int ksplit = 1) {
    if constexpr (KT % 2 == 0) {
        if (a.wscale) return launch_w<NB, MT, KW, KT, EPI, true>(a, nblocks, st, ksplit);
    } else {
        T3_REQUIRE(a.wscale == nullptr, "gemm: this shape has no int8 form");
    }
    return launch_w<NB, MT, KW, KT, EPI, false>(a, nblocks, st, ksplit);
}

// the projection with the residual-add + RMSNorm in its prologue (GemmArgs::fn_*), a.M <= GEMM_FUSED_NORM_ROWS
template <int EPI>
static int launch_fused_norm(const GemmArgs& a, int nblocks, hipStream_t st) {
    T3_REQUIRE(a.M <= GEMM_FUSED_NORM_ROWS && a.K == 1024, "gemm: the norm prologue takes at most 4 rows and K = 1024");
    T3_REQUIRE(a.fn_res_in && (a.fn_nsplit == 0 || ((a.fn_nsplit == GEMM_SLAB_SPLIT || a.fn_nsplit == ATTN_OPROJ_SPLIT) &&
                                                    a.fn_slab && a.fn_res_out && a.fn_res_out != a.fn_res_in)),
               "gemm: the norm prologue needs the residual rows, and for folded slabs (8 or 16) a separate buffer for the new rows");
    if (a.fn_nsplit == 0) {
        if (a.wscale) return launch_w<2, 1, 8, 4, EPI, true, 0>(a, nblocks, st, 1);
        return launch_w<2, 1, 8, 4, EPI, false, 0>(a, nblocks, st, 1);
    }
    if (a.fn_nsplit == ATTN_OPROJ_SPLIT) {
        if (a.wscale) return launch_w<2, 1, 8, 4, EPI, true, ATTN_OPROJ_SPLIT>(a, nblocks, st, 1);
        return launch_w<2, 1, 8, 4, EPI, false, ATTN_OPROJ_SPLIT>(a, nblocks, st, 1);
    }
    if (a.wscale) return launch_w<2, 1, 8, 4, EPI, true, GEMM_SLAB_SPLIT>(a, nblocks, st, 1);
    return launch_w<2, 1, 8, 4, EPI, false, GEMM_SLAB_SPLIT>(a, nblocks, st, 1);
}

template <int NB, int KW, int KT, int EPI>
static int launch_mt(const GemmArgs& a, int nblocks, hipStream_t st, int ksplit = 1) {
    if (a.M <= 16) return launch_one<NB, 1, KW, KT, EPI>(a, nblocks, st, ksplit);
    if (a.M <= 32) return launch_one<NB, 2, KW, KT, EPI>(a, nblocks, st, ksplit);
    if (a.M <= 64) return launch_one<NB, 4, KW, KT, EPI>(a, nblocks, st, ksplit);
    return launch_one<NB, 8, KW, KT, EPI>(a, nblocks, st, ksplit);
}

int launch_gemm(int epi, const GemmArgs& a, hipStream_t st) {
    T3_REQUIRE(a.M >= 1, "gemm: no rows");
    T3_REQUIRE(a.N % 16 == 0, "gemm: N must be a multiple of 16");
    T3_REQUIRE(a.lda % 8 == 0, "gemm: lda must be a multiple of 8");
    T3_REQUIRE(!(a.a_tiled && a.a_rows), "gemm: a row gather reads row-major activations");
    T3_REQUIRE(!a.c_tiled || epi == EPI_SILU, "gemm: only the SiLU form writes tile-ordered output");
    T3_REQUIRE(a.K == 1024 || a.K == 4096, "gemm: K must be 1024 or 4096");
    T3_REQUIRE(a.fn_w == nullptr || epi == EPI_SILU || epi == EPI_QKV_ROPE, "gemm: only the SiLU and QKV forms take the norm prologue");
    switch (epi) {
        case EPI_STORE:
            if (a.a_rows) {
                T3_REQUIRE(a.K == 1024, "gemm: row gather is built for K = 1024");
                return launch_mt<1, 8, 4, EPI_STORE_GATHER>(a, a.N / 16, st);
            }
            // many columns (the speech head: 513 column blocks): three blocks per workgroup put the launch into one
            // generation of workgroups; the per-element accumulation order does not depend on the grouping
            if (a.K == 1024 && a.N % 48 == 0 && a.N >= 4096 && a.n_valid == a.N && a.M <= 64)   // MT = 8 would not fit the LDS reduction
                return launch_mt<3, 8, 4, EPI_STORE>(a, a.N / 48, st);
            if (a.K == 1024) return launch_mt<1, 8, 4, EPI_STORE>(a, a.N / 16, st);
            return launch_mt<1, 16, 8, EPI_STORE>(a, a.N / 16, st);
        case EPI_RESID:
            if (a.K == 1024) return launch_mt<1, 8, 4, EPI_RESID>(a, a.N / 16, st);
            return launch_mt<1, 16, 8, EPI_RESID>(a, a.N / 16, st);
        case EPI_SILU:
            T3_REQUIRE(a.K == 1024 && a.N % 64 == 0, "gemm_silu: K must be 1024, N = 2F with F % 32 == 0");
            if (a.fn_w) return launch_fused_norm<EPI_SILU>(a, a.N / 32, st);
            return launch_mt<2, 8, 4, EPI_SILU>(a, a.N / 32, st);
        case EPI_QKV_ROPE:
            T3_REQUIRE(a.K == 1024 && a.kv.hd == 64 && a.N == 3 * a.kv.H * 64, "gemm_qkv: needs K=1024, head_dim 64");
            T3_REQUIRE(a.ldc % 4 == 0, "gemm_qkv: ldc must be a multiple of 4");
            if (a.fn_w) return launch_fused_norm<EPI_QKV_ROPE>(a, a.N / 32, st);
            // only 96 column pairs: above 32 rows, row groups of 32 put twice the workgroups to work, each
            // reading half the activations (the groups of one column pair share an XCD, hence its L2 copy of W)
            if (a.M > 32) return launch_one<2, 2, 8, 4, EPI_QKV_ROPE>(a, a.N / 32, st);
            return launch_mt<2, 8, 4, EPI_QKV_ROPE>(a, a.N / 32, st);
    }
    t3_set_error("gemm: unknown epilogue");
    return 1;
}

// Split-K form for the two projections with few output columns (o_proj, down_proj: N = 1024):
// GEMM_SLAB_SPLIT blocks share one 32-column group, each over 1/8 of K, so 256 workgroups stream
// the weights; every block writes its fp32 partial tile to slab[split][row][col].  The partials
// are summed in split order by the residual-add + RMSNorm kernel that follows.
int launch_gemm_slab(const GemmArgs& a, hipStream_t st) {
    T3_REQUIRE(a.slab != nullptr && a.M >= 1 && a.N % 32 == 0 && a.lda % 8 == 0, "gemm_slab: bad arguments");
    if (a.a_states) {                                 // o_proj behind the few-row split attention: rows merged in the kernel
        T3_REQUIRE(a.K == 1024 && a.N == 1024 && a.M <= GEMM_FUSED_NORM_ROWS, "gemm_slab: the state merge is built for o_proj of at most 4 rows");
        if (a.wscale) return launch_w<2, 1, 2, 2, EPI_SLAB, true, -1, true>(a, a.N / 32, st, GEMM_SLAB_SPLIT);
        return launch_w<2, 1, 2, 2, EPI_SLAB, false, -1, true>(a, a.N / 32, st, GEMM_SLAB_SPLIT);
    }
    if (a.K == 1024 && a.wscale) return launch_mt<2, 2, 2, EPI_SLAB>(a, a.N / 32, st, GEMM_SLAB_SPLIT);   // int8: 2 waves x 1 tile pair
    // K = 1024 (o_proj): split s = heads 2s, 2s+1; wave j chains the two k-tiles of head 2s+j on one accumulator and the two
    // waves' sums are added: exactly the additions the attention launch + 16-partial fold perform when o_proj runs there
    if (a.K == 1024) return launch_mt<2, 2, 2, EPI_SLAB>(a, a.N / 32, st, GEMM_SLAB_SPLIT);
    if (a.K == 4096) return launch_mt<2, 8, 2, EPI_SLAB>(a, a.N / 32, st, GEMM_SLAB_SPLIT);   // 8 waves x 2 k-tiles
    t3_set_error("gemm_slab: K must be 1024 or 4096");
    return 1;
}
