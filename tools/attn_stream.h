// Decode attention as a persistent, load-balanced launch (gfx950): included by attn.hip.
//
// attn_decode_kernel gives every (row, head) pair a workgroup of 8 waves.  A launch of 62 rows is 992 workgroups of which
// 512 are resident; a wave whose row is short has no tile, holds its slot until the workgroup's longest wave is done, and
// the launch ends with whatever workgroups the second generation left for last.  Here the launch is ONE workgroup of 8
// waves per CU for its whole duration, and the K/V tiles of the step are dealt to the CUs in equal shares:
//
//   * work item = (row, head, v): the tiles v, v + 8, ... of the pair, i.e. exactly what wave v of attn_decode_kernel
//     folds, in the same lane layout and order, ending in the same merged lane-group state.  Items of a pair are merged in
//     v order by the same operations as att_merge_waves, and a pair's items all live in one workgroup, so the output has
//     the bits of attn_decode_kernel whatever else shares the launch (skipping the empty states of waves without a tile
//     changes nothing: merging (m = -1e30, l = 0, o = 0) multiplies by exp(0) = 1 and adds 0 * 0).
//   * the host deals the pairs to the workgroups longest first onto the least loaded one (the step's row lengths are
//     known when the step is planned), so every CU streams the same number of tiles and all finish together;
//   * an item's descriptor carries its page numbers: a wave's first K/V request goes out after ONE dependent fetch
//     (attn_decode_kernel: row metadata, then the block table, then K/V);
//   * a wave keeps two tiles in flight (two register buffers): the next item's tile is requested before the current one
//     is folded.
#pragma once

struct AttItem {
    int32_t page[2];      // pool pages of tiles v and v + 8 (-1: no such tile)
    int32_t row;          // row of the step (q and output row)
    int32_t hv;           // head | v << 8 | items of the pair << 16
    int32_t ctx;          // positions of the row's stream, this step's included
    int32_t pfx;          // page of the shared prefix, or -1
    int32_t pad[2];
};
static_assert(sizeof(AttItem) == 32, "AttItem is read as two 16-byte pieces");

#define ATT_STREAM_MAX_ITEMS 192          // items one workgroup may be dealt (LDS: 320 B of state each)

// A tile by page number (att_load_tile looks the page up in the block table).
__device__ __forceinline__ void att_load_tile_page(AttTile& t, const bf16_t* __restrict__ lbase, int H, int page, int tile, int h,
                                                   int lane, int pfx, int npfx) {
    const int tg = lane >> 3;
    const size_t slab = (size_t)T3_PAGE * 64;
    const bf16_t* kb = lbase + ((size_t)page * 2 * H + h) * slab + lane * 8;
    const bool shared = tile == 0 && pfx >= 0;
    const long long to_pfx = shared ? ((long long)pfx - page) * (long long)(2 * H) * (long long)slab : 0;
    const int lim = shared ? npfx : 0;
    const size_t vofs = (size_t)H * slab;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const bf16_t* p = kb + it * 512 + (it * 8 + tg < lim ? to_pfx : 0);
        t.k[it] = ATT_LOAD(reinterpret_cast<const u32x4*>(p));
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const bf16_t* p = kb + vofs + it * 512 + (it * 8 + tg < lim ? to_pfx : 0);
        t.v[it] = ATT_LOAD(reinterpret_cast<const u32x4*>(p));
    }
}

// att_merge_waves over the n >= 1 states of a pair that exist (same operations, same order).
__device__ __forceinline__ u32x4 att_merge_items(const float* sp, int stride, int n) {
    float M = sp[0], L = sp[1], O[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) O[j] = sp[2 + j];
    for (int w2 = 1; w2 < n; ++w2) {
        const float* p = sp + (size_t)w2 * stride;
        const float m2 = p[0], l2 = p[1];
        const float mn = fmaxf(M, m2);
        const float a1 = __expf(M - mn), a2 = __expf(m2 - mn);
        L = fmaf(L, a1, l2 * a2);
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

// One workgroup per bin: items[bin_start[b] .. bin_start[b + 1]), a pair's items adjacent and in v order.
template <bool DBUF>
__global__ __launch_bounds__(ATT_NW * 64, DBUF ? 2 : 4) void attn_decode_stream_kernel(
        const bf16_t* __restrict__ qkv, int ldq, bf16_t* __restrict__ out, int ldo, const AttItem* __restrict__ items,
        const int32_t* __restrict__ bin_start, const bf16_t* __restrict__ pool_layer, int H, int npfx) {
    __shared__ float part[ATT_STREAM_MAX_ITEMS][8][ATT_STATE_FLOATS];
    __shared__ __attribute__((aligned(16))) int32_t dsc[ATT_STREAM_MAX_ITEMS][8];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 7, tg = lane >> 3;
    const int i0 = bin_start[blockIdx.x], n = bin_start[blockIdx.x + 1] - i0;
    // the bin's descriptors go through LDS: a wave reads them without waiting for the K/V loads it has in flight
    // (vector loads return in order; LDS reads are counted separately)
    for (int t = threadIdx.x; t < 2 * n; t += ATT_NW * 64)
        reinterpret_cast<u32x4*>(&dsc[0][0])[t] = reinterpret_cast<const u32x4*>(items + i0)[t];
    __syncthreads();

    auto desc = [&](int i) -> AttItem {
        const u32x4 a = *reinterpret_cast<const u32x4*>(&dsc[i][0]);
        const u32x4 b = *reinterpret_cast<const u32x4*>(&dsc[i][4]);
        AttItem d;
        d.page[0] = __builtin_amdgcn_readfirstlane(a[0]); d.page[1] = __builtin_amdgcn_readfirstlane(a[1]);
        d.row = __builtin_amdgcn_readfirstlane(a[2]); d.hv = __builtin_amdgcn_readfirstlane(a[3]);
        d.ctx = __builtin_amdgcn_readfirstlane(b[0]); d.pfx = __builtin_amdgcn_readfirstlane(b[1]);
        d.pad[0] = d.pad[1] = 0;
        return d;
    };
    auto request = [&](AttTile& t, const AttItem& d, int sub) {
        const int v = (d.hv >> 8) & 0xff;
        att_load_tile_page(t, pool_layer, H, d.page[sub], v + 8 * sub, d.hv & 0xff, lane, d.pfx, npfx);
    };
    auto load_q = [&](const AttItem& d) -> u32x4 {
        return *reinterpret_cast<const u32x4*>(qkv + (size_t)d.row * ldq + (d.hv & 0xff) * 64 + c * 8);
    };

    float q[8], m = -1e30f, l = 0.f, o[8];
    auto begin_item = [&](const u32x4& qv) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { q[2 * k] = bf_lo(qv[k]); q[2 * k + 1] = bf_hi(qv[k]); }
        m = -1e30f; l = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = 0.f;
    };
    auto end_item = [&](int i) {
        att_merge_groups(m, l, o);
        if (tg == 0) {
            part[i][c][0] = m; part[i][c][1] = l;
#pragma unroll
            for (int j = 0; j < 8; ++j) part[i][c][2 + j] = o[j];
        }
    };

    // The wave's tile sequence: items w, w + 8, ... of the bin, each one or two tiles (v, v + 8).
    AttTile buf[DBUF ? 2 : 1];
    int i = w;
    if (i < n) {
        AttItem d = desc(i);
        AttItem dn = desc(i + ATT_NW < n ? i + ATT_NW : i);
        u32x4 qraw = load_q(d), qraw_n = qraw;
        int sub = 0;
        request(buf[0], d, 0);
        // one tile: `curb` holds tile (d, sub); with two buffers the tile after it is requested first (and the q row of
        // the item it opens, ahead of its K/V so that it arrives ahead of them)
        auto phase = [&](AttTile& curb, AttTile& nxtb) -> bool {
            const bool last_of_item = sub == 1 || d.page[1] < 0;
            const bool has_next_item = i + ATT_NW < n;
            if constexpr (DBUF) {
                if (!last_of_item) request(nxtb, d, 1);
                else if (has_next_item) { qraw_n = load_q(dn); request(nxtb, dn, 0); }
            }
            if (sub == 0) begin_item(qraw);
            att_fold_tile(curb, q, ((d.hv >> 8) & 0xff) + 8 * sub, d.ctx, tg, m, l, o);
            if (!last_of_item) {
                if constexpr (!DBUF) request(nxtb, d, 1);
                sub = 1;
                return true;
            }
            end_item(i);
            if (!has_next_item) return false;
            if constexpr (!DBUF) { qraw_n = load_q(dn); request(nxtb, dn, 0); }
            i += ATT_NW; d = dn; sub = 0; qraw = qraw_n;
            dn = desc(i + ATT_NW < n ? i + ATT_NW : i);
            return true;
        };
        if constexpr (DBUF) { while (phase(buf[0], buf[1]) && phase(buf[1], buf[0])) {} }
        else { while (phase(buf[0], buf[0])) {} }
    }
    __syncthreads();
    // merge every pair of the bin: 8 threads (one per 8-dim chunk) per pair, pairs found at their v = 0 item
    for (int t = threadIdx.x; t < 8 * n; t += ATT_NW * 64) {
        const int it = t >> 3, cc = t & 7;
        const int hv = dsc[it][3];
        if (((hv >> 8) & 0xff) != 0) continue;
        const u32x4 ov = att_merge_items(&part[it][cc][0], 8 * ATT_STATE_FLOATS, (hv >> 16) & 0xff);
        *reinterpret_cast<u32x4*>(out + (size_t)dsc[it][2] * ldo + (hv & 0xff) * 64 + cc * 8) = ov;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same deal with the K/V tiles streamed through LDS by global_load_lds_dwordx4 (no register destination): a workgroup
// of 4 waves per CU, two 16 KiB tile slots per wave.  A wave requests tile k + 1 into its other slot (16 LDS-DMA
// instructions of 1 KiB each), waits until all but those 16 have landed, reads tile k from LDS into registers in the lane
// layout of att_load_tile and folds it with att_fold_tile: the same bits again.  The q rows and the descriptors of the
// bin are staged into LDS once, before the first LDS-DMA, so that no ordinary vector load has to be waited for while
// LDS-DMA is in flight.  The instructions are issued from inline assembly and waited for by explicit counts.
// ---------------------------------------------------------------------------------------------------------------------
#define ATT_LDS_NW 4
#define ATT_LDS_MAX_ITEMS 48

__device__ __forceinline__ void att_glds_4k(const bf16_t* g, uint32_t lds) {     // 4 KiB: lane address g, LDS base lds
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

__global__ __launch_bounds__(ATT_LDS_NW * 64, 1) void attn_decode_lds_kernel(
        const bf16_t* __restrict__ qkv, int ldq, bf16_t* __restrict__ out, int ldo, const AttItem* __restrict__ items,
        const int32_t* __restrict__ bin_start, const bf16_t* __restrict__ pool_layer, int H, int npfx) {
    __shared__ __attribute__((aligned(1024))) uint8_t slots[ATT_LDS_NW * 2 * 16384];      // [wave][2][16 KiB]
    __shared__ float part[ATT_LDS_MAX_ITEMS][8][ATT_STATE_FLOATS];
    __shared__ __attribute__((aligned(16))) int32_t dsc[ATT_LDS_MAX_ITEMS][8];
    __shared__ __attribute__((aligned(16))) bf16_t qs[ATT_LDS_MAX_ITEMS][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 7, tg = lane >> 3;
    const int i0 = bin_start[blockIdx.x], n = bin_start[blockIdx.x + 1] - i0;
    for (int t = threadIdx.x; t < 2 * n; t += ATT_LDS_NW * 64)
        reinterpret_cast<u32x4*>(&dsc[0][0])[t] = reinterpret_cast<const u32x4*>(items + i0)[t];
    for (int t = threadIdx.x; t < 8 * n; t += ATT_LDS_NW * 64) {
        const int32_t* dp = reinterpret_cast<const int32_t*>(items + i0 + (t >> 3));
        *reinterpret_cast<u32x4*>(&qs[t >> 3][(t & 7) * 8]) =
            *reinterpret_cast<const u32x4*>(qkv + (size_t)dp[2] * ldq + (dp[3] & 0xff) * 64 + (t & 7) * 8);
    }
    __syncthreads();                                        // every ordinary load of the kernel has been waited for here

    const uint32_t slot0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)slots + (uint32_t)w * 32768u;
    auto request = [&](int slot, int i, int sub) {
        const int page = __builtin_amdgcn_readfirstlane(dsc[i][sub]);
        const int hv = __builtin_amdgcn_readfirstlane(dsc[i][3]);
        const int pfx = __builtin_amdgcn_readfirstlane(dsc[i][5]);
        const int tile = ((hv >> 8) & 0xff) + 8 * sub, h = hv & 0xff;
        const size_t slab = (size_t)T3_PAGE * 64;
        const bf16_t* kb = pool_layer + ((size_t)page * 2 * H + h) * slab + lane * 8;
        const bool shared = tile == 0 && pfx >= 0;
        const long long to_pfx = shared ? ((long long)pfx - page) * (long long)(2 * H) * (long long)slab : 0;
        const int lim = shared ? npfx : 0;
        const uint32_t base = slot0 + (uint32_t)slot * 16384u;
        if (!shared) {
            att_glds_4k(kb, base);
            att_glds_4k(kb + 2048, base + 4096);
            att_glds_4k(kb + (size_t)H * slab, base + 8192);
            att_glds_4k(kb + (size_t)H * slab + 2048, base + 12288);
        } else {                                           // tile 0 with a shared prefix: per-lane source pages, piece by piece
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const bf16_t* p = kb + (it >> 3) * (size_t)H * slab + (it & 7) * 512 + ((it & 7) * 8 + tg < lim ? to_pfx : 0);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p), "s"(base + (uint32_t)it * 1024u) : "memory");
            }
        }
    };
    float q[8], m = -1e30f, l = 0.f, o[8];
    AttTile t;
    auto fold = [&](int slot, int i, int sub) {
        const uint8_t* sp = slots + (size_t)w * 32768 + (size_t)slot * 16384 + lane * 16;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            t.k[it] = *reinterpret_cast<const u32x4*>(sp + it * 1024);
            t.v[it] = *reinterpret_cast<const u32x4*>(sp + 8192 + it * 1024);
        }
        const int hv = __builtin_amdgcn_readfirstlane(dsc[i][3]);
        att_fold_tile(t, q, ((hv >> 8) & 0xff) + 8 * sub, __builtin_amdgcn_readfirstlane(dsc[i][4]), tg, m, l, o);
    };
    // the wave's tile sequence (items w, w + 4, ...; one or two tiles each), one tile ahead in the other slot
    int i = w, sub = 0, k = 0;
    if (i < n) {
        request(0, i, 0);
        while (true) {
            const bool last_of_item = sub == 1 || dsc[i][1] < 0;
            const bool has_next_item = i + ATT_LDS_NW < n;
            const bool more = !last_of_item || has_next_item;
            if (!last_of_item) request((k + 1) & 1, i, 1);
            else if (has_next_item) request((k + 1) & 1, i + ATT_LDS_NW, 0);
            if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (sub == 0) {
                const u32x4 qv = *reinterpret_cast<const u32x4*>(&qs[i][c * 8]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { q[2 * kk] = bf_lo(qv[kk]); q[2 * kk + 1] = bf_hi(qv[kk]); }
                m = -1e30f; l = 0.f;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) o[kk] = 0.f;
            }
            fold(k & 1, i, sub);
            ++k;
            if (!last_of_item) { sub = 1; continue; }
            att_merge_groups(m, l, o);
            if (tg == 0) {
                part[i][c][0] = m; part[i][c][1] = l;
#pragma unroll
                for (int j = 0; j < 8; ++j) part[i][c][2 + j] = o[j];
            }
            if (!has_next_item) break;
            i += ATT_LDS_NW; sub = 0;
        }
    }
    __syncthreads();
    for (int tt = threadIdx.x; tt < 8 * n; tt += ATT_LDS_NW * 64) {
        const int it = tt >> 3, cc = tt & 7;
        const int hv = dsc[it][3];
        if (((hv >> 8) & 0xff) != 0) continue;
        const u32x4 ov = att_merge_items(&part[it][cc][0], 8 * ATT_STATE_FLOATS, (hv >> 16) & 0xff);
        *reinterpret_cast<u32x4*>(out + (size_t)dsc[it][2] * ldo + (hv & 0xff) * 64 + cc * 8) = ov;
    }
}
