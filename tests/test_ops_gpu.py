"""Per-kernel parity on the GPU, driven through the C ABI (t3_op_* of include/t3_hip.h).

Each kernel is checked against a plain fp32 PyTorch restatement of the same arithmetic
contract the CPU oracle uses (oracle/t3_oracle.py header).  bf16 outputs may differ from
the checker only where the fp32 summation order flips a rounding: at most 1 bf16 ulp (at
the magnitude of the last op's operands) on a few percent of the elements; integer results
(token ids, cache placement) are exact.

Every device tensor handed to a kernel is bound to a local name that outlives the
synchronize: the kernels take raw addresses.
"""
import ctypes as C

import pytest
import torch

from chatterbox_vllm_amd.config import T3ModelConfig
from chatterbox_vllm_amd import weights as W
from _util import assert_bf16_close, call, p, repack, r16

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("M", [1, 7, 16, 33, 64, 100, 128, 300])
@pytest.mark.parametrize("N,K", [(1024, 1024), (1024, 4096), (8208, 1024)])
def test_gemm_store_and_residual(capi, M, N, K):
    a = _rand((M, K), 1).to(DEV)
    w = _rand((N, K), 2, 0.02).to(DEV)
    wt = repack(capi, w)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_gemm", p(a), p(wt), p(out), M, N, K, N, 0, None)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().T
    assert_bf16_close(out, ref, what=f"gemm M={M} N={N} K={K}")
    if N == 1024:
        res = _rand((M, N), 3).to(DEV)
        res0 = res.clone()
        call(capi, "t3_op_gemm", p(a), p(wt), p(res), M, N, K, N, 1, None)
        torch.cuda.synchronize()
        # bf16(bf16(acc) + res): a flipped rounding of acc moves the sum by one ulp of |acc|
        assert_bf16_close(res, r16(ref) + res0.float(), mag=torch.maximum(ref.abs(), res0.float().abs()),
                          ulps=1.5, what="gemm+residual")


def test_gemm_partial_columns_are_not_written(capi):
    M, N, K, ldc = 5, 8208, 1024, 8194          # speech head: 8194 real columns in a 16-padded matrix
    a = _rand((M, K), 4).to(DEV)
    w = _rand((N, K), 5, 0.02).to(DEV)
    wt = repack(capi, w)
    out = torch.full((M * ldc + 64,), 7.0, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_gemm", p(a), p(wt), p(out), M, N, K, ldc, 0, None)
    torch.cuda.synchronize()
    ref = (a.float() @ w.float().T)[:, :ldc]
    assert_bf16_close(out[: M * ldc].view(M, ldc), ref, what="gemm ldc<N")
    assert (out[M * ldc:] == 7.0).all()


def test_gemm_rows_do_not_depend_on_batch(capi):
    """Batch invariance: a row's result is bit-identical whatever else is in the launch."""
    K, N = 1024, 1024
    a = _rand((128, K), 6).to(DEV)
    w = _rand((N, K), 7, 0.02).to(DEV)
    wt = repack(capi, w)
    full = torch.zeros(128, N, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_gemm", p(a), p(wt), p(full), 128, N, K, N, 0, None)
    for M in (1, 3, 16, 40, 64):
        part = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        call(capi, "t3_op_gemm", p(a), p(wt), p(part), M, N, K, N, 0, None)
        torch.cuda.synchronize()
        assert torch.equal(part, full[:M]), f"rows change with M={M}"


# the weight-streaming (decode) and the LDS-tiled (prompt rows) form of each fused projection
FORMS = ["t3_op_gemm", "t3_op_gemm_prompt"]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("M", [1, 20, 64, 130, 400])
def test_gemm_silu(capi, M, form):
    K, F = 1024, 4096
    a = _rand((M, K), 8).to(DEV)
    w = _rand((2 * F, K), 9, 0.05).to(DEV)
    wt = repack(capi, w)
    h = torch.zeros(M, F, dtype=torch.bfloat16, device=DEV)
    call(capi, form + "_silu", p(a), p(wt), p(h), M, F, K, None)
    torch.cuda.synchronize()
    gu = r16(a.float() @ w.float().T)
    g, u = gu[:, :F], gu[:, F:]
    ref = r16(g / (1 + torch.exp(-g))) * u
    # a flipped rounding of g or of u each moves the product by about one ulp
    assert_bf16_close(h, ref, ulps=3, max_frac=0.03, what="gemm_silu")


def _kv_setup(n_streams, max_pages, n_layers, H=16, hd=64, seed=0):
    n_pages = n_streams * max_pages
    page_elems = 2 * H * 64 * hd
    pool = torch.zeros(n_layers * n_pages * page_elems, dtype=torch.bfloat16, device=DEV)
    g = torch.Generator().manual_seed(seed)
    bt = torch.randperm(n_pages, generator=g).to(torch.int32).view(n_streams, max_pages).contiguous()
    return pool, bt, n_pages * page_elems


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("M", [1, 19, 64, 150, 389])
def test_gemm_qkv_rope_and_cache_write(capi, M, form):
    cfg = T3ModelConfig.english()
    H, hd, K = 16, 64, 1024
    a = _rand((M, K), 10).to(DEV)
    w = _rand((3 * H * hd, K), 11, 0.03).to(DEV)
    wt = repack(capi, w)
    rope = W.rope_table(cfg, 1000).to(DEV)
    g = torch.Generator().manual_seed(12)
    n_streams, max_pages = 4, 16
    pool, bt, layer_stride = _kv_setup(n_streams, max_pages, 2)
    # rows spread over streams with distinct positions per stream
    stream = (torch.arange(M) % n_streams).to(torch.int32)
    pos = torch.zeros(M, dtype=torch.int32)
    perm = {s: torch.randperm(1000, generator=g) for s in range(n_streams)}
    for r in range(M):
        pos[r] = perm[int(stream[r])][r // n_streams]
    if M > 2:
        stream[2] = -1                                   # a row that must not touch the cache
    qkv = torch.zeros(M, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    btd, posd, std = bt.to(DEV), pos.to(DEV), stream.to(DEV)
    call(capi, form + "_qkv_rope", p(a), p(wt), p(qkv), M, H, hd, K, p(rope), p(posd), p(std), p(pool),
         layer_stride, 1, p(btd), max_pages, None)
    torch.cuda.synchronize()
    ref = r16(a.float() @ w.float().T).cpu()
    cs = rope.float().cpu()[pos.long()]
    cos, sin = cs[:, None, :32], cs[:, None, 32:]

    def rot(t):
        t = t.view(M, H, hd)
        x, y = t[..., :32], t[..., 32:]
        out = torch.cat([x * cos - y * sin, y * cos + x * sin], -1).reshape(M, H * hd)
        mag = torch.maximum(x.abs(), y.abs())
        return out, torch.cat([mag, mag], -1).reshape(M, H * hd)
    (refq, magq), (refk, magk), refv = rot(ref[:, :1024]), rot(ref[:, 1024:2048]), ref[:, 2048:]
    got = qkv.cpu()
    # the rotation mixes two rounded GEMM outputs: one flipped input rounding = one ulp of that input
    assert_bf16_close(got[:, :1024], refq, mag=magq, ulps=2.5, max_frac=0.04, what="q")
    assert_bf16_close(got[:, 1024:2048], refk, mag=magk, ulps=2.5, max_frac=0.04, what="k")
    assert_bf16_close(got[:, 2048:], refv, what="v")
    # the cache holds exactly the rows just written, at (block_table[stream, pos/64], pos%64); layer 0 untouched
    poolc = pool.cpu()
    v = poolc[layer_stride:2 * layer_stride].view(-1, 2, H, 64, hd)
    for r in range(M):
        s = int(stream[r])
        if s < 0:
            continue
        pg = int(bt[s, int(pos[r]) // 64])
        assert torch.equal(v[pg, 0, :, int(pos[r]) % 64, :].reshape(-1), got[r, 1024:2048])
        assert torch.equal(v[pg, 1, :, int(pos[r]) % 64, :].reshape(-1), got[r, 2048:])
    written = sum(1 for r in range(M) if int(stream[r]) >= 0)
    assert int((v != 0).any(-1).any(2).any(1).sum()) <= written         # no other (page, token) slot was touched
    assert not poolc[:layer_stride].any()


@pytest.mark.parametrize("rows,d", [(1, 1024), (37, 1024), (5, 4096)])
def test_add_rmsnorm(capi, rows, d):
    x = _rand((rows, d), 13).to(DEV)
    res = _rand((rows, d), 14).to(DEV)
    w = (1 + 0.1 * _rand((d,), 15).float()).to(torch.bfloat16).to(DEV)
    xn = torch.zeros(rows, d, dtype=torch.bfloat16, device=DEV)
    res_in = res.clone()
    call(capi, "t3_op_add_rmsnorm", p(x), p(res), p(w), p(xn), rows, d, C.c_float(1e-5), None)
    torch.cuda.synchronize()
    z = r16(x.float() + res_in.float())
    assert torch.equal(res.float(), z)                                     # residual add is exact
    ref = r16(z * torch.rsqrt((z * z).mean(-1, keepdim=True) + 1e-5)) * w.float()
    assert_bf16_close(xn, ref, ulps=2, what="rmsnorm")
    # without x the residual is left alone
    res2 = res_in.clone()
    call(capi, "t3_op_add_rmsnorm", None, p(res2), p(w), p(xn), rows, d, C.c_float(1e-5), None)
    torch.cuda.synchronize()
    assert torch.equal(res2, res_in)
    zz = res_in.float()
    assert_bf16_close(xn, r16(zz * torch.rsqrt((zz * zz).mean(-1, keepdim=True) + 1e-5)) * w.float(), ulps=2,
                      what="rmsnorm(no add)")


def _fill_cache(pool, layer_stride, layer, bt, stream, k, v, H=16, hd=64):
    view = pool[layer * layer_stride:(layer + 1) * layer_stride].view(-1, 2, H, 64, hd)
    for pos in range(k.shape[0]):
        pg = int(bt[stream, pos // 64])
        view[pg, 0, :, pos % 64, :] = k[pos]
        view[pg, 1, :, pos % 64, :] = v[pos]


def test_attn_decode_matches_softmax_attention(capi):
    H, hd = 16, 64
    ctxs = [1, 2, 63, 64, 65, 200, 511, 513, 1000]
    n = len(ctxs)
    max_pages = 16
    pool, bt, layer_stride = _kv_setup(n, max_pages, 1, seed=3)
    pool.copy_(_rand(tuple(pool.shape), 20, 3.0))            # stale finite data in every page
    ks, vs = [], []
    for s, c in enumerate(ctxs):
        k, v = _rand((c, H, hd), 30 + s).to(DEV), _rand((c, H, hd), 60 + s).to(DEV)
        _fill_cache(pool, layer_stride, 0, bt, s, k, v)
        ks.append(k.float()), vs.append(v.float())
    q = _rand((n, H, hd), 21).to(DEV)
    order = [3, 0, 8, 1, 2, 7, 4, 6, 5]                       # rows need not be stream-ordered
    assert sorted(order) == list(range(n))
    qkv_o = torch.zeros(n, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    qkv_o[:, : H * hd] = q.view(n, -1)[order]
    stream_d = torch.tensor(order, dtype=torch.int32, device=DEV)
    ctx_d = torch.tensor([ctxs[s] for s in order], dtype=torch.int32, device=DEV)
    bt_d = bt.to(DEV)
    assert int(bt_d.max()) < n * max_pages and int(stream_d.max()) < n and int(ctx_d.max()) <= max_pages * 64
    out = torch.zeros(n, H * hd, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_attn_decode", p(qkv_o), p(out), n, H, hd, p(stream_d), p(ctx_d), p(pool), layer_stride, 0,
         p(bt_d), max_pages, None)
    torch.cuda.synchronize()
    for row, s in enumerate(order):
        sc = torch.einsum("hd,shd->hs", q[s].float(), ks[s]) * 0.125
        ref = torch.einsum("hs,shd->hd", torch.softmax(sc, -1), vs[s]).reshape(-1)
        assert_bf16_close(out[row], ref, max_frac=0.05, what=f"attn_decode ctx={ctxs[s]}")
    # the same row alone in a launch: identical bits
    q1 = qkv_o[2:3].contiguous()
    s1 = stream_d[2:3].contiguous()
    c1 = ctx_d[2:3].contiguous()
    one = torch.zeros(1, H * hd, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_attn_decode", p(q1), p(one), 1, H, hd, p(s1), p(c1), p(pool), layer_stride, 0, p(bt_d),
         max_pages, None)
    torch.cuda.synchronize()
    assert torch.equal(one[0], out[2])


@pytest.mark.parametrize("int8", [False, True])
def test_attn_decode_with_output_projection_in_the_launch(capi, int8):
    """t3_op_attn_decode_oproj: attention of (row, head) and that head's share of o_proj in one launch (what full decode
    steps run, four rows per workgroup).  Its 16 head partials, added pairwise (and scaled, for int8 weights), ARE the eight
    split-K slabs t3_op_gemm_slab computes from the stored attention output, bit for bit; so the fold
    (t3_op_head_slab_add_rmsnorm) gives the bits of t3_op_slab_add_rmsnorm over those eight slabs.  A row alone in a
    launch has identical bits; a row without a stream (no attention) projects to zero."""
    H, hd = 16, 64
    ctxs = [1, 64, 65, 200, 513, 1000, 37]
    n = len(ctxs)
    max_pages = 16
    pool, bt, layer_stride = _kv_setup(n, max_pages, 2, seed=5)
    pool.copy_(_rand(tuple(pool.shape), 22, 3.0))
    for s, c in enumerate(ctxs):
        _fill_cache(pool, layer_stride, 1, bt, s, _rand((c, H, hd), 130 + s).to(DEV), _rand((c, H, hd), 160 + s).to(DEV))
    qkv = torch.zeros(n, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    qkv[:, : H * hd] = _rand((n, H * hd), 23).to(DEV)
    stream_d = torch.arange(n, dtype=torch.int32, device=DEV)
    stream_d[6] = -1                                             # a row that gets no attention
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    ctx_d[6] = 0
    bt_d = bt.to(DEV)
    wo = _rand((1024, 1024), 24, 0.02).to(DEV)
    att = torch.zeros(n, H * hd, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_attn_decode", p(qkv), p(att), n, H, hd, p(stream_d), p(ctx_d), p(pool), layer_stride, 1, p(bt_d),
         max_pages, None)
    from chatterbox_vllm_amd._capi import T3GemmDesc
    desc = T3GemmDesc()
    if int8:
        wq = torch.empty(1024 * 1024, dtype=torch.uint8, device=DEV)
        sc = torch.empty(1024, dtype=torch.float32, device=DEV)
        call(capi, "t3_op_quantize_weight", p(wo), p(wq), p(sc), 1024, 1024, None)
        wt, wsc = wq, sc
    else:
        wt, wsc = repack(capi, wo), None
    nsp, n8 = capi.t3_op_oproj_split(), capi.t3_op_slab_split()
    assert nsp == 16 and n8 == 8
    heads = torch.full((nsp, n, 1024), 7.0, dtype=torch.float32, device=DEV)
    call(capi, "t3_op_attn_decode_oproj", p(qkv), p(heads), n, H, hd, p(stream_d), p(ctx_d), p(pool), layer_stride, 1,
         p(bt_d), max_pages, p(wt), int(int8), 4, None, 0, None)
    # one workgroup per (head, row) instead of per (head, 4 rows): the same bits
    heads1 = torch.full((nsp, n, 1024), 5.0, dtype=torch.float32, device=DEV)
    call(capi, "t3_op_attn_decode_oproj", p(qkv), p(heads1), n, H, hd, p(stream_d), p(ctx_d), p(pool), layer_stride, 1,
         p(bt_d), max_pages, p(wt), int(int8), 1, None, 0, None)
    torch.cuda.synchronize()
    assert torch.equal(heads1, heads)
    # the stand-alone projection of the stored attention rows: eight split-K slabs
    slab8 = torch.full((n8, n, 1024), -7.0, dtype=torch.float32, device=DEV)
    desc.a_dev, desc.w_dev, desc.w_scale_dev, desc.slab_dev = att.data_ptr(), wt.data_ptr(), (wsc.data_ptr() if int8 else None), slab8.data_ptr()
    desc.M, desc.N, desc.K, desc.epilogue = n, 1024, 1024, 4
    call(capi, "t3_op_gemm_ex", C.byref(desc), None)
    torch.cuda.synchronize()
    assert torch.equal(att[6], torch.zeros_like(att[6])) and not heads[:, 6].any()
    pairs = heads[0::2] + heads[1::2]
    if int8:
        pairs = pairs * sc[None, None, :]
    assert torch.equal(pairs, slab8), "head partials, added pairwise, are not the stand-alone projection's slabs"
    # and against plain arithmetic: o_proj of the stored attention rows
    w_eff = (W.quantize_int8(wo.cpu())[0].float() * W.quantize_int8(wo.cpu())[1][:, None]).to(DEV) if int8 else wo.float()
    proj = att.float() @ w_eff.T
    assert (slab8.sum(0) - proj).abs().max().item() <= 1e-4 * float(proj.abs().max()) + 1e-6
    # the two folds give the same residual rows and normalised rows
    nw = (1 + 0.1 * _rand((1024,), 26).float()).to(torch.bfloat16).to(DEV)
    res_a = _rand((n, 1024), 25).to(DEV)
    res_b = res_a.clone()
    xn_a = torch.zeros(n, 1024, dtype=torch.bfloat16, device=DEV)
    xn_b = torch.zeros_like(xn_a)
    call(capi, "t3_op_head_slab_add_rmsnorm", p(heads), p(wsc), p(res_a), p(nw), p(xn_a), n, 1024, C.c_float(1e-5), None)
    call(capi, "t3_op_slab_add_rmsnorm", p(slab8), n8, p(res_b), p(nw), p(xn_b), n, 1024, C.c_float(1e-5), None)
    torch.cuda.synchronize()
    assert torch.equal(res_a, res_b) and torch.equal(xn_a, xn_b)
    # one row alone in a launch: identical bits
    one = torch.zeros(nsp, 1, 1024, dtype=torch.float32, device=DEV)
    q1, s1, c1 = qkv[4:5].contiguous(), stream_d[4:5].contiguous(), ctx_d[4:5].contiguous()
    call(capi, "t3_op_attn_decode_oproj", p(q1), p(one), 1, H, hd, p(s1), p(c1), p(pool), layer_stride, 1,
         p(bt_d), max_pages, p(wt), int(int8), 4, None, 0, None)
    torch.cuda.synchronize()
    assert torch.equal(one[:, 0], heads[:, 4])


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("rows", [1, 2, 4])
def test_few_row_split_attention_feeds_o_proj_the_same_bits(capi, rows, int8):
    """t3_op_attn_decode_split_oproj (decode steps of 1..4 rows: eight single-wave workgroups per (row, head), the states
    merged inside the o_proj kernel) writes the eight slabs that t3_op_attn_decode + t3_op_gemm_slab write, bit for bit:
    a request may pass between the two forms as the batch grows and shrinks."""
    H, hd = 16, 64
    ctxs = [1000, 65, 1, 513][:rows]
    max_pages = 16
    pool, bt, layer_stride = _kv_setup(rows, max_pages, 2, seed=6)
    pool.copy_(_rand(tuple(pool.shape), 42, 3.0))
    for s, c in enumerate(ctxs):
        _fill_cache(pool, layer_stride, 1, bt, s, _rand((c, H, hd), 230 + s).to(DEV), _rand((c, H, hd), 260 + s).to(DEV))
    qkv = torch.zeros(rows, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    qkv[:, : H * hd] = _rand((rows, H * hd), 43).to(DEV)
    stream_d = torch.arange(rows, dtype=torch.int32, device=DEV)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    if rows == 4:
        stream_d[2] = -1                                         # a row that gets no attention
        ctx_d[2] = 0
    bt_d = bt.to(DEV)
    wo = _rand((1024, 1024), 44, 0.02).to(DEV)
    if int8:
        wt = torch.empty(1024 * 1024, dtype=torch.uint8, device=DEV)
        sc = torch.empty(1024, dtype=torch.float32, device=DEV)
        call(capi, "t3_op_quantize_weight", p(wo), p(wt), p(sc), 1024, 1024, None)
    else:
        wt, sc = repack(capi, wo), None
    att = torch.zeros(rows, H * hd, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_attn_decode", p(qkv), p(att), rows, H, hd, p(stream_d), p(ctx_d), p(pool), layer_stride, 1, p(bt_d),
         max_pages, None)
    n8 = capi.t3_op_slab_split()
    want = torch.full((n8, rows, 1024), -7.0, dtype=torch.float32, device=DEV)
    from chatterbox_vllm_amd._capi import T3GemmDesc
    desc = T3GemmDesc()
    desc.a_dev, desc.w_dev, desc.w_scale_dev, desc.slab_dev = att.data_ptr(), wt.data_ptr(), (sc.data_ptr() if int8 else None), want.data_ptr()
    desc.M, desc.N, desc.K, desc.epilogue = rows, 1024, 1024, 4
    call(capi, "t3_op_gemm_ex", C.byref(desc), None)
    states = torch.full((rows * capi.t3_op_attn_states_floats(H),), float("nan"), dtype=torch.float32, device=DEV)
    got = torch.full((n8, rows, 1024), 7.0, dtype=torch.float32, device=DEV)
    call(capi, "t3_op_attn_decode_split_oproj", p(qkv), p(states), p(got), rows, H, hd, p(stream_d), p(ctx_d), p(pool),
         layer_stride, 1, p(bt_d), max_pages, p(wt), p(sc), None, 0, None)
    torch.cuda.synchronize()
    assert att.float().abs().max().item() > 0.1
    assert torch.equal(got, want)
    if rows == 4:
        assert not got[:, 2].any()


def test_attn_prefill_is_causal_per_sequence(capi):
    H, hd = 16, 64
    lens = [1, 43, 64, 65, 200]
    starts = [0]
    for L in lens:
        starts.append(starts[-1] + L)
    N = starts[-1]
    qkv = _rand((N, 3 * H * hd), 40).to(DEV)
    out = torch.zeros(N, H * hd, dtype=torch.bfloat16, device=DEV)
    ss_h = torch.tensor(starts, dtype=torch.int32)
    ss_d = ss_h.to(DEV)
    call(capi, "t3_op_attn_prefill", p(qkv), p(out), len(lens), p(ss_d), C.cast(ss_h.data_ptr(), C.POINTER(C.c_int32)),
         H, hd, None)
    torch.cuda.synchronize()
    f = qkv.float()
    for i, L in enumerate(lens):
        s0 = starts[i]
        q = f[s0:s0 + L, :1024].view(L, H, hd)
        k = f[s0:s0 + L, 1024:2048].view(L, H, hd)
        v = f[s0:s0 + L, 2048:].view(L, H, hd)
        sc = torch.einsum("thd,shd->hts", q, k) * 0.125
        mask = torch.triu(torch.ones(L, L, dtype=torch.bool, device=DEV), 1)
        sc = sc.masked_fill(mask[None], float("-inf"))
        ref = torch.einsum("hts,shd->thd", torch.softmax(sc, -1), v).reshape(L, -1)
        assert_bf16_close(out[s0:s0 + L], ref, max_frac=0.05, what=f"attn_prefill len={L}")


@pytest.mark.parametrize("int8", [False, True])
@pytest.mark.parametrize("nsplit_on", [True, False])
@pytest.mark.parametrize("M", [1, 2, 3, 4])
def test_norm_in_the_projection_prologue_has_the_bits_of_the_norm_kernel(capi, M, nsplit_on, int8):
    """Few-row decode form: the projection folds the split-K slabs into the residual rows and normalises them itself.
    Outputs (h of gate_up, q|k|v and the cache rows of the QKV projection) and the updated residual rows must equal,
    bit for bit, what the norm kernel followed by the plain projection gives, over several seeds and magnitudes."""
    H, hd, K, F = 16, 64, 1024, 4096
    S = capi.t3_op_slab_split()
    cfg = T3ModelConfig.english()
    wgu = _rand((2 * F, K), 301, 0.03).to(DEV)
    wqkv = _rand((3 * H * hd, K), 302, 0.03).to(DEV)
    if int8:
        wgu_t, sgu = _quantized(capi, wgu)
        wqkv_t, sqkv = _quantized(capi, wqkv)
    else:
        wgu_t, sgu, wqkv_t, sqkv = repack(capi, wgu), None, repack(capi, wqkv), None
    rope = W.rope_table(cfg, 1000).to(DEV)
    for seed, scale in ((310, 1.0), (311, 30.0), (312, 0.01), (313, 4.0)):
        res = _rand((M, K), seed, scale).to(DEV)
        slab = (torch.randn(S, M, K, generator=torch.Generator().manual_seed(seed + 50)) * scale * 0.3).to(DEV)
        nw = (1.0 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(seed + 60))).to(torch.bfloat16).to(DEV)
        ns = S if nsplit_on else 0
        # the two-kernel way: norm kernel (tile-ordered xn), then the projections
        res_a = res.clone()
        xn = torch.zeros(16 * K, dtype=torch.bfloat16, device=DEV)
        call(capi, "t3_op_slab_add_rmsnorm_tiled", p(slab), ns, p(res_a), p(nw), p(xn), M, K, 1e-5, None)
        h_a = torch.zeros(M, F, dtype=torch.bfloat16, device=DEV)
        d = _desc(a_dev=xn, w_dev=wgu_t, w_scale_dev=sgu, out_dev=h_a, M=M, N=2 * F, K=K, epilogue=2, a_tiled=1)
        call(capi, "t3_op_gemm_ex", C.byref(d), None)
        pool_a, bt, layer_stride = _kv_setup(M, 16, 1, seed=seed)
        pos = torch.randperm(1000, generator=torch.Generator().manual_seed(seed + 70))[:M].to(torch.int32).to(DEV)
        std, btd = torch.arange(M, dtype=torch.int32, device=DEV), bt.to(DEV)
        qkv_a = torch.zeros(M, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
        common = dict(M=M, N=3 * H * hd, K=K, epilogue=3, H=H, hd=hd, layer=0, max_pages=16, rope_dev=rope, pos_dev=pos,
                      stream_dev=std, layer_stride_elems=layer_stride, block_table_dev=btd, w_dev=wqkv_t, w_scale_dev=sqkv)
        d = _desc(a_dev=xn, out_dev=qkv_a, kv_pool_dev=pool_a, a_tiled=1, **common)
        call(capi, "t3_op_gemm_ex", C.byref(d), None)
        # the one-kernel way
        res_b = torch.zeros_like(res)
        fn = dict(fn_slab_dev=slab if ns else None, fn_res_in_dev=res, fn_res_out_dev=res_b if ns else None, fn_norm_w_dev=nw,
                  fn_nsplit=ns, fn_eps=1e-5)
        h_b = torch.zeros_like(h_a)
        d = _desc(w_dev=wgu_t, w_scale_dev=sgu, out_dev=h_b, M=M, N=2 * F, K=K, epilogue=2, **fn)
        call(capi, "t3_op_gemm_ex", C.byref(d), None)
        pool_b = torch.zeros_like(pool_a)
        qkv_b = torch.zeros_like(qkv_a)
        d = _desc(out_dev=qkv_b, kv_pool_dev=pool_b, **common, **fn)
        call(capi, "t3_op_gemm_ex", C.byref(d), None)
        torch.cuda.synchronize()
        assert torch.equal(h_a, h_b), f"gate_up seed {seed}"
        assert torch.equal(qkv_a, qkv_b) and torch.equal(pool_a, pool_b), f"qkv seed {seed}"
        if ns:
            assert torch.equal(res_a, res_b), f"residual rows seed {seed}"
        assert h_a.float().abs().sum() > 0 and qkv_a.float().abs().sum() > 0


def test_attn_decode_with_shared_prefix_page_has_the_bits_of_private_copies(capi):
    """Positions [0, 34) of some rows live in one shared page; the row's own page 0 holds garbage there."""
    H, hd, NP = 16, 64, 34
    ctxs = [35, 63, 64, 65, 200, 513, 1000, 34, 100]
    shared = [True, True, False, True, True, False, True, True, True]
    n, max_pages = len(ctxs), 16
    pool, bt, layer_stride = _kv_setup(n + 1, max_pages, 2, seed=5)     # stream n owns the prefix page
    pool.copy_(_rand(tuple(pool.shape), 22, 3.0))
    pk, pv = _rand((NP, H, hd), 90).to(DEV), _rand((NP, H, hd), 91).to(DEV)
    for layer in (0, 1):
        _fill_cache(pool, layer_stride, layer, bt, n, pk, pv)
    private = pool.clone()
    for s, c in enumerate(ctxs):
        k, v = _rand((c, H, hd), 130 + s).to(DEV), _rand((c, H, hd), 160 + s).to(DEV)
        if shared[s]:
            k[:NP], v[:NP] = pk, pv
        _fill_cache(private, layer_stride, 1, bt, s, k, v)
        if shared[s]:                                                    # own copy of the prefix positions: garbage
            k, v = k.clone(), v.clone()
            k[:NP], v[:NP] = _rand((NP, H, hd), 7).to(DEV) * 50, _rand((NP, H, hd), 8).to(DEV) * 50
        _fill_cache(pool, layer_stride, 1, bt, s, k, v)
    qkv = torch.zeros(n, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    qkv[:, : H * hd] = _rand((n, H * hd), 23).to(DEV)
    stream_d = torch.arange(n, dtype=torch.int32, device=DEV)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    bt_d = bt.to(DEV)
    pfx_d = torch.tensor([int(bt[n, 0]) if sh else -1 for sh in shared], dtype=torch.int32, device=DEV)
    want = torch.zeros(n, H * hd, dtype=torch.bfloat16, device=DEV)
    got = torch.zeros_like(want)
    call(capi, "t3_op_attn_decode", p(qkv), p(want), n, H, hd, p(stream_d), p(ctx_d), p(private), layer_stride, 1,
         p(bt_d), max_pages, None)
    call(capi, "t3_op_attn_decode_prefix", p(qkv), p(got), n, H, hd, p(stream_d), p(ctx_d), p(pool), layer_stride, 1,
         p(bt_d), max_pages, p(pfx_d), NP, None)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_attn_prefill_with_shared_prefix_page_has_the_bits_of_whole_sequences(capi):
    """Sequences whose first 34 keys come from a pool page equal the tail rows of the whole sequences."""
    H, hd, NP = 16, 64, 34
    D = H * hd
    tails = [1, 29, 30, 31, 100, 230]                                   # rows after the prefix; 34 + 30 = one key tile
    n_pages, layer = 3, 1
    layer_stride = n_pages * 2 * H * 64 * hd
    pool = (_rand((2 * layer_stride,), 24, 3.0)).to(DEV)
    page = 2
    pre = _rand((NP, 3 * D), 41).to(DEV)                                # q|k|v rows of the prefix positions
    view = pool[layer * layer_stride:(layer + 1) * layer_stride].view(-1, 2, H, 64, hd)
    view[page, 0, :, :NP, :] = pre[:, D:2 * D].view(NP, H, hd).transpose(0, 1)
    view[page, 1, :, :NP, :] = pre[:, 2 * D:].view(NP, H, hd).transpose(0, 1)
    rows = [_rand((t, 3 * D), 50 + i).to(DEV) for i, t in enumerate(tails)]
    # whole sequences (prefix rows + tail rows), and one sequence that has no prefix at all
    whole = torch.cat([torch.cat([pre, r]) for r in rows] + [rows[4]])
    ws = [0]
    for t in tails:
        ws.append(ws[-1] + NP + t)
    ws.append(ws[-1] + tails[4])
    want = torch.zeros(whole.shape[0], D, dtype=torch.bfloat16, device=DEV)
    ws_h = torch.tensor(ws, dtype=torch.int32)
    ws_d = ws_h.to(DEV)
    call(capi, "t3_op_attn_prefill", p(whole), p(want), len(ws) - 1, p(ws_d), C.cast(ws_h.data_ptr(), C.POINTER(C.c_int32)),
         H, hd, None)
    part = torch.cat(rows + [rows[4]])
    ps = [0]
    for t in tails + [tails[4]]:
        ps.append(ps[-1] + t)
    got = torch.zeros(part.shape[0], D, dtype=torch.bfloat16, device=DEV)
    ps_h = torch.tensor(ps, dtype=torch.int32)
    ps_d = ps_h.to(DEV)
    pfx_d = torch.tensor([page] * len(tails) + [-1], dtype=torch.int32, device=DEV)
    call(capi, "t3_op_attn_prefill_prefix", p(part), p(got), len(ps) - 1, p(ps_d), C.cast(ps_h.data_ptr(), C.POINTER(C.c_int32)),
         H, hd, p(pfx_d), NP, p(pool), layer_stride, layer, None)
    torch.cuda.synchronize()
    for i in range(len(tails)):
        assert torch.equal(got[ps[i]:ps[i + 1]], want[ws[i] + NP:ws[i + 1]]), f"sequence {i}"
    assert torch.equal(got[ps[-2]:], want[ws[-2]:])


def test_sampler_cfg_penalty_argmax_is_exact(capi):
    from oracle.t3_oracle import T3Oracle
    n, vocab, ldl = 6, 8194, 8208
    logits = _rand((2 * n, ldl), 50, 2.0).to(DEV)
    logits[0, 77] = logits[0, 5000] = 30.0                    # a tie: lowest index must win
    logits[n, 77] = logits[n, 5000] = 1.0
    row_c = torch.arange(n, dtype=torch.int32)
    row_u = torch.arange(n, 2 * n, dtype=torch.int32)
    row_u[3] = -1                                             # one request without CFG
    seen = torch.zeros(n, 9216, dtype=torch.uint8)
    g = torch.Generator().manual_seed(51)
    for i in range(n):
        seen[i, torch.randint(0, 9194, (300,), generator=g)] = 1
    seen[0, 1077] = 0
    seen[0, 6000] = 0
    pen = torch.tensor([2.0, 2.0, 1.0, 2.0, 1.3, 2.0])
    toks = torch.zeros(n, dtype=torch.int32, device=DEV)
    outl = torch.zeros(n, vocab + 1000, device=DEV)
    seen_d, row_c_d, row_u_d, pen_d = seen.to(DEV), row_c.to(DEV), row_u.to(DEV), pen.to(DEV)
    cfg_scale = 3.0
    call(capi, "t3_op_sample_greedy", p(logits), ldl, vocab, p(row_c_d), p(row_u_d), n, C.c_float(cfg_scale),
         p(pen_d), p(seen_d), 9216, p(toks), p(outl), None)
    torch.cuda.synchronize()
    o = T3Oracle.__new__(T3Oracle)
    o.bf16, o.cfg_scale = True, cfg_scale
    lc = logits.float().cpu()
    for i in range(n):
        u = lc[int(row_u[i]), :vocab] if row_u[i] >= 0 else None
        full = o.cfg_logits(lc[i, :vocab], u)
        ref = T3Oracle.apply_repetition_penalty(full, seen[i].nonzero().flatten().tolist(), float(pen[i]))
        got = outl[i].cpu()
        assert torch.equal(got, ref), f"request {i}: penalised logits differ"
        assert int(toks[i]) == T3Oracle.greedy(ref)
        assert int(seen_d[i, int(toks[i])]) == 1              # chosen column joins the seen set
    assert int(toks[0]) == 1077


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("M", [1, 30, 64, 100, 300])
@pytest.mark.parametrize("K", [1024, 4096])
def test_split_k_slabs_then_residual_norm(capi, M, K, form):
    """Decode-path pair: t3_op_gemm_slab leaves fp32 partial sums, t3_op_slab_add_rmsnorm folds them
    into the residual stream and normalises.  Equals gemm + residual add + rmsnorm of the contract."""
    N = 1024
    S = capi.t3_op_slab_split()
    a = _rand((M, K), 70).to(DEV)
    w = _rand((N, K), 71, 0.02).to(DEV)
    wt = repack(capi, w)
    slab = torch.full((S, M, N), float("nan"), device=DEV)
    call(capi, form + "_slab", p(a), p(wt), p(slab), M, N, K, None)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().T
    assert torch.isfinite(slab).all()
    assert torch.allclose(slab.sum(0), ref, rtol=1e-4, atol=1e-4)
    res = _rand((M, N), 72).to(DEV)
    res0 = res.clone()
    g = (1 + 0.1 * _rand((N,), 73).float()).to(torch.bfloat16).to(DEV)
    xn = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_slab_add_rmsnorm", p(slab), S, p(res), p(g), p(xn), M, N, C.c_float(1e-5), None)
    torch.cuda.synchronize()
    z = r16(r16(ref) + res0.float())
    assert_bf16_close(res, r16(ref) + res0.float(), mag=torch.maximum(ref.abs(), res0.float().abs()), ulps=1.5,
                      what="slab residual")
    zz = res.float()                                        # normalise what the kernel actually stored
    assert_bf16_close(xn, r16(zz * torch.rsqrt((zz * zz).mean(-1, keepdim=True) + 1e-5)) * g.float(), ulps=2,
                      what="slab rmsnorm")
    # rows do not depend on the batch
    if M > 1:
        slab1 = torch.zeros(S, 1, N, device=DEV)
        a1 = a[:1].contiguous()
        call(capi, form + "_slab", p(a1), p(wt), p(slab1), 1, N, K, None)
        torch.cuda.synchronize()
        assert torch.equal(slab1[:, 0], slab[:, 0])
    # nsplit == 0 leaves the residual alone
    res2 = res0.clone()
    call(capi, "t3_op_slab_add_rmsnorm", None, 0, p(res2), p(g), p(xn), M, N, C.c_float(1e-5), None)
    torch.cuda.synchronize()
    assert torch.equal(res2, res0)


# ---- int8 weights (W8A16, one scale per output column) ---------------------------------------------
def _quantized(capi, w):
    """Device quantisation of W [N,K] -> (pair-tile bytes, scales), checked against the host rule."""
    N, K = w.shape
    wq = torch.zeros(N * K, dtype=torch.uint8, device=DEV)
    sc = torch.zeros(N, dtype=torch.float32, device=DEV)
    call(capi, "t3_op_quantize_weight", p(w), p(wq), p(sc), N, K, None)
    torch.cuda.synchronize()
    return wq, sc


def _untile_int8(wq, N, K):
    """pair-tile bytes -> q [N, K] (int): byte (lane*16 + t*8 + j) of tile (nb, kp) is
    u8[nb*16 + lane%16][kp*64 + t*32 + 8*(lane//16) + j]."""
    t = wq.cpu().view(N // 16, K // 64, 4, 16, 2, 8).to(torch.int32) - 128      # nb, kp, lane//16, lane%16, t, j
    return t.permute(0, 3, 1, 4, 2, 5).reshape(N, K)                            # nb, nr, kp, t, g, j


@pytest.mark.parametrize("N,K", [(1024, 1024), (1024, 4096), (3072, 1024)])
def test_quantize_weight_matches_host_rule(capi, N, K):
    w = _rand((N, K), 80, 0.02)
    w[5] = 0                                                   # an all-zero row keeps scale 1
    w[7, 3] = 0.5                                              # an outlier sets its row's scale
    wd = w.to(DEV)
    wq, sc = _quantized(capi, wd)
    q_ref, s_ref = W.quantize_int8(w)
    assert torch.equal(sc.cpu(), s_ref)
    assert torch.equal(_untile_int8(wq, N, K), q_ref.to(torch.int32))


def _desc(**kw):
    from chatterbox_vllm_amd._capi import T3GemmDesc
    d = T3GemmDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


@pytest.mark.parametrize("prompt", [0, 1])
@pytest.mark.parametrize("M", [1, 33, 64, 200])
def test_int8_projections_match_dequantised_weights(capi, M, prompt):
    """Every int8 projection form against fp32 matmul with scale * q: same contract as the bf16 forms."""
    H, hd, K, F = 16, 64, 1024, 4096
    a = _rand((M, K), 81).to(DEV)
    # store / residual (decode form only), SiLU, split-K slabs
    w = _rand((2 * F, K), 82, 0.05).to(DEV)
    wq, sc = _quantized(capi, w)
    wdq = (W.quantize_int8(w.cpu())[0].float() * sc.cpu()[:, None]).to(DEV)
    ref = a.float() @ wdq.T
    if not prompt:
        out = torch.zeros(M, 2 * F, dtype=torch.bfloat16, device=DEV)
        d = _desc(a_dev=a, w_dev=wq, w_scale_dev=sc, out_dev=out, M=M, N=2 * F, K=K, ldc=2 * F, epilogue=0)
        call(capi, "t3_op_gemm_ex", C.byref(d), None)
        torch.cuda.synchronize()
        assert_bf16_close(out, ref, what="int8 store")
    h = torch.zeros(M, F, dtype=torch.bfloat16, device=DEV)
    d = _desc(a_dev=a, w_dev=wq, w_scale_dev=sc, out_dev=h, M=M, N=2 * F, K=K, epilogue=2, prompt_form=prompt)
    call(capi, "t3_op_gemm_ex", C.byref(d), None)
    torch.cuda.synchronize()
    gu = r16(ref)
    g, u = gu[:, :F], gu[:, F:]
    assert_bf16_close(h, r16(g / (1 + torch.exp(-g))) * u, ulps=3, max_frac=0.03, what="int8 silu")
    S = capi.t3_op_slab_split()
    for Kd in (1024, 4096):
        ad = _rand((M, Kd), 83).to(DEV)
        wd_ = _rand((1024, Kd), 84, 0.02).to(DEV)
        wdq_, scd = _quantized(capi, wd_)
        deq = (W.quantize_int8(wd_.cpu())[0].float() * scd.cpu()[:, None]).to(DEV)
        slab = torch.full((S, M, 1024), float("nan"), device=DEV)
        d = _desc(a_dev=ad, w_dev=wdq_, w_scale_dev=scd, slab_dev=slab, M=M, N=1024, K=Kd, epilogue=4, prompt_form=prompt)
        call(capi, "t3_op_gemm_ex", C.byref(d), None)
        torch.cuda.synchronize()
        assert torch.allclose(slab.sum(0), ad.float() @ deq.T, rtol=1e-4, atol=1e-4), f"int8 slab K={Kd}"
    # q|k|v + RoPE + cache write
    cfg = T3ModelConfig.english()
    wqkv = _rand((3 * H * hd, K), 85, 0.03).to(DEV)
    wq3, sc3 = _quantized(capi, wqkv)
    deq3 = (W.quantize_int8(wqkv.cpu())[0].float() * sc3.cpu()[:, None]).to(DEV)
    rope = W.rope_table(cfg, 1000).to(DEV)
    pool, bt, layer_stride = _kv_setup(1, 16, 1)
    pos = torch.randperm(1000, generator=torch.Generator().manual_seed(86))[:M].to(torch.int32)
    posd, std, btd = pos.to(DEV), torch.zeros(M, dtype=torch.int32, device=DEV), bt.to(DEV)
    qkv = torch.zeros(M, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
    d = _desc(a_dev=a, w_dev=wq3, w_scale_dev=sc3, out_dev=qkv, M=M, N=3 * H * hd, K=K, epilogue=3, prompt_form=prompt,
              H=H, hd=hd, layer=0, max_pages=16, rope_dev=rope, pos_dev=posd, stream_dev=std, kv_pool_dev=pool,
              layer_stride_elems=layer_stride, block_table_dev=btd)
    call(capi, "t3_op_gemm_ex", C.byref(d), None)
    torch.cuda.synchronize()
    r3 = r16(a.float() @ deq3.T).cpu()
    cs = rope.float().cpu()[pos.long()]
    cos, sin = cs[:, None, :32], cs[:, None, 32:]
    for lo, name in ((0, "q"), (1024, "k")):
        t = r3[:, lo:lo + 1024].view(M, H, hd)
        x, y = t[..., :32], t[..., 32:]
        want = torch.cat([x * cos - y * sin, y * cos + x * sin], -1).reshape(M, H * hd)
        mag = torch.cat([torch.maximum(x.abs(), y.abs())] * 2, -1).reshape(M, H * hd)
        assert_bf16_close(qkv[:, lo:lo + 1024], want, mag=mag, ulps=2.5, max_frac=0.04, what="int8 " + name)
    assert_bf16_close(qkv[:, 2048:], r3[:, 2048:], what="int8 v")
    v = pool.cpu().view(-1, 2, H, 64, hd)
    for r in range(M):
        pg = int(bt[0, int(pos[r]) // 64])
        assert torch.equal(v[pg, 0, :, int(pos[r]) % 64, :].reshape(-1), qkv[r, 1024:2048].cpu())
        assert torch.equal(v[pg, 1, :, int(pos[r]) % 64, :].reshape(-1), qkv[r, 2048:].cpu())


# ---- activation tile order (the decode step's intermediate layout) ---------------------------------------
def _tile_act(a):
    """[M, K] -> activation tile order: rows padded to 16; element (r, k) at
    ((r//16)*(K//32) + k//32)*512 + ((k%32)//8*16 + r%16)*8 + k%8."""
    M, K = a.shape
    Mp = (M + 15) // 16 * 16
    pad = torch.zeros(Mp, K, dtype=a.dtype, device=a.device)
    pad[:M] = a
    return pad.view(Mp // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def _untile_act(t, M, K):
    Mp = (M + 15) // 16 * 16
    return t.view(Mp // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(Mp, K)[:M]


@pytest.mark.parametrize("M", [1, 20, 52, 64, 100])
def test_tile_ordered_activations_give_the_same_bits(capi, M):
    """Every decode projection fed tile-ordered activations (and the SiLU form writing them) returns exactly the
    bits of the row-major call: the layout changes addresses, not arithmetic."""
    K, F = 1024, 4096
    a = _rand((M, K), 90).to(DEV)
    at = _tile_act(a)
    w = _rand((2 * F, K), 91, 0.05).to(DEV)
    wt = repack(capi, w)
    Mp = (M + 15) // 16 * 16
    # store (head-shaped) and SiLU
    for epi, N_out in ((0, 2 * F), (2, F)):
        outs = []
        for tiled in (0, 1):
            out = torch.full((Mp * N_out,), 3.0, dtype=torch.bfloat16, device=DEV)
            d = _desc(a_dev=at if tiled else a, w_dev=wt, out_dev=out, M=M, N=2 * F, K=K, ldc=2 * F, epilogue=epi,
                      a_tiled=tiled, out_tiled=tiled if epi == 2 else 0)
            call(capi, "t3_op_gemm_ex", C.byref(d), None)
            torch.cuda.synchronize()
            outs.append(out)
        plain = outs[0][: M * N_out].view(M, N_out)
        got = _untile_act(outs[1], M, N_out) if epi == 2 else outs[1][: M * N_out].view(M, N_out)
        assert torch.equal(got, plain), f"epilogue {epi}"
    # split-K slabs over K = 4096 (down_proj reads the tile-ordered h) and the norm that writes tile order
    h = _rand((M, F), 92).to(DEV)
    wd = _rand((1024, F), 93, 0.02).to(DEV)
    wdt = repack(capi, wd)
    S = capi.t3_op_slab_split()
    slabs = []
    for tiled in (0, 1):
        slab = torch.zeros(S, M, 1024, device=DEV)
        hin = _tile_act(h) if tiled else h
        d = _desc(a_dev=hin, w_dev=wdt, slab_dev=slab, M=M, N=1024, K=F, epilogue=4, a_tiled=tiled)
        call(capi, "t3_op_gemm_ex", C.byref(d), None)
        torch.cuda.synchronize()
        slabs.append(slab)
    assert torch.equal(slabs[0], slabs[1])
    res = _rand((M, 1024), 94).to(DEV)
    g = (1 + 0.1 * _rand((1024,), 95).float()).to(torch.bfloat16).to(DEV)
    r1, r2 = res.clone(), res.clone()
    xn1 = torch.zeros(M, 1024, dtype=torch.bfloat16, device=DEV)
    xn2 = torch.zeros(Mp * 1024, dtype=torch.bfloat16, device=DEV)
    call(capi, "t3_op_slab_add_rmsnorm", p(slabs[0]), S, p(r1), p(g), p(xn1), M, 1024, C.c_float(1e-5), None)
    call(capi, "t3_op_slab_add_rmsnorm_tiled", p(slabs[0]), S, p(r2), p(g), p(xn2), M, 1024, C.c_float(1e-5), None)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2) and torch.equal(_untile_act(xn2, M, 1024), xn1)


@pytest.mark.parametrize("M", [1, 130, 512, 513, 1300])
def test_prompt_gemm_forms_give_the_same_bits(capi, M):
    """The row-walking prompt kernel (one workgroup per weight tile, all rows; every weight byte fetched once) against the
    kernel with one workgroup per 128-row tile: gate_up + SiLU, split-K slabs at K = 1024 and 4096, QKV + RoPE + cache
    write, bf16 weights; outputs and KV pool must be bit-identical (both accumulate an output in k order in one accumulator)."""
    cfg = T3ModelConfig.english()
    H, hd, K, F = 16, 64, 1024, 4096
    a = _rand((M, K), 80).to(DEV)
    wgu = repack(capi, _rand((2 * F, K), 81, 0.05).to(DEV))
    wo = repack(capi, _rand((1024, K), 82, 0.05).to(DEV))
    a4 = _rand((M, 4096), 83).to(DEV)
    wd = repack(capi, _rand((1024, 4096), 84, 0.03).to(DEV))
    wqkv = repack(capi, _rand((3 * H * hd, K), 85, 0.03).to(DEV))
    rope = W.rope_table(cfg, 1000).to(DEV)
    n_streams, max_pages = 8, 16
    stream = (torch.arange(M) % n_streams).to(torch.int32)
    pos = (torch.arange(M) // n_streams).to(torch.int32)
    S = capi.t3_op_slab_split()
    out = {}
    try:
        for walk in (1, 0):
            call(capi, "t3_op_set_prompt_gemm_form", walk)
            h = torch.zeros(M, F, dtype=torch.bfloat16, device=DEV)
            call(capi, "t3_op_gemm_prompt_silu", p(a), p(wgu), p(h), M, F, K, None)
            s1 = torch.zeros(S, M, 1024, dtype=torch.float32, device=DEV)
            call(capi, "t3_op_gemm_prompt_slab", p(a), p(wo), C.cast(p(s1), C.POINTER(C.c_float)), M, 1024, K, None)
            s4 = torch.zeros(S, M, 1024, dtype=torch.float32, device=DEV)
            call(capi, "t3_op_gemm_prompt_slab", p(a4), p(wd), C.cast(p(s4), C.POINTER(C.c_float)), M, 1024, 4096, None)
            pool, bt, layer_stride = _kv_setup(n_streams, max_pages, 1, seed=3)
            qkv = torch.zeros(M, 3 * H * hd, dtype=torch.bfloat16, device=DEV)
            btd, posd, std = bt.to(DEV), pos.to(DEV), stream.to(DEV)
            call(capi, "t3_op_gemm_prompt_qkv_rope", p(a), p(wqkv), p(qkv), M, H, hd, K, p(rope), p(posd), p(std), p(pool),
                 layer_stride, 0, p(btd), max_pages, None)
            torch.cuda.synchronize()
            out[walk] = (h, s1, s4, qkv, pool)
    finally:
        call(capi, "t3_op_set_prompt_gemm_form", 0)
    for x, y, what in zip(out[1], out[0], ("gate_up", "o slab", "down slab", "qkv", "kv pool")):
        assert torch.equal(x, y), f"{what}: the two prompt forms differ at M = {M}"
    assert out[1][0].abs().sum() > 0 and out[1][4].abs().sum() > 0
