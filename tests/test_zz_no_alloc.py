"""t3_op_attn_prefill_static: the prefill attention entry point that allocates nothing (its tile list is built on the
device into an array inside the library's code object) against t3_op_attn_prefill, which stages the same list
through a buffer it allocates.  Same kernel, same tiles: the outputs must be bit-identical, call after call, also when
a later call is smaller than the one before it (the array keeps the earlier call's entries behind the new ones)."""
import ctypes as C

import pytest
import torch

from chatterbox_vllm_amd import _capi
from _util import call, p

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, HD = 16, 64


def _case(lens, seed):
    g = torch.Generator().manual_seed(seed)
    rows = sum(lens)
    qkv = (torch.randn(rows, 3 * H * HD, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    ss_h = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    return qkv, ss_h, ss_h.to(DEV)


def _both(capi, lens, seed):
    qkv, ss_h, ss_d = _case(lens, seed)
    want = torch.full((sum(lens), H * HD), 7.0, dtype=torch.bfloat16, device=DEV)
    got = torch.full_like(want, -3.0)
    host = C.cast(ss_h.data_ptr(), C.POINTER(C.c_int32))
    call(capi, "t3_op_attn_prefill", p(qkv), p(want), len(lens), p(ss_d), host, H, HD, None)
    call(capi, "t3_op_attn_prefill_static", p(qkv), p(got), len(lens), p(ss_d), host, H, HD, None)
    torch.cuda.synchronize()
    assert torch.equal(got, want), f"lens={lens}: the static-scratch form differs from t3_op_attn_prefill"
    return got


def test_static_scratch_prefill_attention_equals_the_allocating_entry_point(capi):
    big = _both(capi, [300, 64, 1, 129, 65, 519], 1)           # 19 tiles, ragged, tile edges on both sides
    small = _both(capi, [70, 3], 2)                            # second call, smaller: 3 tiles, old entries stay behind them
    again = _both(capi, [300, 64, 1, 129, 65, 519], 1)
    assert torch.equal(again, big) and small.shape[0] == 73
    _both(capi, [1] * 40, 3)                                   # many one-row sequences
    # a side stream (the kernels are only enqueued on the stream given)
    st = torch.cuda.Stream()
    qkv, ss_h, ss_d = _case([200, 31], 4)
    want = torch.empty(231, H * HD, dtype=torch.bfloat16, device=DEV)
    got = torch.empty_like(want)
    host = C.cast(ss_h.data_ptr(), C.POINTER(C.c_int32))
    torch.cuda.synchronize()
    call(capi, "t3_op_attn_prefill", p(qkv), p(want), 2, p(ss_d), host, H, HD, None)
    call(capi, "t3_op_attn_prefill_static", p(qkv), p(got), 2, p(ss_d), host, H, HD, C.c_void_p(st.cuda_stream))
    st.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_static_scratch_prefill_attention_refuses_what_does_not_fit(capi):
    qkv, ss_h, ss_d = _case([64, 64], 5)
    out = torch.empty(128, H * HD, dtype=torch.bfloat16, device=DEV)
    too_many_tiles = torch.tensor([0, 64 * 4097], dtype=torch.int32)
    with pytest.raises(_capi.T3Error, match="4096"):
        call(capi, "t3_op_attn_prefill_static", p(qkv), p(out), 1, p(ss_d),
             C.cast(too_many_tiles.data_ptr(), C.POINTER(C.c_int32)), H, HD, None)
    many = torch.arange(0, 1026, dtype=torch.int32)
    with pytest.raises(_capi.T3Error, match="1024"):
        call(capi, "t3_op_attn_prefill_static", p(qkv), p(out), 1025, p(ss_d),
             C.cast(many.data_ptr(), C.POINTER(C.c_int32)), H, HD, None)
    torch.cuda.synchronize()
