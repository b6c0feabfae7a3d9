"""Helpers shared by the GPU parity tests."""
import ctypes as C

import torch

from chatterbox_vllm_amd import _capi


def ulp_distance(a_bf16: torch.Tensor, b_bf16: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 units-in-the-last-place (order-preserving integer map of the bit patterns)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a_bf16) - key(b_bf16)).abs()


def assert_bf16_close(got: torch.Tensor, ref_fp32: torch.Tensor, max_ulp: int = 1, max_frac: float = 0.02, what=""):
    """`got` (bf16) must equal bf16(ref) except for roundings flipped by fp32 summation order:
    at most `max_frac` of the elements may differ, each by at most `max_ulp` bf16 ulp."""
    ref = ref_fp32.to(torch.bfloat16)
    d = ulp_distance(got.cpu(), ref.cpu())
    frac = (d > 0).float().mean().item()
    assert d.max().item() <= max_ulp, f"{what}: max ulp distance {d.max().item()} > {max_ulp}"
    assert frac <= max_frac, f"{what}: {frac:.4f} of elements differ (> {max_frac})"


def call(lib, name, *args):
    _capi.check(lib, getattr(lib, name)(*args))


def p(t):
    return _capi.ptr(t)


def repack(lib, w):
    """W [N,K] bf16 cuda -> tile order."""
    wt = torch.empty_like(w)
    call(lib, "t3_op_repack_weight", p(w), p(wt), w.shape[0], w.shape[1], None)
    return wt


def r16(x):
    return x.to(torch.bfloat16).to(torch.float32)
