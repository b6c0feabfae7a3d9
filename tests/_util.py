"""Helpers shared by the GPU parity tests."""
import ctypes as C

import torch

from chatterbox_vllm_amd import _capi


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """Size of one bf16 unit-in-the-last-place at magnitude |x| (8 significant bits)."""
    x = x.abs().float().clamp_min(2.0 ** -100)
    return torch.exp2(torch.floor(torch.log2(x)) - 7)


def assert_bf16_close(got: torch.Tensor, ref_fp32: torch.Tensor, mag: torch.Tensor = None, ulps: float = 1.0,
                      max_frac: float = 0.02, what: str = ""):
    """`got` (bf16) must equal bf16(ref) except where the fp32 summation order flips a rounding.

    Every element must lie within `ulps` bf16 ulp of bf16(ref), the ulp taken at the larger of
    |ref| and `mag` (the magnitude of the last op's operands: a sum of two rounded terms that
    cancel inherits the ulp of the terms, not of the small result), plus an absolute floor for
    dot products that cancel to ~0.  At most `max_frac` of the elements may differ at all.
    """
    got = got.detach().float().cpu()
    ref32 = ref_fp32.detach().float().cpu()
    ref = ref32.to(torch.bfloat16).float()
    m = ref.abs() if mag is None else torch.maximum(ref.abs(), mag.detach().float().cpu().abs())
    atol = 2.0 ** -10 * ref32.pow(2).mean().sqrt().item()
    err = (got - ref).abs()
    bound = ulps * bf16_ulp(m) + atol
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())} elements off by more than {ulps} ulp; worst error "
                           f"{err[bad].max().item():.3e} vs bound {bound[bad][err[bad].argmax()].item():.3e}")
    frac = (got != ref).float().mean().item()
    assert frac <= max_frac, f"{what}: {frac:.4f} of elements differ from the checker (> {max_frac})"


def call(lib, name, *args):
    _capi.check(lib, getattr(lib, name)(*args))


def p(t):
    """Address of a tensor.  The caller must keep `t` alive until the kernel has run: a pointer
    taken from a temporary (`p(x.to('cuda'))`) dangles as soon as the allocator reuses the block."""
    return _capi.ptr(t)


def repack(lib, w):
    """W [N,K] bf16 cuda -> tile order."""
    wt = torch.empty_like(w)
    call(lib, "t3_op_repack_weight", p(w), p(wt), w.shape[0], w.shape[1], None)
    return wt


def r16(x):
    return x.to(torch.bfloat16).to(torch.float32)
