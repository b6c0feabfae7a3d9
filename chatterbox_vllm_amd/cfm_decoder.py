"""Flow-matching decoder of S3Gen's flow stage on the GPU: the host-side mirror of the reference's
``CausalConditionalCFM`` (src/chatterbox_vllm/models/s3gen/flow_matching.py:198-228) with its ``ConditionalDecoder``
estimator (decoder.py:100-317), over the C ABI of include/s3cfm_hip.h (entry points in ``lib/libs3hift.so``).

``CausalConditionalCFM(state_dict)(mu=, mask=, spks=, cond=, n_timesteps=)`` takes and returns what the reference module's
``forward`` does, so it is the ``decoder`` that ``S3Token2Wav`` calls.  fp32 on the handle's GPU; torch carries device memory
only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import cfm_weights as CW
from .hift import HiftError, check, load, _P, _ptr, _stream_ptr

_SIGNATURES = {
    "s3cfm_create": ([C.c_int32, C.c_int32, C.POINTER(_P)], C.c_int),
    "s3cfm_create_batched": ([C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)], C.c_int),
    "s3cfm_destroy": ([_P], C.c_int),
    "s3cfm_load_tensor": ([_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32], C.c_int),
    "s3cfm_finalize": ([_P], C.c_int),
    "s3cfm_set_split_bf16": ([_P, C.c_int32], C.c_int),
    "s3cfm_estimator": ([_P, _P, _P, _P, _P, C.c_float, C.c_int32, _P, _P], C.c_int),
    "s3cfm_solve": ([_P, _P, _P, _P, _P, C.c_int32, C.POINTER(C.c_float), C.c_int32, C.c_double, _P, _P], C.c_int),
    "s3cfm_solve_batch": ([_P, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _P, _P, _P, _P, C.POINTER(C.c_float), C.c_int32, C.c_double, _P,
                           _P], C.c_int),
    "s3cfm_op_linear": ([_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32,
                         _P, _P, C.c_int32, C.POINTER(C.c_float), _P], C.c_int),
    "s3cfm_op_linear_split": ([_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32,
                               _P, _P, C.c_int32, C.POINTER(C.c_float), _P], C.c_int),
    "s3cfm_op_attention": ([_P, _P, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, _P, C.c_int32, C.POINTER(C.c_float), _P], C.c_int),
    "s3cfm_op_attention_split": ([_P, _P, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, _P, C.c_int32, C.POINTER(C.c_float),
                                  C.POINTER(C.c_float), _P], C.c_int),
}

VT_LD = 576                  # row stride of the frame-major V tensor (512 channels + padding)
MAX_BATCH = 8                # S3CFM_MAX_BATCH


def _lib() -> C.CDLL:
    lib = load()
    if not getattr(lib, "_s3cfm_declared", False):
        for name, (argtypes, restype) in _SIGNATURES.items():
            if not hasattr(lib, name):
                raise HiftError(f"libs3hift.so does not export {name}: rebuild it (make -C chatterbox_vllm_amd/csrc_s3gen)")
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, restype
        lib._s3cfm_declared = True
    return lib


class CausalConditionalCFM:
    """``state_dict`` holds ``S3Token2Wav.flow.state_dict()`` (only the ``decoder.estimator.*`` entries are taken).

    ``rand_noise`` is the fixed noise the reference draws once at construction (``torch.randn([1, 80, 50 * 300])``,
    flow_matching.py:201) and slices per utterance; it is drawn the same way here, from torch's global CPU generator
    unless one is passed.

    ``max_batch`` > 1 gives the workspace room for that many utterances per ``forward_batch`` call (no counterpart in the
    reference, which runs one utterance per call).

    ``split_bf16`` (no counterpart in the reference either; off by default): the attention launches hold every fp32 operand
    as three bf16 values and take six partial products per product on the bf16 matrix instruction, fp32 accumulation; the
    results stay within the fp32 form's distance of an fp64 computation (``set_split_bf16`` switches it later)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], max_frames: int = 3000, device: int = 0,
                 inference_cfg_rate: float = CW.CFG_RATE, generator: Optional[torch.Generator] = None, max_batch: int = 1,
                 split_bf16=False):
        self._lib = _lib()
        self.device = torch.device("cuda", device)
        self.max_frames = max_frames
        self.max_batch = max_batch
        self.inference_cfg_rate = inference_cfg_rate
        self.rand_noise = torch.randn([1, CW.MEL, CW.NOISE_FRAMES], generator=generator)
        self._noise_dev = None
        h = _P()
        check(self._lib, self._lib.s3cfm_create_batched(device, max_frames, max_batch, C.byref(h)))
        self._h = h
        try:
            for name, _ in CW.tensor_inventory():
                if name not in state_dict:
                    raise KeyError(f"state dict lacks {name}")
                t = state_dict[name].detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                check(self._lib, self._lib.s3cfm_load_tensor(h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            check(self._lib, self._lib.s3cfm_finalize(h))
            self.split_bf16 = False
            if split_bf16:
                self.set_split_bf16(split_bf16)
        except Exception:
            self.shutdown()
            raise

    def set_split_bf16(self, on) -> None:
        """False / 0: fp32 matrix instruction everywhere; True / 1: attention on split-bf16 operands; 2: the linear layers and
        convolutions as well."""
        mode = int(on) if not isinstance(on, bool) else int(on)
        check(self._lib, self._lib.s3cfm_set_split_bf16(self._h, mode))
        self.split_bf16 = on if mode else False

    def _check_len(self, L: int) -> None:
        if L < 1 or L > self.max_frames:
            raise ValueError(f"{L} mel frames exceed the workspace ({self.max_frames})")

    def estimator(self, x: torch.Tensor, mu: torch.Tensor, t: float, spks: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """``ConditionalDecoder.forward`` on two rows with one time value: x, mu, cond [2, 80, L], spks [2, 80] -> [2, 80, L]."""
        f = lambda a: a.to(self.device, torch.float32).contiguous()
        x, mu, spks, cond = f(x), f(mu), f(spks), f(cond)
        L = x.shape[2]
        self._check_len(L)
        assert x.shape == (2, CW.MEL, L) and mu.shape == x.shape and cond.shape == x.shape and spks.shape == (2, CW.MEL)
        out = torch.empty_like(x)
        check(self._lib, self._lib.s3cfm_estimator(self._h, _ptr(x), _ptr(mu), _ptr(spks), _ptr(cond), float(t), L, _ptr(out),
                                                   _stream_ptr(self.device)))
        return out

    def solve_euler(self, z: torch.Tensor, t_span: torch.Tensor, mu: torch.Tensor, spks: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """flow_matching.py:82-133 for one utterance: z, mu, cond [1, 80, L], spks [1, 80], t_span [n + 1] -> [1, 80, L]."""
        f = lambda a: a.to(self.device, torch.float32).contiguous()
        z, mu, spks, cond = f(z), f(mu), f(spks), f(cond)
        L = mu.shape[2]
        self._check_len(L)
        assert mu.shape == (1, CW.MEL, L) and z.shape == mu.shape and cond.shape == mu.shape and spks.shape == (1, CW.MEL)
        ts = t_span.detach().to("cpu", torch.float32).contiguous()
        n = ts.numel() - 1
        grid = (C.c_float * (n + 1))(*ts.tolist())
        mel = torch.empty_like(mu)
        check(self._lib, self._lib.s3cfm_solve(self._h, _ptr(mu), _ptr(spks), _ptr(cond), _ptr(z), L, grid, n,
                                               float(self.inference_cfg_rate), _ptr(mel), _stream_ptr(self.device)))
        return mel

    @torch.inference_mode()
    def forward(self, mu: torch.Tensor, mask: Optional[torch.Tensor] = None, n_timesteps: int = 10, temperature: float = 1.0,
                spks: Optional[torch.Tensor] = None, cond: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, None]:
        """flow_matching.py:203-228.  ``mask`` must be all ones (one utterance, as flow.py:278 builds it)."""
        L = mu.shape[2]
        if L > self.rand_noise.shape[2]:
            raise ValueError(f"{L} mel frames exceed the fixed noise buffer ({self.rand_noise.shape[2]})")
        if self._noise_dev is None:
            self._noise_dev = self.rand_noise.to(self.device)
        z = self._noise_dev[:, :, :L] * temperature
        return self.solve_euler(z, CW.t_span(n_timesteps), mu, spks, cond), None

    __call__ = forward

    def solve_euler_batch(self, z: torch.Tensor, t_span: torch.Tensor, mu: torch.Tensor, spks: torch.Tensor, cond: torch.Tensor,
                          frames: Sequence[int]) -> torch.Tensor:
        """``solve_euler`` for n utterances in shared launches: z, mu, cond [n, 80, L] (zeros from ``frames[u]`` on), spks
        [n, 80] -> [n, 80, L]; columns of row u from ``frames[u]`` on are unspecified."""
        f = lambda a: a.to(self.device, torch.float32).contiguous()
        z, mu, spks, cond = f(z), f(mu), f(spks), f(cond)
        n, L = mu.shape[0], mu.shape[2]
        self._check_len(L)
        if n < 1 or n > self.max_batch:
            raise ValueError(f"{n} utterances exceed the workspace ({self.max_batch})")
        assert mu.shape == (n, CW.MEL, L) and z.shape == mu.shape and cond.shape == mu.shape and spks.shape == (n, CW.MEL)
        assert len(frames) == n and all(1 <= int(v) <= L for v in frames)
        ts = t_span.detach().to("cpu", torch.float32).contiguous()
        steps = ts.numel() - 1
        grid = (C.c_float * (steps + 1))(*ts.tolist())
        lens = (C.c_int32 * n)(*[int(v) for v in frames])
        mel = torch.empty_like(mu)
        check(self._lib, self._lib.s3cfm_solve_batch(self._h, n, lens, L, _ptr(mu), _ptr(spks), _ptr(cond), _ptr(z), grid, steps,
                                                     float(self.inference_cfg_rate), _ptr(mel), _stream_ptr(self.device)))
        return mel

    @torch.inference_mode()
    def forward_batch(self, mu: Sequence[torch.Tensor], spks: Sequence[torch.Tensor], cond: Sequence[torch.Tensor], n_timesteps: int = 10,
                      temperature: float = 1.0) -> List[torch.Tensor]:
        """``forward`` for several utterances at once: lists of mu [1, 80, L_u], spks [1, 80], cond [1, 80, L_u] -> list of mel
        [1, 80, L_u].  Every utterance takes the leading columns of ``rand_noise``, as its own ``forward`` call would."""
        n = len(mu)
        frames = [int(m.shape[2]) for m in mu]
        L = max(frames)
        if L > self.rand_noise.shape[2]:
            raise ValueError(f"{L} mel frames exceed the fixed noise buffer ({self.rand_noise.shape[2]})")
        if self._noise_dev is None:
            self._noise_dev = self.rand_noise.to(self.device)
        mu_b = torch.zeros(n, CW.MEL, L, dtype=torch.float32, device=self.device)
        cond_b, z_b = torch.zeros_like(mu_b), torch.zeros_like(mu_b)
        for u in range(n):
            mu_b[u, :, :frames[u]] = mu[u][0]
            cond_b[u, :, :frames[u]] = cond[u][0]
            z_b[u, :, :frames[u]] = self._noise_dev[0, :, :frames[u]] * temperature
        spks_b = torch.cat([s.to(self.device, torch.float32) for s in spks], dim=0)
        mel = self.solve_euler_batch(z_b, CW.t_span(n_timesteps), mu_b, spks_b, cond_b, frames)
        return [mel[u:u + 1, :, :frames[u]] for u in range(n)]

    def shutdown(self) -> None:
        if getattr(self, "_h", None):
            self._lib.s3cfm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass


def linear_op(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, pad: int = 0, gelu: bool = False, res: Optional[torch.Tensor] = None,
              v_from: Optional[int] = None, iters: int = 0, ln: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, split: bool = False):
    """``s3cfm_op_linear`` (``split``: ``s3cfm_op_linear_split``, products on split-bf16 operands) on x [2, Cin, L] (a CUDA tensor; its row stride is the ld the kernel sees), w [Cout, Cin, K] and
    bias [Cout] (CPU): returns (y [2, Cout or v_from, L], vt [2, L, 512] or None, average microseconds of `iters` launches)."""
    lib = _lib()
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[0] == 2 and x.stride(2) == 1
    Cin, L, ld = x.shape[1], x.shape[2], x.stride(1)
    assert x.stride(0) == Cin * ld
    w = w.detach().to("cpu", torch.float32).contiguous()
    bias = bias.detach().to("cpu", torch.float32).contiguous()
    Cout, K = w.shape[0], w.shape[2]
    rows = Cout if v_from is None else v_from
    y_full = torch.zeros(2, rows, ld, dtype=torch.float32, device=x.device)
    vt_full = None if v_from is None else torch.zeros(2, ld, VT_LD, dtype=torch.float32, device=x.device)
    if res is not None:
        r_full = torch.zeros_like(y_full)
        r_full[:, :, :L] = res
    us = C.c_float(0)
    gamma = beta = None
    if ln is not None:                                       # (gain, shift) of a LayerNorm over the input channels, folded into the layer
        gamma, beta = (t.detach().to("cpu", torch.float32).contiguous() for t in ln)
    fn = lib.s3cfm_op_linear_split if split else lib.s3cfm_op_linear
    check(lib, fn(_ptr(x), Cin, L, ld, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), Cout, K, pad, int(gelu),
                                   _ptr(r_full) if res is not None else None, _ptr(y_full), _ptr(vt_full) if vt_full is not None else None,
                                   v_from or 0, C.c_void_p(gamma.data_ptr()) if ln is not None else None,
                                   C.c_void_p(beta.data_ptr()) if ln is not None else None, iters, C.byref(us), _stream_ptr(x.device)))
    return y_full[:, :, :L], (None if vt_full is None else vt_full[:, :L, :512]), us.value


def attention_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, iters: int = 0, keys: Optional[Sequence[int]] = None,
                 split: bool = False):
    """``s3cfm_op_attention`` on q, k, v [2 n, 512, L] (channel-major, head h in channels 64 h ..; rows 2 u and 2 u + 1 belong
    to utterance u, whose first ``keys[u]`` frames are real): returns (o [2 n, 512, L], average microseconds of `iters`
    launches).  ``split``: ``s3cfm_op_attention_split`` (operands as bf16 triples); the time returned is then the pair
    (attention launch, operand re-arrangement launch)."""
    lib = _lib()
    L, rows = q.shape[2], q.shape[0]
    assert rows % 2 == 0 and rows // 2 <= MAX_BATCH
    ld = (L + 63) // 64 * 64
    if (ld // 64) % 2 == 0:
        ld += 64
    dev = q.device
    qk = torch.zeros(rows, 1024, ld, dtype=torch.float32, device=dev)
    qk[:, :512, :L] = q
    qk[:, 512:, :L] = k
    vt = torch.zeros(rows, ld, VT_LD, dtype=torch.float32, device=dev)
    vt[:, :L, :512] = v.transpose(1, 2)
    o = torch.zeros(rows, 512, ld, dtype=torch.float32, device=dev)
    us = C.c_float(0)
    lens = None if keys is None else (C.c_int32 * (rows // 2))(*[int(v) for v in keys])
    if split:
        us2 = C.c_float(0)
        check(lib, lib.s3cfm_op_attention_split(_ptr(qk), _ptr(vt), rows // 2, lens, L, ld, _ptr(o), iters, C.byref(us), C.byref(us2),
                                                _stream_ptr(dev)))
        return o[:, :, :L], (us.value, us2.value)
    check(lib, lib.s3cfm_op_attention(_ptr(qk), _ptr(vt), rows // 2, lens, L, ld, _ptr(o), iters, C.byref(us), _stream_ptr(dev)))
    return o[:, :, :L], us.value
