"""HiFT vocoder (mel -> 24 kHz waveform) on the GPU: the host-side mirror of the reference's ``HiFTGenerator``
(src/chatterbox_vllm/models/s3gen/hifigan.py:286-474) over the C ABI of ``lib/libs3hift.so`` (include/s3hift_hip.h).

Same method names and argument meaning as the reference class: ``inference(speech_feat, cache_source)`` returns
``(waveform, source)``, ``decode(x, s)`` the waveform, ``f0_predictor(x)`` the f0 track.  Tensors are fp32 on the
handle's GPU; torch is used for device memory only.  Loading fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Dict, Optional, Tuple

import torch

from . import hift_weights as HW

ABI_VERSION = 1
LIB_NAME = "libs3hift.so"
_P = C.c_void_p

_SIGNATURES = {
    "s3hift_abi_version": ([], C.c_int),
    "s3hift_last_error": ([], C.c_char_p),
    "s3hift_create": ([C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)], C.c_int),
    "s3hift_destroy": ([_P], C.c_int),
    "s3hift_load_tensor": ([_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32], C.c_int),
    "s3hift_finalize": ([_P], C.c_int),
    "s3hift_f0": ([_P, _P, C.c_int32, C.c_int32, _P, _P], C.c_int),
    "s3hift_source": ([_P, _P, C.c_int32, C.c_int32, C.c_uint64, _P, _P, _P, _P], C.c_int),
    "s3hift_decode": ([_P, _P, _P, C.c_int32, C.c_int32, _P, _P], C.c_int),
    "s3hift_inference": ([_P, _P, C.c_int32, C.c_int32, C.c_uint64, _P, C.c_int32, _P, _P, _P], C.c_int),
    "s3hift_set_timing": ([_P, C.c_int32], C.c_int),
    "s3hift_last_timing": ([_P, C.POINTER(C.c_float), C.c_int32], C.c_int),
    "s3hift_op_conv1d": ([_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                          C.c_int32, C.c_int32, C.c_float, _P], C.c_int),
}


class HiftError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("S3HIFT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", LIB_NAME)


@functools.lru_cache(maxsize=None)
def load() -> C.CDLL:
    """Opens libs3hift.so (after torch, so that both bind to one HIP runtime) and declares every entry point."""
    path = lib_path()
    if not os.path.exists(path):
        raise HiftError(f"{path} is missing: build it with `make -C chatterbox_vllm_amd/csrc_s3gen` (there is no CPU fallback)")
    lib = C.CDLL(path)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, restype
    if lib.s3hift_abi_version() != ABI_VERSION:
        raise HiftError(f"{path} has ABI {lib.s3hift_abi_version()}, this package needs {ABI_VERSION}")
    return lib


def check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        raise HiftError((lib.s3hift_last_error() or b"?").decode())


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream_ptr(device: torch.device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class HiFTGenerator:
    """``HiFTGenerator(state_dict)``: the vocoder with the geometry of s3gen.py:237-245.  ``state_dict`` holds the tensors
    of ``S3Token2Wav.mel2wav.state_dict()`` (weight-norm parametrised, as checkpoints store them, or already folded)."""

    sampling_rate = HW.SR

    def __init__(self, state_dict: Dict[str, torch.Tensor], max_batch: int = 1, max_frames: int = 1500, device: int = 0):
        self._lib = load()
        self.device = torch.device("cuda", device)
        self.max_batch, self.max_frames = max_batch, max_frames
        h = _P()
        check(self._lib, self._lib.s3hift_create(device, max_batch, max_frames, C.byref(h)))
        self._h = h
        try:
            for name, t in HW.fold_weight_norm(state_dict).items():
                t = t.detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                check(self._lib, self._lib.s3hift_load_tensor(h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            check(self._lib, self._lib.s3hift_finalize(h))
        except Exception:
            self.shutdown()
            raise

    # -- reference surface -------------------------------------------------------------------------------------------
    def f0_predictor(self, x: torch.Tensor) -> torch.Tensor:
        """ConvRNNF0Predictor.forward (f0_predictor.py:52-55): x [B, 80, T] -> f0 [B, T]."""
        x, B, T = self._mel(x)
        f0 = torch.empty(B, T, dtype=torch.float32, device=self.device)
        check(self._lib, self._lib.s3hift_f0(self._h, _ptr(x), B, T, _ptr(f0), _stream_ptr(self.device)))
        return f0

    def source(self, f0: torch.Tensor, seed: int = 0, phase: Optional[torch.Tensor] = None,
               noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """f0 [B, T] -> source [B, 1, 480 T] (hifigan.py:455-457).  ``phase`` [B, 9] and ``noise`` [B, 9, 480 T] replace
        the two random draws when given; otherwise they come from ``seed`` on the device."""
        f0 = f0.to(self.device, torch.float32).contiguous()
        B, T = f0.shape
        if phase is not None:
            phase = phase.to(self.device, torch.float32).reshape(B, HW.NB_HARMONICS + 1).contiguous()
        if noise is not None:
            noise = noise.to(self.device, torch.float32).reshape(B, HW.NB_HARMONICS + 1, T * HW.SAMPLES_PER_FRAME).contiguous()
        s = torch.empty(B, 1, T * HW.SAMPLES_PER_FRAME, dtype=torch.float32, device=self.device)
        check(self._lib, self._lib.s3hift_source(self._h, _ptr(f0), B, T, C.c_uint64(seed & (2 ** 64 - 1)), _ptr(phase),
                                                 _ptr(noise), _ptr(s), _stream_ptr(self.device)))
        return s

    def decode(self, x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
        """HiFTGenerator.decode (hifigan.py:412-444): x [B, 80, T], s [B, 1, 480 T] -> waveform [B, 480 T]."""
        x, B, T = self._mel(x)
        s = s.to(self.device, torch.float32).contiguous()
        assert s.shape == (B, 1, T * HW.SAMPLES_PER_FRAME), f"source shape {tuple(s.shape)} does not fit {T} frames"
        wav = torch.empty(B, T * HW.SAMPLES_PER_FRAME, dtype=torch.float32, device=self.device)
        check(self._lib, self._lib.s3hift_decode(self._h, _ptr(x), _ptr(s), B, T, _ptr(wav), _stream_ptr(self.device)))
        return wav

    def inference(self, speech_feat: torch.Tensor, cache_source: Optional[torch.Tensor] = None,
                  seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """HiFTGenerator.inference (hifigan.py:462-474): (waveform [B, 480 T], source [B, 1, 480 T])."""
        x, B, T = self._mel(speech_feat)
        n_cache = 0
        if cache_source is not None and cache_source.shape[-1] != 0:
            cache_source = cache_source.to(self.device, torch.float32).reshape(-1, cache_source.shape[-1])
            cache_source = cache_source.expand(B, -1).contiguous()
            n_cache = cache_source.shape[1]
        else:
            cache_source = None
        L = T * HW.SAMPLES_PER_FRAME
        wav = torch.empty(B, L, dtype=torch.float32, device=self.device)
        s = torch.empty(B, 1, L, dtype=torch.float32, device=self.device)
        check(self._lib, self._lib.s3hift_inference(self._h, _ptr(x), B, T, C.c_uint64(seed & (2 ** 64 - 1)),
                                                    _ptr(cache_source), n_cache, _ptr(wav), _ptr(s), _stream_ptr(self.device)))
        return wav, s

    # -- extras ------------------------------------------------------------------------------------------------------
    def set_timing(self, on: bool) -> None:
        check(self._lib, self._lib.s3hift_set_timing(self._h, int(on)))

    def last_timing(self):
        """Milliseconds of the last decode: [STFT + conv_pre, stage 0, stage 1, stage 2, conv_post + ISTFT, whole call]."""
        out = (C.c_float * 6)()
        check(self._lib, self._lib.s3hift_last_timing(self._h, out, 6))
        return list(out)

    def shutdown(self) -> None:
        if getattr(self, "_h", None):
            self._lib.s3hift_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    def _mel(self, x: torch.Tensor):
        x = x.to(self.device, torch.float32).contiguous()
        assert x.dim() == 3 and x.shape[1] == HW.MEL, f"mel must be [B, 80, T], got {tuple(x.shape)}"
        B, T = x.shape[0], x.shape[2]
        if B > self.max_batch or T > self.max_frames:
            raise ValueError(f"batch {B} x {T} frames exceeds the workspace ({self.max_batch} x {self.max_frames})")
        return x, B, T


def conv1d_op(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, dilation: int = 1, padding: int = 0, act: int = 0,
              alpha: Optional[torch.Tensor] = None, slope: float = 0.0, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One launch of the convolution kernel (s3hift_op_conv1d), for tests."""
    lib = load()
    B, Cin, Tin = x.shape
    Cout, _, K = w.shape
    Tout = Tin + 2 * padding - dilation * (K - 1)
    y = torch.empty(B, Cout, Tout, dtype=torch.float32, device=x.device)
    wh, bh = w.detach().to("cpu", torch.float32).contiguous(), bias.detach().to("cpu", torch.float32).contiguous()
    check(lib, lib.s3hift_op_conv1d(_ptr(x.contiguous()), C.c_void_p(wh.data_ptr()), C.c_void_p(bh.data_ptr()), _ptr(alpha),
                                    _ptr(res), _ptr(y), B, Cin, Cout, K, Tin, dilation, padding, act, C.c_float(slope),
                                    _stream_ptr(x.device)))
    return y
