"""Conditioning prefix: (speaker embedding, 150 prompt-speech embeddings, emotion scalar) -> [34, 1024].

The reference's ``T3CondEnc`` + ``Perceiver`` (src/chatterbox_vllm/models/t3/modules/cond_enc.py:57-123,
perceiver.py:116-215) run as fp32 HIP kernels in ``libt3hip.so`` (csrc/condenc.hip) behind the
``t3_condenc_*`` entry points; this module loads the ``cond_enc.*`` checkpoint tensors into that
encoder and calls it.  It runs once per voice and the caller caches the result
(``get_audio_conditionals``), so it is not on the per-token path.

  row 0       spkr_enc(speaker_emb)                     Linear 256 -> 1024
  rows 1..32  perceiver(prompt embeddings)              32 learned queries cross-attend to the prompt,
                                                        then self-attend, through ONE shared block
                                                        (LayerNorm, biased q/k/v/out Linears, 4 heads x 256)
  row 33      emotion_adv_fc(exaggeration)              Linear 1 -> 1024, no bias
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _capi

N_QUERIES, N_HEADS, DIM, SPK = 32, 4, 1024, 256

_SHAPES = {
    "spkr_enc.weight": (DIM, SPK), "spkr_enc.bias": (DIM,),
    "emotion_adv_fc.weight": (DIM, 1),
    "perceiver.pre_attention_query": (1, N_QUERIES, DIM),
    "perceiver.attn.norm.weight": (DIM,), "perceiver.attn.norm.bias": (DIM,),
    "perceiver.attn.to_q.weight": (DIM, DIM), "perceiver.attn.to_q.bias": (DIM,),
    "perceiver.attn.to_k.weight": (DIM, DIM), "perceiver.attn.to_k.bias": (DIM,),
    "perceiver.attn.to_v.weight": (DIM, DIM), "perceiver.attn.to_v.bias": (DIM,),
    "perceiver.attn.proj_out.weight": (DIM, DIM), "perceiver.attn.proj_out.bias": (DIM,),
}


def synthetic_weights(seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Seeded stand-in for the ``cond_enc.*`` tensors of a checkpoint (keys keep the prefix)."""
    g = torch.Generator().manual_seed(seed + 99)
    out = {}
    for name, shape in _SHAPES.items():
        if name.endswith("norm.weight"):
            t = torch.ones(shape)
        elif name.endswith("bias"):
            t = torch.zeros(shape)
        else:
            t = torch.randn(shape, generator=g) * 0.02
        out["cond_enc." + name] = t
    return out


def _fptr(t: torch.Tensor):
    return C.cast(_capi.ptr(t), C.POINTER(C.c_float))


class CondEncoder:
    """The GPU conditioning encoder of one checkpoint.  ``weights``: name (without ``cond_enc.``) -> tensor."""

    def __init__(self, weights: Dict[str, torch.Tensor], device: int = 0):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        _capi.check(self._lib, self._lib.t3_condenc_create(int(device), C.byref(self._h)))
        missing = [n for n in _SHAPES if n not in weights]
        if missing:
            raise KeyError(f"checkpoint is missing cond_enc tensors: {missing[:3]}")
        for name in _SHAPES:
            t = weights[name].detach().to(device="cpu", dtype=torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _capi.check(self._lib, self._lib.t3_condenc_load_tensor(self._h, name.encode(), _fptr(t), shape, t.dim()))

    def encode(self, speaker_emb: torch.Tensor, prompt_emb: torch.Tensor, exaggeration: float = 0.5) -> torch.Tensor:
        """speaker_emb: 256 values (any shape), prompt_emb [n, 1024] -> fp32 [34, 1024] on the CPU."""
        spk = speaker_emb.detach().to(device="cpu", dtype=torch.float32).reshape(SPK).contiguous()
        pe = prompt_emb.detach().to(device="cpu", dtype=torch.float32).reshape(-1, DIM).contiguous()
        out = torch.empty(N_QUERIES + 2, DIM, dtype=torch.float32)
        _capi.check(self._lib, self._lib.t3_condenc_encode(self._h, _fptr(spk), _fptr(pe), pe.shape[0],
                                                          C.c_float(float(exaggeration)), _fptr(out)))
        return out

    def emotion_row(self, exaggeration: float) -> torch.Tensor:
        out = torch.empty(DIM, dtype=torch.float32)
        _capi.check(self._lib, self._lib.t3_condenc_emotion_row(self._h, C.c_float(float(exaggeration)), _fptr(out)))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.t3_condenc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
