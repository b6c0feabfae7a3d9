"""Token -> mu half of S3Gen's flow stage on the GPU: the host-side mirror of what the reference's
``CausalMaskedDiffWithXvec.inference`` (src/chatterbox_vllm/models/s3gen/flow.py:228-278) computes before it calls its
flow-matching decoder, over the C ABI of include/s3flow_hip.h (entry points in ``lib/libs3hift.so``).

``FlowEncoder.decoder_inputs`` takes the arguments of the reference method and returns the four tensors the reference
hands to its decoder (``mu``, ``mask``, ``spks``, ``cond``).  fp32 on the handle's GPU; torch carries device memory only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import flow_weights as FW
from .hift import HiftError, check, load, _P, _ptr, _stream_ptr

_SIGNATURES = {
    "s3flow_enc_create": ([C.c_int32, C.c_int32, C.POINTER(_P)], C.c_int),
    "s3flow_enc_destroy": ([_P], C.c_int),
    "s3flow_enc_load_tensor": ([_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32], C.c_int),
    "s3flow_enc_finalize": ([_P], C.c_int),
    "s3flow_enc_forward": ([_P, _P, C.c_int32, _P, _P], C.c_int),
    "s3flow_enc_speaker": ([_P, _P, _P, _P], C.c_int),
    "s3flow_enc_last_hidden": ([_P, _P, C.c_int32, _P], C.c_int),
}


def _lib() -> C.CDLL:
    lib = load()
    if not getattr(lib, "_s3flow_declared", False):
        for name, (argtypes, restype) in _SIGNATURES.items():
            if not hasattr(lib, name):
                raise HiftError(f"libs3hift.so does not export {name}: rebuild it (make -C chatterbox_vllm_amd/csrc_s3gen)")
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = argtypes, restype
        lib._s3flow_declared = True
    return lib


class FlowEncoder:
    """``FlowEncoder(state_dict)``: ``state_dict`` holds ``S3Token2Wav.flow.state_dict()`` (entries of the decoder are
    ignored), geometry of s3gen.py:58-73."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], max_tokens: int = 1500, device: int = 0):
        self._lib = _lib()
        self.device = torch.device("cuda", device)
        self.max_tokens = max_tokens
        h = _P()
        check(self._lib, self._lib.s3flow_enc_create(device, max_tokens, C.byref(h)))
        self._h = h
        try:
            wanted = {name for name, _ in FW.tensor_inventory()}
            for name in wanted:
                if name not in state_dict:
                    raise KeyError(f"state dict lacks {name}")
                t = state_dict[name].detach().to("cpu", torch.float32).contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                check(self._lib, self._lib.s3flow_enc_load_tensor(h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()))
            check(self._lib, self._lib.s3flow_enc_finalize(h))
        except Exception:
            self.shutdown()
            raise

    def token_to_mu(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [T] or [1, T] (prompt tokens followed by the new ones) -> mu [1, 80, 2T], the reference's
        ``encoder_proj(encoder(input_embedding(token))).transpose(1, 2)`` (flow.py:250-258, 269)."""
        tok = tokens.reshape(-1).to(self.device, torch.int32).contiguous()
        T = tok.numel()
        if T < 1 or T > self.max_tokens:
            raise ValueError(f"{T} tokens exceed the workspace ({self.max_tokens})")
        mu = torch.empty(1, FW.MEL, 2 * T, dtype=torch.float32, device=self.device)
        check(self._lib, self._lib.s3flow_enc_forward(self._h, _ptr(tok), T, _ptr(mu), _stream_ptr(self.device)))
        return mu

    def last_hidden(self, T: int) -> torch.Tensor:
        """The encoder output before ``encoder_proj`` of the last call, [2T, 512]."""
        out = torch.empty(FW.D_MODEL, 2 * T, dtype=torch.float32, device=self.device)
        check(self._lib, self._lib.s3flow_enc_last_hidden(self._h, _ptr(out), T, _stream_ptr(self.device)))
        return out.t()

    def speaker(self, embedding: torch.Tensor) -> torch.Tensor:
        """x-vector [1, 192] -> [1, 80] (flow.py:244-245)."""
        x = embedding.reshape(-1).to(self.device, torch.float32).contiguous()
        assert x.numel() == FW.SPK
        out = torch.empty(1, FW.MEL, dtype=torch.float32, device=self.device)
        check(self._lib, self._lib.s3flow_enc_speaker(self._h, _ptr(x), _ptr(out), _stream_ptr(self.device)))
        return out

    def decoder_inputs(self, token: torch.Tensor, prompt_token: torch.Tensor, prompt_feat: torch.Tensor,
                       embedding: torch.Tensor) -> Dict[str, torch.Tensor]:
        """The keyword arguments of the reference's decoder call (flow.py:268-274) with ``finalize=True``:
        mu [1, 80, L], mask [1, 1, L], spks [1, 80], cond [1, 80, L] (the prompt's mel frames, then zeros)."""
        assert token.shape[0] == 1 and prompt_token.shape[0] == 1, "one utterance per call, as the reference asserts"
        tokens = torch.cat([prompt_token.reshape(-1).to(self.device), token.reshape(-1).to(self.device)])
        mu = self.token_to_mu(tokens)
        L = mu.shape[2]
        mel_len1 = prompt_feat.shape[1]
        cond = torch.zeros(1, FW.MEL, L, dtype=torch.float32, device=self.device)
        cond[:, :, :mel_len1] = prompt_feat.to(self.device, torch.float32).transpose(1, 2)
        mask = torch.ones(1, 1, L, dtype=torch.float32, device=self.device)
        return {"mu": mu, "mask": mask, "spks": self.speaker(embedding), "cond": cond, "prompt_len": mel_len1}

    def shutdown(self) -> None:
        if getattr(self, "_h", None):
            self._lib.s3flow_enc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
