"""Checkpoint tensor inventory, seeded synthetic checkpoints, and the RoPE table.

The tensor names are the ones the reference's loader routes
(src/chatterbox_vllm/models/t3/t3.py:347-379): everything under ``tfmr.`` goes to
the Llama backbone, ``text_emb`` / ``speech_emb`` / ``text_pos_emb.emb`` /
``speech_pos_emb.emb`` / ``speech_head`` are the T3 additions, ``cond_enc.*`` is the
conditioning encoder (not consumed by the decode engine) and unknown names such as
``text_head.weight`` are skipped (t3.py:363-366).

There is no network in the build or on the GPU box, so benchmarks and tests run on a
seeded random checkpoint of the exact architecture: every matrix N(0, 0.02^2)
(``initializer_range`` in t3-model/config.json:11, position tables as in
modules/learned_pos_emb.py:12), RMSNorm gains 1, stored as bf16.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Iterator, Tuple

import torch

from .config import T3ModelConfig, CONDITIONING_SIZE


def tensor_inventory(cfg: T3ModelConfig) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every tensor the decode engine consumes, in load order."""
    d, f = cfg.d_model, cfg.d_ff
    yield "text_emb.weight", (cfg.text_vocab, d)
    yield "speech_emb.weight", (cfg.speech_vocab, d)
    yield "text_pos_emb.emb.weight", (cfg.text_pos_rows, d)
    yield "speech_pos_emb.emb.weight", (cfg.speech_pos_rows, d)
    yield "speech_head.weight", (cfg.speech_vocab, d)
    for i in range(cfg.n_layers):
        p = f"tfmr.layers.{i}."
        yield p + "input_layernorm.weight", (d,)
        yield p + "self_attn.q_proj.weight", (d, d)
        yield p + "self_attn.k_proj.weight", (d, d)
        yield p + "self_attn.v_proj.weight", (d, d)
        yield p + "self_attn.o_proj.weight", (d, d)
        yield p + "post_attention_layernorm.weight", (d,)
        yield p + "mlp.gate_proj.weight", (f, d)
        yield p + "mlp.up_proj.weight", (f, d)
        yield p + "mlp.down_proj.weight", (d, f)
    yield "tfmr.norm.weight", (d,)


def synthetic_state_dict(cfg: T3ModelConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Seeded random checkpoint (bf16, CPU).  Same bits on every machine: the values
    come from one CPU ``torch.Generator`` walked in ``tensor_inventory`` order."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in tensor_inventory(cfg):
        if name.endswith("layernorm.weight") or name == "tfmr.norm.weight":
            sd[name] = torch.ones(shape, dtype=torch.bfloat16)
        else:
            t = torch.randn(shape, generator=g, dtype=torch.float32) * cfg.initializer_range
            sd[name] = t.to(torch.bfloat16)
    return sd


DECISIVE_ACTIVE = 96          # speech ids that carry the structure of ``decisive_state_dict``


def decisive_state_dict(cfg: T3ModelConfig, seed: int = 1234, n_active: int = DECISIVE_ACTIVE, emb_scale: float = 4.0,
                        gamma: float = 0.9, second: float = 0.75) -> Dict[str, torch.Tensor]:
    """A seeded checkpoint whose greedy choice is decisive at every decode step, for tests that require token ids to
    be EQUAL between two implementations that round alike but sum in different orders.

    ``synthetic_state_dict`` gives nearly flat logits: 8194 Gaussian values whose top-2 gap is a few bf16 ulp, so two
    correct implementations may legitimately pick different ids.  Here ``n_active`` speech ids get embeddings large
    enough to dominate the residual stream, and the speech head maps the direction of an active id ``a`` onto two
    successors: ``first(a)`` with weight ``gamma`` and ``alt(a)`` with ``second * gamma``.  After token ``a`` the
    logit of ``first(a)`` stands out (about 24, ``alt(a)`` 18, against a background of standard deviation 0.7); once
    ``first(a)`` has been emitted the repetition penalty 2.0 halves it and ``alt(a)`` wins (18 against 12; with both
    emitted 12 against 9), so the walk depends on the request's history
    and, for multilingual prompts, on which text ids collide with speech columns.  Half of the active ids lie below
    1454 (the columns multilingual text ids can reach).  A few ids lead to the stop id, so some walks end on "stop".
    The FIRST token of a request is not structured: it comes from the start-of-speech row and depends on the
    whole prompt, as with the flat checkpoint; tests pick prompts whose first step happens to be decisive.
    Everything else (backbone, text tables, position tables) is ``synthetic_state_dict``'s."""
    sd = synthetic_state_dict(cfg, seed)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 77)
    low = torch.randperm(1454, generator=g)[: n_active // 2]
    high = 1454 + torch.randperm(6561 - 1454, generator=g)[: n_active - n_active // 2]
    act = torch.cat([low, high])
    emb = sd["speech_emb.weight"].to(torch.float32)
    emb[act] = torch.randn(n_active, cfg.d_model, generator=g) * emb_scale
    first = act[torch.randint(0, n_active, (n_active,), generator=g)]
    alt = act[torch.randint(0, n_active, (n_active,), generator=g)]
    stop_id = 6562                                        # modules/t3_config.py:9 (stop_speech_token)
    first[torch.randperm(n_active, generator=g)[:3]] = stop_id
    head = sd["speech_head.weight"].to(torch.float32)
    unit = emb[act] / (emb_scale * math.sqrt(cfg.d_model))
    for i in range(n_active):
        head[first[i]] += gamma * unit[i]
        head[alt[i]] += second * gamma * unit[i]
    sd["speech_emb.weight"] = emb.to(torch.bfloat16)
    sd["speech_head.weight"] = head.to(torch.bfloat16)
    return sd


def synthetic_cond_emb(voice: int = 0, d_model: int = 1024, seed: int = 4321) -> torch.Tensor:
    """Stand-in for ``T3CondEnc`` output: fp32 [34, d_model] on the CPU, which is what
    ``get_audio_conditionals`` hands to the engine (src/chatterbox_vllm/tts.py:240-247)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + voice)
    return torch.randn((CONDITIONING_SIZE, d_model), generator=g, dtype=torch.float32) * 0.02


def llama3_inv_freq(cfg: T3ModelConfig) -> torch.Tensor:
    """Inverse RoPE frequencies with the llama3 long-context rescale
    (t3-model/config.json:21-28).  fp32, [head_dim/2]."""
    dim = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    low_wavelen = cfg.rope_original_max_pos / cfg.rope_low_freq_factor
    high_wavelen = cfg.rope_original_max_pos / cfg.rope_high_freq_factor
    wavelen = 2 * math.pi / inv
    scaled = torch.where(wavelen > low_wavelen, inv / cfg.rope_factor, inv)
    smooth = (cfg.rope_original_max_pos / wavelen - cfg.rope_low_freq_factor) / (
        cfg.rope_high_freq_factor - cfg.rope_low_freq_factor)
    mid = (1 - smooth) * scaled / cfg.rope_factor + smooth * scaled
    is_mid = (wavelen >= high_wavelen) & (wavelen <= low_wavelen)
    return torch.where(is_mid, mid, scaled).float()


def rope_table(cfg: T3ModelConfig, max_pos: int) -> torch.Tensor:
    """bf16 [max_pos, head_dim]: columns 0:32 cos, 32:64 sin.  Angles are formed in
    fp32 and the table is then cast to the model dtype, so the engine and any checker
    fed the same table rotate with identical bits."""
    inv = llama3_inv_freq(cfg)
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1).to(torch.bfloat16)


def route_checkpoint(items: Iterable[Tuple[str, torch.Tensor]], cfg: T3ModelConfig
                     ) -> Tuple[Dict[str, torch.Tensor], list]:
    """Split a checkpoint stream into (tensors the engine loads, skipped names),
    mirroring t3.py:347-369: ``cond_enc.*`` belongs to the host-side conditioning
    encoder, names outside the inventory are ignored rather than rejected."""
    wanted = dict(tensor_inventory(cfg))
    keep: Dict[str, torch.Tensor] = {}
    skipped = []
    for name, t in items:
        if name in wanted:
            if tuple(t.shape) != wanted[name]:
                raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != expected {wanted[name]}")
            keep[name] = t
        else:
            skipped.append(name)
    missing = [n for n in wanted if n not in keep]
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, first: {missing[:3]}")
    return keep, skipped


PROJECTION_SUFFIXES = ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                       "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")


def quantize_int8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """The engine's W8A16 rule for one projection matrix [N, K] (``quantization="int8"``): per output row,
    scale = absmax / 127 in fp32 (1 for an all-zero row), q = round-half-even(w / scale) clamped to +-127.
    Returns (q int8 [N, K], scale fp32 [N]).  The engine quantises on the device with the same arithmetic."""
    wf = w.detach().to(torch.float32)
    amax = wf.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 127.0, torch.ones_like(amax))
    q = torch.round(wf / scale[:, None]).clamp_(-127, 127).to(torch.int8)
    return q, scale


def int8_dequantized_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``sd`` with every backbone projection replaced by scale * q (fp32): the weights an int8 engine
    computes with, for checkers that want to follow it exactly.  Embeddings, norms and the speech head
    stay as they are (the engine keeps them bf16)."""
    out = dict(sd)
    for name, t in sd.items():
        if name.startswith("tfmr.layers.") and name.endswith(PROJECTION_SUFFIXES):
            q, scale = quantize_int8(t)
            out[name] = q.to(torch.float32) * scale[:, None]
    return out
