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


DECISIVE_CODES = 256          # width of one code slice = number of orthogonal token codes
DECISIVE_ACTIVE = 208         # speech ids that carry the structure of ``decisive_state_dict``
DECISIVE_TEXT_CLASSES = 47    # text id t carries the code of class t % 47
DECISIVE_TAG = 768            # residual coordinate that holds the same large value in every structured row
DECISIVE_QK = 3.5             # q and k weight on the tag coordinate: peak score (3.97 * 3.5)^2 * 8 / 8 = 193
DECISIVE_GROUPS = ((5, 1, 1.0), (37, 2, 1.0))                      # (look-back distance, slice written, amplitude)
# head weights of the edges: first (strong, weak source), near, far (speech source, text source); 1 is a logit of 18.3
DECISIVE_LEVELS = (1.57, 0.71, 1.05, 0.85, 0.85)


def _sylvester(n: int) -> torch.Tensor:
    """+-1 Hadamard matrix of order n (a power of two): rows are exactly orthogonal and exact in bf16."""
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return h


def decisive_state_dict(cfg: T3ModelConfig, seed: int = 1234, groups=DECISIVE_GROUPS) -> Dict[str, torch.Tensor]:
    """A seeded checkpoint whose greedy walk is decisive at (nearly) every step AND is decided by attention over the
    paged history, the rotary positions, the text rows and the cond/uncond pairing, for tests that require token ids
    to be EQUAL between two implementations that round alike but sum in different orders.

    ``synthetic_state_dict`` gives nearly flat logits (top-2 gaps of a few bf16 ulp), so two correct implementations
    may pick different ids.  Here the embeddings, the attention weights of layer 0 and the speech head are structured;
    every other tensor (the MLP of layer 0, layers 1.., position tables, norms) is ``synthetic_state_dict``'s.

    * Codes.  256 exactly orthogonal +-1 codes of 256 entries.  Code i < 208 belongs to active speech id ``act[i]``
      (half of them below 1454, the columns multilingual text ids can reach), codes 208..254 to the 47 classes of
      text ids (class = id % 47), code 255 to the start-of-speech id.  A structured embedding row is
      ``[16 * code | 0 | 0 | 32 at coordinate 768, 0 ...]``: coordinates 0..255 identify the token, coordinate 768 is a
      tag whose normalised value (3.97) is the same in every structured row.
    * Layer 0 attention looks back a fixed distance.  Head group g (4 heads) has q = k = a constant taken from the tag
      coordinate in the 8 fastest rotary pairs, with the key pre-rotated by ``delta_g`` positions, so after RoPE the
      score is ``sum_j 12.3 cos((p - s - delta_g) theta_j)``: 98 at the row ``delta_g`` back, 88 next to it, below 75
      elsewhere.  v copies the code coordinates, o writes them (scaled) into residual slice 1 (``delta = 5``) or slice
      2 (``delta = 37``).  Rows that are not structured
      (the 34 conditioning rows: random tag; the zeroed text rows of the uncond stream: no key at all) are poor keys,
      so a look-back that lands on them settles on another row.  Everything here is exact in bf16 or taken from one
      coordinate, so two implementations form bit-identical scores.
    * Speech head.  The row of ``first(a)`` holds a's code in slice 0, the rows of ``near(b)`` / ``far(b)`` hold b's
      code in slice 1 / 2, with strengths that depend on the source token (strong / weak ``first``, loud / quiet
      ``near`` and ``far``).  After token a at position p the candidates are first(a) (from the residual, no attention
      needed), near(token at p-5) and far(token at p-37) (carried by attention, from the KV pages), at distinct logit
      levels between 9 and 33; which one wins depends on the three source tokens and on what the repetition penalty has
      already halved.
    * Stream modes.  With CFG the logit is c + s (c - u).  When the uncond stream attends over its own KV, u carries
      the same near / far votes as c unless the row looked at is a text row (zero in the uncond stream), whose vote is
      then amplified by 1 + s.  When the uncond row has no attention, u carries no vote at all: every attention-borne
      vote is amplified and the ``first`` vote is reduced.  With CFG off nothing is amplified.  The three modes
      therefore walk differently from the first steps on, and a K/V page that is lost, misplaced or read at a wrong
      position changes the ids.
    A few ``first`` edges lead to the stop id, so some walks end on "stop"."""
    sd = synthetic_state_dict(cfg, seed)
    d, S, n_act, n_cls, tag = cfg.d_model, DECISIVE_CODES, DECISIVE_ACTIVE, DECISIVE_TEXT_CLASSES, DECISIVE_TAG
    assert d == 4 * S and cfg.n_heads * cfg.head_dim == d and n_act + n_cls + 1 == S and cfg.n_layers >= 1
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 77)
    codes = _sylvester(S)
    low = torch.randperm(1454, generator=g)[: n_act // 2]
    high = 1454 + torch.randperm(6561 - 1454, generator=g)[: n_act - n_act // 2]
    act = torch.cat([low, high])[torch.randperm(n_act, generator=g)]
    sos_id, stop_id = 6561, 6562                         # modules/t3_config.py:8-9

    def structured_row(code: int) -> torch.Tensor:
        r = torch.zeros(d)
        r[:S] = 16.0 * codes[code]
        r[tag] = 32.0
        return r

    semb = sd["speech_emb.weight"].to(torch.float32)
    for i in range(n_act):
        semb[act[i]] = structured_row(i)
    semb[sos_id] = structured_row(S - 1)
    temb = sd["text_emb.weight"].to(torch.float32)
    for t in range(cfg.text_vocab):
        temb[t] = structured_row(n_act + t % n_cls)
    sd["speech_emb.weight"] = semb.to(torch.bfloat16)
    sd["text_emb.weight"] = temb.to(torch.bfloat16)

    # layer 0: fixed look-back attention
    inv = llama3_inv_freq(cfg)
    wq, wk, wv, wo = (torch.zeros(d, d) for _ in range(4))
    assert len(groups) <= cfg.n_heads // 4
    for gi, (delta, slc, amp) in enumerate(groups):
        for h in range(4 * gi, 4 * gi + 4):
            for j in range(8):
                phi = delta * float(inv[j])
                wq[h * 64 + j, tag] = DECISIVE_QK
                wk[h * 64 + j, tag] = DECISIVE_QK * math.cos(phi)
                wk[h * 64 + 32 + j, tag] = DECISIVE_QK * math.sin(phi)
        for i in range(S):
            wv[S * gi + i, i] = 2.0
            wo[S * slc + i, S * gi + i] = 4.0 * amp
    p = "tfmr.layers.0.self_attn."
    for name, w in (("q", wq), ("k", wk), ("v", wv), ("o", wo)):
        sd[p + name + "_proj.weight"] = w.to(torch.bfloat16)

    # speech head: first / near / far edges
    head = sd["speech_head.weight"].to(torch.float32)
    head[act] *= 0.25

    def cycle() -> torch.Tensor:
        order = torch.randperm(n_act, generator=g)
        nxt = torch.empty(n_act, dtype=torch.long)
        nxt[order] = order.roll(-1)
        return nxt

    edges = []
    for _ in range(3):                                   # one cycle over the active ids + targets for the 48 other codes
        edges.append(act[torch.cat([cycle(), torch.randint(0, n_act, (S - n_act,), generator=g)])])
    first, near, far = edges
    first[torch.randperm(n_act, generator=g)[:3]] = stop_id
    jitter = 1.0 + 0.03 * (2.0 * torch.rand(3, S, generator=g) - 1.0)      # no two edges of one kind weigh the same
    f_strong, f_weak, w_near, w_far, w_far_text = DECISIVE_LEVELS
    for i in range(S):
        u = codes[i] / 16.0                               # unit norm
        head[first[i], 0:S] += (f_strong if i % 2 == 0 else f_weak) * jitter[0, i] * u
        head[near[i], S:2 * S] += w_near * jitter[1, i] * u
        head[far[i], 2 * S:3 * S] += (w_far if i < n_act or i == S - 1 else w_far_text) * jitter[2, i] * u
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
