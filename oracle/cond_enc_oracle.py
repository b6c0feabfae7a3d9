"""CPU checker for the conditioning encoder.  TEST INFRASTRUCTURE ONLY (imported by tests/ only).

fp32 torch restatement of the reference's ``T3CondEnc`` + ``Perceiver``
(src/chatterbox_vllm/models/t3/modules/cond_enc.py:57-123, perceiver.py:116-215).  Pinned by
tests/golden/ref_plugin_fixture.npz (``condenc_out_*``: outputs of the reference's own T3CondEnc on the
seeded weights, captured by tests/golden/make_reference_fixtures.py).

  row 0       spkr_enc(speaker_emb)                     Linear 256 -> 1024
  rows 1..32  perceiver(prompt embeddings)              32 learned queries cross-attend to the prompt,
                                                        then self-attend, through ONE shared block
                                                        (LayerNorm, biased q/k/v/out Linears, 4 heads x 256)
  row 33      emotion_adv_fc(exaggeration)              Linear 1 -> 1024, no bias

Weights use the checkpoint names with the ``cond_enc.`` prefix removed.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

N_QUERIES, N_HEADS, DIM, SPK = 32, 4, 1024, 256

def _attention_block(w: Dict[str, torch.Tensor], x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """x1 + proj_out(attention(q(norm x1), k(norm x2), v(norm x2)))   (perceiver.py:159-173)."""
    p = "perceiver.attn."
    ln = lambda x: F.layer_norm(x, (DIM,), w[p + "norm.weight"], w[p + "norm.bias"])
    lin = lambda x, n: F.linear(x, w[p + n + ".weight"], w[p + n + ".bias"])
    n1, n2 = ln(x1), ln(x2)
    split = lambda t: t.view(t.shape[0], N_HEADS, DIM // N_HEADS).permute(1, 0, 2)
    q, k, v = split(lin(n1, "to_q")), split(lin(n2, "to_k")), split(lin(n2, "to_v"))
    att = torch.softmax(q @ k.transpose(1, 2) * (DIM // N_HEADS) ** -0.5, dim=-1) @ v
    att = att.permute(1, 0, 2).reshape(x1.shape[0], DIM)
    return x1 + lin(att, "proj_out")


def emotion_row(w: Dict[str, torch.Tensor], exaggeration: float) -> torch.Tensor:
    return (w["emotion_adv_fc.weight"].float() * float(exaggeration)).reshape(DIM)


@torch.no_grad()
def encode(w: Dict[str, torch.Tensor], speaker_emb: torch.Tensor, prompt_emb: torch.Tensor,
           exaggeration: float = 0.5) -> torch.Tensor:
    """speaker_emb [256] (any shape with 256 elements), prompt_emb [n, 1024] -> fp32 [34, 1024]."""
    w = {k: v.float() for k, v in w.items()}
    spk = F.linear(speaker_emb.float().reshape(SPK), w["spkr_enc.weight"], w["spkr_enc.bias"])[None]
    q = w["perceiver.pre_attention_query"][0]
    pre = _attention_block(w, q, prompt_emb.float())
    res = _attention_block(w, pre, pre)
    return torch.cat([spk, res, emotion_row(w, exaggeration)[None]], 0)
