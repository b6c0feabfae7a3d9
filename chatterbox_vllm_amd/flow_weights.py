"""Token -> mu half of S3Gen's flow stage (token embedding, upsampling conformer encoder, projection to 80 mel channels,
speaker projection): geometry, tensor inventory and seeded synthetic weights.

Geometry as the reference builds it in src/chatterbox_vllm/models/s3gen/s3gen.py:58-73 (UpsampleConformerEncoder) and
flow.py:178-236 (CausalMaskedDiffWithXvec); tensor names are those of ``S3Token2Wav.flow.state_dict()`` so that a real
checkpoint's ``flow.*`` entries load unchanged (the decoder's entries are ignored)."""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import torch

D_MODEL = 512
HEADS = 8
FFN = 2048
N_BLOCKS = 6
N_UP_BLOCKS = 4
VOCAB = 6561
MEL = 80
SPK = 192


def _block(p: str):
    for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
        yield f"{p}self_attn.{n}.weight", (D_MODEL, D_MODEL)
        yield f"{p}self_attn.{n}.bias", (D_MODEL,)
    yield f"{p}self_attn.linear_pos.weight", (D_MODEL, D_MODEL)
    yield f"{p}self_attn.pos_bias_u", (HEADS, D_MODEL // HEADS)
    yield f"{p}self_attn.pos_bias_v", (HEADS, D_MODEL // HEADS)
    yield f"{p}feed_forward.w_1.weight", (FFN, D_MODEL)
    yield f"{p}feed_forward.w_1.bias", (FFN,)
    yield f"{p}feed_forward.w_2.weight", (D_MODEL, FFN)
    yield f"{p}feed_forward.w_2.bias", (D_MODEL,)
    for n in ("norm_ff", "norm_mha"):
        yield f"{p}{n}.weight", (D_MODEL,)
        yield f"{p}{n}.bias", (D_MODEL,)


def tensor_inventory() -> Iterator[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every tensor of the token -> mu path."""
    yield "input_embedding.weight", (VOCAB, D_MODEL)
    yield "spk_embed_affine_layer.weight", (MEL, SPK)
    yield "spk_embed_affine_layer.bias", (MEL,)
    for e in ("embed", "up_embed"):
        yield f"encoder.{e}.out.0.weight", (D_MODEL, D_MODEL)
        yield f"encoder.{e}.out.0.bias", (D_MODEL,)
        yield f"encoder.{e}.out.1.weight", (D_MODEL,)
        yield f"encoder.{e}.out.1.bias", (D_MODEL,)
    yield "encoder.after_norm.weight", (D_MODEL,)
    yield "encoder.after_norm.bias", (D_MODEL,)
    yield "encoder.pre_lookahead_layer.conv1.weight", (D_MODEL, D_MODEL, 4)
    yield "encoder.pre_lookahead_layer.conv1.bias", (D_MODEL,)
    yield "encoder.pre_lookahead_layer.conv2.weight", (D_MODEL, D_MODEL, 3)
    yield "encoder.pre_lookahead_layer.conv2.bias", (D_MODEL,)
    for i in range(N_BLOCKS):
        yield from _block(f"encoder.encoders.{i}.")
    yield "encoder.up_layer.conv.weight", (D_MODEL, D_MODEL, 5)
    yield "encoder.up_layer.conv.bias", (D_MODEL,)
    for i in range(N_UP_BLOCKS):
        yield from _block(f"encoder.up_encoders.{i}.")
    yield "encoder_proj.weight", (MEL, D_MODEL)
    yield "encoder_proj.bias", (MEL,)


def synthetic_state_dict(seed: int = 91) -> Dict[str, torch.Tensor]:
    """Seeded weights that keep activations of order one through the ten blocks: projections N(0, 1/fan_in) with gains
    below one on the residual branches, layer-norm gains around one, attention logits of a few units so that the softmax
    is neither flat nor one-hot."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in tensor_inventory():
        if name == "input_embedding.weight":
            t = torch.randn(shape, generator=g)
        elif name.endswith("pos_bias_u") or name.endswith("pos_bias_v"):
            t = 0.3 * torch.randn(shape, generator=g)
        elif name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".weight"):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 1.0
            if "linear_q" in name or "linear_k" in name or "linear_pos" in name:
                gain = 1.6
            elif "linear_out" in name or "w_2" in name or "pre_lookahead_layer.conv2" in name:
                gain = 0.6
            t = gain * torch.randn(shape, generator=g) / fan_in ** 0.5
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        sd[name] = t.float().contiguous()
    return sd


def synthetic_tokens(n: int, seed: int = 9) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, VOCAB, (n,), generator=g, dtype=torch.int64)
