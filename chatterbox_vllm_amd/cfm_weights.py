"""Flow-matching decoder of S3Gen's flow stage (mu -> mel): geometry, tensor inventory and seeded synthetic weights.

Geometry as the reference builds it: ``ConditionalDecoder(in_channels=320, out_channels=80, causal=True, channels=[256],
attention_head_dim=64, n_blocks=4, num_mid_blocks=12, num_heads=8, act_fn="gelu")``
(src/chatterbox_vllm/models/s3gen/s3gen.py:76-100, decoder.py:100-217) inside ``CausalConditionalCFM``
(flow_matching.py:198-228).  Tensor names are those of ``S3Token2Wav.flow.state_dict()`` (prefix ``decoder.estimator.``), so
a real checkpoint's entries load unchanged."""
from __future__ import annotations

import math
from typing import Dict, Iterator, Tuple

import torch

PREFIX = "decoder.estimator."
MEL = 80
IN_CH = 320                  # x, mu, speaker vector, prompt condition: 4 x 80 rows
CH = 256
TIME_DIM = 1024
HEADS = 8
HEAD_DIM = 64
INNER = HEADS * HEAD_DIM     # 512
FF = 4 * CH
N_BLOCKS = 4                 # transformer blocks after every resnet block
N_MID = 12
CFG_RATE = 0.7               # flow_matching.py:21-28 inference_cfg_rate
NOISE_FRAMES = 50 * 300      # flow_matching.py:201


def _resnet(p: str, cin: int):
    for blk, c in (("block1", cin), ("block2", CH)):
        yield f"{p}{blk}.block.0.weight", (CH, c, 3)
        yield f"{p}{blk}.block.0.bias", (CH,)
        yield f"{p}{blk}.block.2.weight", (CH,)
        yield f"{p}{blk}.block.2.bias", (CH,)
    yield f"{p}mlp.1.weight", (CH, TIME_DIM)
    yield f"{p}mlp.1.bias", (CH,)
    yield f"{p}res_conv.weight", (CH, cin, 1)
    yield f"{p}res_conv.bias", (CH,)


def _transformer(p: str):
    yield f"{p}norm1.weight", (CH,)
    yield f"{p}norm1.bias", (CH,)
    for n in ("to_q", "to_k", "to_v"):
        yield f"{p}attn1.{n}.weight", (INNER, CH)
    yield f"{p}attn1.to_out.0.weight", (CH, INNER)
    yield f"{p}attn1.to_out.0.bias", (CH,)
    yield f"{p}norm3.weight", (CH,)
    yield f"{p}norm3.bias", (CH,)
    yield f"{p}ff.net.0.proj.weight", (FF, CH)
    yield f"{p}ff.net.0.proj.bias", (FF,)
    yield f"{p}ff.net.2.weight", (CH, FF)
    yield f"{p}ff.net.2.bias", (CH,)


def stages() -> Iterator[Tuple[str, int, bool]]:
    """(prefix, input channels of the resnet block, has a trailing causal convolution) of the 14 resnet + transformer
    stages in execution order (decoder.py:264-314): one down stage, twelve mid stages, one up stage fed with the skip."""
    yield "down_blocks.0.", IN_CH, True
    for i in range(N_MID):
        yield f"mid_blocks.{i}.", CH, False
    yield "up_blocks.0.", 2 * CH, True


def tensor_inventory() -> Iterator[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every tensor of the estimator."""
    p = PREFIX
    yield p + "time_mlp.linear_1.weight", (TIME_DIM, IN_CH)
    yield p + "time_mlp.linear_1.bias", (TIME_DIM,)
    yield p + "time_mlp.linear_2.weight", (TIME_DIM, TIME_DIM)
    yield p + "time_mlp.linear_2.bias", (TIME_DIM,)
    for sp, cin, tail in stages():
        yield from _resnet(f"{p}{sp}0.", cin)
        for j in range(N_BLOCKS):
            yield from _transformer(f"{p}{sp}1.{j}.")
        if tail:
            yield f"{p}{sp}2.weight", (CH, CH, 3)
            yield f"{p}{sp}2.bias", (CH,)
    yield p + "final_block.block.0.weight", (CH, CH, 3)
    yield p + "final_block.block.0.bias", (CH,)
    yield p + "final_block.block.2.weight", (CH,)
    yield p + "final_block.block.2.bias", (CH,)
    yield p + "final_proj.weight", (MEL, CH, 1)
    yield p + "final_proj.bias", (MEL,)


def synthetic_state_dict(seed: int = 57) -> Dict[str, torch.Tensor]:
    """Seeded weights that keep activations of order one through the 14 stages: projections N(0, 1/fan_in), gains below
    one on the residual branches, layer-norm gains around one, q / k gains that give attention logits of a few units."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in tensor_inventory():
        if name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".weight"):
            fan_in = math.prod(shape[1:])
            gain = 1.0
            if "to_q" in name or "to_k" in name:
                gain = 1.5
            elif "to_out" in name or "ff.net.2" in name:
                gain = 0.5
            elif "final_proj" in name:
                gain = 2.0
            t = gain * torch.randn(shape, generator=g) / fan_in ** 0.5
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        sd[name] = t.float().contiguous()
    return sd


def synthetic_inputs(frames: int, prompt_frames: int, seed: int = 5) -> Dict[str, torch.Tensor]:
    """mu [1, 80, L], spks [1, 80], cond [1, 80, L] (prompt mel frames, then zeros), noise [1, 80, L] of the magnitudes the
    stages before the decoder produce."""
    g = torch.Generator().manual_seed(seed)
    mu = torch.randn(1, MEL, frames, generator=g)
    spks = 0.5 * torch.randn(1, MEL, generator=g)
    cond = torch.zeros(1, MEL, frames)
    cond[:, :, :prompt_frames] = 2.0 * torch.randn(1, MEL, prompt_frames, generator=g) - 4.0
    noise = torch.randn(1, MEL, frames, generator=g)
    return {"mu": mu, "spks": spks, "cond": cond, "noise": noise}


def t_span(n_timesteps: int) -> torch.Tensor:
    """The cosine time grid of flow_matching.py:225-227, computed with the same fp32 torch operations."""
    t = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)
    return 1 - torch.cos(t * 0.5 * torch.pi)


def estimator_flop(frames: int) -> float:
    """Multiply-add arithmetic (2 x MACs) of one estimator call on both guidance rows."""
    per_frame = 0.0
    for _, cin, tail in stages():
        per_frame += 2 * (3 * cin * CH + 3 * CH * CH + cin * CH)
        per_frame += N_BLOCKS * 2 * (3 * CH * INNER + INNER * CH + 2 * CH * FF + 2 * frames * INNER)
        if tail:
            per_frame += 2 * 3 * CH * CH
    per_frame += 2 * (3 * CH * CH + CH * MEL)
    return 2.0 * frames * per_frame
