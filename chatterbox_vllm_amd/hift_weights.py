"""HiFT vocoder (mel -> waveform): geometry, tensor inventory, seeded synthetic weights and the weight-norm fold.

The geometry is the one the reference builds in src/chatterbox_vllm/models/s3gen/s3gen.py:237-245
(HiFTGenerator of hifigan.py:286-380 with a ConvRNNF0Predictor of f0_predictor.py:19-50); tensor names are those of
``S3Token2Wav.mel2wav.state_dict()`` so that a real ``s3gen`` checkpoint's ``mel2wav.*`` entries load unchanged."""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import torch

SR = 24000
MEL = 80
BASE = 512
UP_RATES = (8, 5, 3)
UP_KERNELS = (16, 11, 7)
RES_KERNELS = (3, 7, 11)
RES_DILATIONS = (1, 3, 5)
SRC_RES_KERNELS = (7, 7, 11)
SRC_DOWN = ((30, 15, 7), (6, 3, 1), (1, 1, 0))      # (kernel, stride, padding) of source_downs.i (hifigan.py:353-363)
N_FFT, HOP = 16, 4
NB_HARMONICS = 8
SAMPLES_PER_FRAME = 8 * 5 * 3 * HOP
F0_CHANNELS = 512


def _wn(name: str, shape: Tuple[int, ...]):
    yield name + ".bias", (shape[1] if name.startswith("ups.") else shape[0],)
    yield name + ".parametrizations.weight.original0", (shape[0], 1, 1)
    yield name + ".parametrizations.weight.original1", shape


def _resblock(prefix: str, ch: int, k: int):
    for part in ("convs1", "convs2"):
        for i in range(3):
            yield from _wn(f"{prefix}.{part}.{i}", (ch, ch, k))
    for part in ("activations1", "activations2"):
        for i in range(3):
            yield f"{prefix}.{part}.{i}.alpha", (ch,)


def tensor_inventory() -> Iterator[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every tensor of the vocoder, in state-dict order."""
    yield "m_source.l_linear.weight", (1, NB_HARMONICS + 1)
    yield "m_source.l_linear.bias", (1,)
    yield from _wn("conv_pre", (BASE, MEL, 7))
    for i, k in enumerate(UP_KERNELS):
        yield from _wn(f"ups.{i}", (BASE >> i, BASE >> (i + 1), k))          # ConvTranspose1d: [in, out, k]
    for i, (k, _, _) in enumerate(SRC_DOWN):
        yield f"source_downs.{i}.weight", (BASE >> (i + 1), N_FFT + 2, k)
        yield f"source_downs.{i}.bias", (BASE >> (i + 1),)
    for i, k in enumerate(SRC_RES_KERNELS):
        yield from _resblock(f"source_resblocks.{i}", BASE >> (i + 1), k)
    for i in range(3):
        for j, k in enumerate(RES_KERNELS):
            yield from _resblock(f"resblocks.{i * 3 + j}", BASE >> (i + 1), k)
    yield from _wn("conv_post", (N_FFT + 2, BASE >> 3, 7))
    for i, cin in zip((0, 2, 4, 6, 8), (MEL, F0_CHANNELS, F0_CHANNELS, F0_CHANNELS, F0_CHANNELS)):
        yield from _wn(f"f0_predictor.condnet.{i}", (F0_CHANNELS, cin, 3))
    yield "f0_predictor.classifier.weight", (1, F0_CHANNELS)
    yield "f0_predictor.classifier.bias", (1,)


def synthetic_state_dict(seed: int = 77) -> Dict[str, torch.Tensor]:
    """Seeded weights with activations of order one at every depth (random directions, per-row gains below one so that
    the residual stacks neither die out nor blow up, Snake frequencies around one, an f0 head that lands in the voiced
    range for some frames and below the voicing threshold for others)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in tensor_inventory():
        if name.endswith("original1") or name.startswith("source_downs") and name.endswith("weight"):
            t = torch.randn(shape, generator=g)
            if name.startswith("source_downs"):
                t = t / (shape[1] * shape[2]) ** 0.5
        elif name.endswith("original0"):
            gain = 0.45
            if name.startswith("conv_post"):
                gain = 0.25
            elif name.startswith("conv_pre") or name.startswith("f0_predictor"):
                gain = 1.0
            elif name.startswith("ups."):
                gain = 0.7
            t = gain * (0.75 + 0.5 * torch.rand(shape, generator=g))
        elif name.endswith("alpha"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif name == "f0_predictor.classifier.weight":
            t = torch.randn(shape, generator=g) * 4.0
        elif name == "f0_predictor.classifier.bias":
            t = torch.full(shape, 60.0)
        elif name == "m_source.l_linear.weight":
            t = torch.randn(shape, generator=g)
        else:                                                                   # biases
            t = 0.1 * torch.randn(shape, generator=g)
        sd[name] = t.float().contiguous()
    return sd


def synthetic_mel(n_frames: int, seed: int = 5, batch: int = 1) -> torch.Tensor:
    """A seeded stand-in for a log-mel spectrogram [batch, 80, n_frames] (values in the range S3Gen's flow produces)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(batch, MEL, n_frames, generator=g)
    slow = torch.randn(batch, MEL, (n_frames + 3) // 4, generator=g).repeat_interleave(4, dim=2)[:, :, :n_frames]
    return (-4.0 + 1.5 * slow + 0.5 * base).float().contiguous()


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """w = g * v / ||v|| with the norm over all dims but 0 (torch's weight_norm, dim=0; hifigan.py:120, 331, 339,
    f0_predictor.py:29), in fp32 on the host at load time.  Plain tensors pass through."""
    out: Dict[str, torch.Tensor] = {}
    suffix_v, suffix_g = ".parametrizations.weight.original1", ".parametrizations.weight.original0"
    for k, v in sd.items():
        if k.endswith(suffix_v):
            base = k[: -len(suffix_v)]
            vv = v.float()
            norm = vv.flatten(1).norm(dim=1).view(-1, *([1] * (vv.dim() - 1)))
            out[base + ".weight"] = (sd[base + suffix_g].float() * vv / norm).contiguous()
        elif not k.endswith(suffix_g):
            out[k] = v.float().contiguous()
    return out


def decode_flop(n_frames: int):
    """Algorithmic FLOP (2 x multiply-adds) of the convolutions of ``HiFTGenerator.decode`` for one utterance of
    ``n_frames`` mel frames: (conv_pre, [stage 0, stage 1, stage 2], conv_post).  A stage = its transposed convolution
    (every input position meets every tap), source_downs, the source residual block and the three residual blocks."""
    pre = 2 * MEL * BASE * 7 * n_frames
    stages, length = [], n_frames
    for i in range(3):
        cin, c = BASE >> i, BASE >> (i + 1)
        out_len = length * UP_RATES[i] + (1 if i == 2 else 0)
        f = 2 * cin * c * UP_KERNELS[i] * length
        f += 2 * (N_FFT + 2) * c * SRC_DOWN[i][0] * out_len
        f += 6 * 2 * c * c * SRC_RES_KERNELS[i] * out_len
        for rk in RES_KERNELS:
            f += 6 * 2 * c * c * rk * out_len
        stages.append(f)
        length = out_len
    post = 2 * (BASE >> 3) * (N_FFT + 2) * 7 * length
    return pre, stages, post
