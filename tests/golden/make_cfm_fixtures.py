#!/usr/bin/env python3
"""Writes tests/golden/cfm_fixture.npz: outputs of the REFERENCE's own flow-matching decoder code.

Run in the build container only (it reads /root/reference):   python3 tests/golden/make_cfm_fixtures.py

The reference's modules for this stage (decoder.py, matcha/decoder.py, matcha/transformer.py, flow_matching.py,
matcha/flow_matching.py under src/chatterbox_vllm/models/s3gen/) are imported UNMODIFIED through an empty parent package
whose __path__ is the reference's s3gen directory.  They name three third-party packages that are not installed here
(`conformer`, `diffusers`, `omegaconf`); those names are provided by stand-in modules built below, and what each
stand-in is decides what a recorded output pins:

PART A -- the reference's own arithmetic only.  The stand-ins that run here are `OmegaConf.create` (a namespace around
the dict it is given: the reference reads `cfm_params.<key>`), `maybe_allow_in_graph` (an identity decorator) and
`get_activation("silu")` (torch's nn.SiLU, which is what diffusers returns for that name); every other stand-in is an
empty class that is never instantiated.  Recorded with this repository's seeded weights (cfm_weights.synthetic_state_dict):
  a_posemb_*      SinusoidalPosEmb(320)                          matcha/decoder.py:14-29
  a_temb_*        TimestepEmbedding(320, 1024, "silu")           matcha/decoder.py:73-118
  a_conv_*        CausalConv1d(256, 256, 3)                      decoder.py:70-97
  a_block_*       CausalBlock1D(256, 256)                        decoder.py:48-61
  a_resnet_*      CausalResnetBlock1D(320, 256, 1024)            decoder.py:63-68, matcha/decoder.py:46-61
  a_euler_*       ConditionalCFM.solve_euler and CausalConditionalCFM.forward with a CLOSED-FORM estimator module
                  (elementwise in x, mu, t, spks, cond), which pins the cosine time grid, the two-row guidance batch with
                  its zeroed second row, the guidance rate and the fp32 order of the Euler update, flow_matching.py:82-133, 203-228

PART B -- ConditionalDecoder.forward whole (decoder.py:233-317) and the 10-step solve around it, with the three classes the
reference takes from diffusers replaced by LABELLED MINIMAL STAND-INS written here from diffusers' published behaviour:
  Attention(query_dim, heads, dim_head, bias=False): to_q / to_k / to_v without bias, softmax(q k^T / sqrt(dim_head) +
      attention_mask) v per head, to_out = [Linear(inner, query_dim) with bias, Dropout];
  GELU(dim_in, dim_out): proj = Linear, then erf-GELU;    LoRACompatibleLinear: nn.Linear.
Part B therefore pins everything of the estimator EXCEPT the inside of those three classes: the input packing order
(x, mu, spks, cond), the time embedding path, the order of resnet block / transformer blocks / trailing convolution in each
stage, the skip connection's channel order, masks, the placement of norm1 / norm3 and the residual adds inside
BasicTransformerBlock (matcha/transformer.py:243-316, the reference's own), final block and projection.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chatterbox_vllm_amd import cfm_weights as CW     # noqa: E402

REF = "/root/reference/src/chatterbox_vllm/models/s3gen"


# ---- stand-ins for the absent third-party names --------------------------------------------------------------------
class StandInAttention(nn.Module):
    """LABELLED MINIMAL STAND-IN for diffusers.models.attention_processor.Attention (self-attention, default processor)."""

    def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0, bias=False, cross_attention_dim=None,
                 upcast_attention=False, **kw):
        super().__init__()
        assert cross_attention_dim is None
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        assert encoder_hidden_states is None
        B, T, _ = hidden_states.shape
        split = lambda t: t.view(B, T, self.heads, self.dim_head).transpose(1, 2)
        q, k, v = split(self.to_q(hidden_states)), split(self.to_k(hidden_states)), split(self.to_v(hidden_states))
        s = q @ k.transpose(-1, -2) / self.dim_head ** 0.5
        if attention_mask is not None:
            s = s + attention_mask.view(B, 1, *attention_mask.shape[-2:])
        o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, self.heads * self.dim_head)
        return self.to_out[1](self.to_out[0](o))


class StandInGELU(nn.Module):
    """LABELLED MINIMAL STAND-IN for diffusers.models.attention.GELU."""

    def __init__(self, dim_in, dim_out, approximate="none"):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


def install_stand_ins():
    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    empty = lambda n: type(n, (), {})
    module("conformer", ConformerBlock=empty("ConformerBlock"))
    module("omegaconf", OmegaConf=type("OmegaConf", (), {"create": staticmethod(lambda d: types.SimpleNamespace(**d))}))
    module("diffusers")
    module("diffusers.models")
    module("diffusers.utils")
    module("diffusers.models.activations", get_activation=lambda name: {"silu": nn.SiLU, "swish": nn.SiLU}[name]())
    module("diffusers.models.attention", GEGLU=empty("GEGLU"), GELU=StandInGELU, AdaLayerNorm=empty("AdaLayerNorm"),
           AdaLayerNormZero=empty("AdaLayerNormZero"), ApproximateGELU=empty("ApproximateGELU"))
    module("diffusers.models.attention_processor", Attention=StandInAttention)
    module("diffusers.models.lora", LoRACompatibleLinear=nn.Linear)
    module("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda cls: cls)


def reference_modules():
    install_stand_ins()
    pkg = types.ModuleType("ref_s3gen")
    pkg.__path__ = [REF]
    sys.modules["ref_s3gen"] = pkg
    for sub in ("matcha", "utils"):
        m = types.ModuleType(f"ref_s3gen.{sub}")
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[f"ref_s3gen.{sub}"] = m
    return (importlib.import_module("ref_s3gen.decoder"), importlib.import_module("ref_s3gen.matcha.decoder"),
            importlib.import_module("ref_s3gen.flow_matching"))


class ClosedFormEstimator(nn.Module):
    """Elementwise stand-in for the network inside solve_euler: every input shows in the output, so the recorded solve
    depends on the guidance batch's layout (which row sees mu / spks / cond), on t and on the update's arithmetic."""

    def forward(self, x, mask, mu, t, spks, cond):
        return (0.5 * x - 0.25 * mu + torch.sin(3.0 * t)[:, None, None] + 0.125 * spks[:, :, None] * x + 0.0625 * cond) * mask


def sub_state(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def main():
    torch.set_num_threads(8)
    dec_mod, matcha_mod, fm_mod = reference_modules()
    sd = CW.synthetic_state_dict()
    P = CW.PREFIX
    out = {}
    g = torch.Generator().manual_seed(2024)
    with torch.inference_mode():
        # ---- part A
        t = torch.tensor([0.0, 0.0123, 0.25, 0.7071, 1.0])
        out["a_posemb_t"] = t.numpy()
        pe = matcha_mod.SinusoidalPosEmb(CW.IN_CH)(t)
        out["a_posemb_out"] = pe.numpy()
        tm = matcha_mod.TimestepEmbedding(in_channels=CW.IN_CH, time_embed_dim=CW.TIME_DIM, act_fn="silu")
        tm.load_state_dict(sub_state(sd, P + "time_mlp."), strict=True)
        temb = tm(pe)
        out["a_temb_out"] = temb.numpy()

        x256 = torch.randn(2, CW.CH, 37, generator=g)
        out["a_x256"] = x256.numpy()
        conv = dec_mod.CausalConv1d(CW.CH, CW.CH, 3)
        conv.load_state_dict(sub_state(sd, P + "down_blocks.0.2."), strict=True)
        out["a_conv_out"] = conv(x256).numpy()
        mask = torch.ones(2, 1, 37)
        mask[1, :, 29:] = 0.0                                        # a padded second row: the masks really act
        out["a_mask"] = mask.numpy()
        blk = dec_mod.CausalBlock1D(CW.CH, CW.CH)
        blk.load_state_dict(sub_state(sd, P + "final_block."), strict=True)
        out["a_block_out"] = blk(x256, mask).numpy()
        x320 = torch.randn(2, CW.IN_CH, 37, generator=g)
        out["a_x320"] = x320.numpy()
        res = dec_mod.CausalResnetBlock1D(dim=CW.IN_CH, dim_out=CW.CH, time_emb_dim=CW.TIME_DIM)
        res.load_state_dict(sub_state(sd, P + "down_blocks.0.0."), strict=True)
        out["a_resnet_temb"] = temb[1:3].numpy()
        out["a_resnet_out"] = res(x320, mask, temb[1:3]).numpy()

        inp = CW.synthetic_inputs(24, 7)
        cfm = fm_mod.CausalConditionalCFM(estimator=ClosedFormEstimator())
        ones = torch.ones(1, 1, 24)
        ts = CW.t_span(10)
        out["a_euler_t_span"] = ts.numpy()
        out["a_euler_solve"] = cfm.solve_euler(inp["noise"].clone(), t_span=ts, mu=inp["mu"], mask=ones, spks=inp["spks"],
                                               cond=inp["cond"]).numpy()
        cfm.rand_noise = torch.randn(1, 80, 64, generator=g)
        out["a_euler_rand_noise"] = cfm.rand_noise.numpy()
        for n in (10, 4):
            mel, _ = cfm.forward(inp["mu"], ones, n_timesteps=n, temperature=0.8 if n == 4 else 1.0, spks=inp["spks"],
                                 cond=inp["cond"])
            out[f"a_euler_forward_{n}"] = mel.numpy()

        # ---- part B
        est = dec_mod.ConditionalDecoder(in_channels=CW.IN_CH, out_channels=CW.MEL, causal=True, channels=[CW.CH],
                                         dropout=0.0, attention_head_dim=CW.HEAD_DIM, n_blocks=CW.N_BLOCKS,
                                         num_mid_blocks=CW.N_MID, num_heads=CW.HEADS, act_fn="gelu")
        assert set(est.state_dict()) == {k[len(P):] for k in sd}, "tensor inventory differs from the reference module tree"
        est.load_state_dict(sub_state(sd, P), strict=True)
        est.eval()
        for tag, frames, prompt, tt in (("b1", 40, 12, 0.3), ("b2", 97, 30, 0.9)):
            i = CW.synthetic_inputs(frames, prompt, seed=11 if tag == "b2" else 5)
            z = torch.zeros(1, 80, frames)
            x = torch.cat([i["noise"], i["noise"]])
            mu = torch.cat([i["mu"], z])
            sp = torch.cat([i["spks"], torch.zeros(1, 80)])
            c = torch.cat([i["cond"], z])
            tv = torch.tensor([tt, tt])
            y = est(x, torch.ones(2, 1, frames), mu, tv, sp, c)
            out[f"{tag}_frames"], out[f"{tag}_prompt"], out[f"{tag}_t"] = np.int64(frames), np.int64(prompt), np.float32(tt)
            out[f"{tag}_seed"] = np.int64(11 if tag == "b2" else 5)
            out[f"{tag}_estimator_out"] = y.numpy()
        i = CW.synthetic_inputs(40, 12)
        whole = fm_mod.CausalConditionalCFM(estimator=est)
        whole.rand_noise = i["noise"].clone()
        mel, _ = whole.forward(i["mu"], torch.ones(1, 1, 40), n_timesteps=10, spks=i["spks"], cond=i["cond"])
        out["b_forward_10"] = mel.numpy()
    for k, v in out.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float32 and v.size > 8:
            print(k, v.shape, "absmax %.4f" % np.abs(v).max(), "mean|.| %.4f" % np.abs(v).mean())
    path = os.path.join(ROOT, "tests", "golden", "cfm_fixture.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
