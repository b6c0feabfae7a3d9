#!/usr/bin/env python3
"""Writes tests/golden/flow_encoder_fixture.npz: what the REFERENCE's flow stage hands to its flow-matching decoder.

Run in the build container only (it reads /root/reference):   python3 tests/golden/make_flow_encoder_fixtures.py

The reference's `s3gen` package cannot be imported as a package here (its __init__ pulls torchaudio, omegaconf, diffusers
and conformer, none installed).  The modules of the token -> mu half need only torch and numpy: flow.py, configs.py,
utils/mask.py, utils/class_utils.py and transformer/*.py.  They are imported unmodified through an empty parent package
whose __path__ is the reference's s3gen directory (so their relative imports resolve, and s3gen/__init__.py is not run).
`CausalMaskedDiffWithXvec` takes its decoder as a constructor argument; it is given a recorder that keeps the arguments
of the call (mu, mask, spks, cond) and returns zeros, so the reference's own `inference` (flow.py:228-278) runs up to the
point where the flow-matching decoder would start.  Weights: flow_weights.synthetic_state_dict, loaded with
load_state_dict (strict over the encoder half)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chatterbox_vllm_amd import flow_weights as FW     # noqa: E402

REF = "/root/reference/src/chatterbox_vllm/models/s3gen"


def reference_modules():
    pkg = types.ModuleType("ref_s3gen")
    pkg.__path__ = [REF]
    sys.modules["ref_s3gen"] = pkg
    for sub in ("transformer", "utils"):
        m = types.ModuleType(f"ref_s3gen.{sub}")
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[f"ref_s3gen.{sub}"] = m
    flow = importlib.import_module("ref_s3gen.flow")
    enc = importlib.import_module("ref_s3gen.transformer.upsample_encoder")
    return flow, enc


class Recorder(torch.nn.Module):
    """Stands where the flow-matching decoder is plugged in: keeps what it is called with."""

    def __init__(self):
        super().__init__()
        self.seen = {}

    def forward(self, mu, mask, spks, cond, n_timesteps, **kw):
        self.seen = {"mu": mu.clone(), "mask": mask.clone(), "spks": spks.clone(), "cond": cond.clone()}
        return torch.zeros_like(mu[:, :80]), None


def main():
    torch.set_num_threads(8)
    flow_mod, enc_mod = reference_modules()
    encoder = enc_mod.UpsampleConformerEncoder(
        output_size=512, attention_heads=8, linear_units=2048, num_blocks=6, dropout_rate=0.1,
        positional_dropout_rate=0.1, attention_dropout_rate=0.1, normalize_before=True, input_layer="linear",
        pos_enc_layer_type="rel_pos_espnet", selfattention_layer_type="rel_selfattn", input_size=512,
        use_cnn_module=False, macaron_style=False)
    rec = Recorder()
    flow = flow_mod.CausalMaskedDiffWithXvec(encoder=encoder, decoder=rec)
    sd = FW.synthetic_state_dict()
    own = {k for k in flow.state_dict() if not k.startswith("decoder.")}
    assert own == set(sd), sorted(own ^ set(sd))[:10]
    flow.load_state_dict(sd, strict=True)
    flow.eval()
    out = {}
    g = torch.Generator().manual_seed(123)
    for tag, n_prompt, n_new in (("a", 7, 30), ("b", 0, 65)):
        prompt = FW.synthetic_tokens(n_prompt, seed=3).view(1, -1)
        new = FW.synthetic_tokens(n_new, seed=4).view(1, -1)
        prompt_feat = torch.randn(1, 2 * n_prompt, 80, generator=g)
        xvec = torch.randn(1, 192, generator=g)
        flow.inference(token=new, token_len=torch.tensor([n_new]), prompt_token=prompt,
                       prompt_token_len=torch.tensor([n_prompt]), prompt_feat=prompt_feat, prompt_feat_len=None,
                       embedding=xvec, finalize=True, n_timesteps=10)
        out[f"{tag}_n_prompt"], out[f"{tag}_n_new"] = np.int64(n_prompt), np.int64(n_new)
        out[f"{tag}_xvec"] = xvec.numpy()
        out[f"{tag}_mu"] = rec.seen["mu"].numpy()             # [1, 80, 2 (n_prompt + n_new)]
        out[f"{tag}_spks"] = rec.seen["spks"].numpy()         # [1, 80]
        out[f"{tag}_cond"] = rec.seen["cond"].numpy()
        out[f"{tag}_prompt_feat"] = prompt_feat.numpy()
    # intermediate: the encoder alone on case a (before encoder_proj)
    with torch.inference_mode():
        tok = torch.cat([FW.synthetic_tokens(7, seed=3), FW.synthetic_tokens(30, seed=4)]).view(1, -1)
        x = flow.input_embedding(tok)
        h, _ = flow.encoder(x, torch.tensor([37]))
        out["a_encoder_out"] = h.numpy()
    for k, v in out.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float32 and v.size:
            print(k, v.shape, "absmax %.4f" % np.abs(v).max(), "mean|.| %.4f" % np.abs(v).mean())
    path = os.path.join(ROOT, "tests", "golden", "flow_encoder_fixture.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
