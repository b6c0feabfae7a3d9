"""CPU tests of the flow-matching decoder's restatement (oracle/cfm_oracle.py) and of the package's host side for it.

The restatement is PARITY UNPINNED: the reference's modules for this stage import `diffusers` and `conformer`, which are
not installed, and the reference holds no outputs of the stage.  These tests hold the restatement to properties the
reference's source implies: the tensor inventory is the module tree of decoder.py:100-217, the convolutions are causal
while attention is not, the guidance combination is linear in the two estimates, and the time grid is the cosine one."""
import math

import pytest
import torch

from chatterbox_vllm_amd import cfm_weights as CW
from oracle import cfm_oracle as O


@pytest.fixture(scope="module")
def weights():
    return {k: v.double() for k, v in CW.synthetic_state_dict().items()}


def _rows(inp, L):
    x = inp["noise"].double()
    z = torch.zeros_like(x)
    return (torch.cat([x, x]), torch.cat([inp["mu"].double(), z]), torch.cat([inp["spks"].double(), torch.zeros(1, 80, dtype=torch.float64)]),
            torch.cat([inp["cond"].double(), z]))


def test_inventory_is_the_reference_module_tree():
    inv = dict(CW.tensor_inventory())
    # decoder.py:130-216: 1 down + 12 mid + 1 up stages, each a resnet block and 4 transformer blocks
    n_resnets = sum(1 for k in inv if k.endswith("res_conv.weight"))
    n_tf = sum(1 for k in inv if k.endswith("attn1.to_q.weight"))
    assert n_resnets == 14 and n_tf == 56
    assert inv["decoder.estimator.down_blocks.0.0.block1.block.0.weight"] == (256, 320, 3)
    assert inv["decoder.estimator.up_blocks.0.0.block1.block.0.weight"] == (256, 512, 3)
    assert inv["decoder.estimator.mid_blocks.11.1.3.ff.net.0.proj.weight"] == (1024, 256)
    assert inv["decoder.estimator.final_proj.weight"] == (80, 256, 1)
    assert "decoder.estimator.mid_blocks.0.2.weight" not in inv          # mid stages have no trailing convolution
    assert sum(math.prod(s) for s in inv.values()) == 71_302_480


def test_time_embedding_layout():
    # matcha/decoder.py:20-29: sines of 1000 t f_i, then cosines, f_0 = 1 and f_159 = 1e-4
    e = O.sinusoidal_pos_emb(torch.tensor([0.25], dtype=torch.float64))
    assert e.shape == (1, 320)
    assert abs(e[0, 0].item() - math.sin(250.0)) < 1e-9 and abs(e[0, 160].item() - math.cos(250.0)) < 1e-9
    assert abs(e[0, 159].item() - math.sin(250.0 * 1e-4)) < 1e-7


def test_resnet_blocks_are_causal_and_attention_is_not(weights):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 320, 24, generator=g, dtype=torch.float64)
    temb = torch.randn(1, 1024, generator=g, dtype=torch.float64)
    p = "decoder.estimator.down_blocks.0.0."
    y = O.resnet_block(weights, p, x, temb)
    x2 = x.clone()
    x2[:, :, 16:] += 1.0
    y2 = O.resnet_block(weights, p, x2, temb)
    assert torch.allclose(y[:, :, :16], y2[:, :, :16], atol=1e-12, rtol=0) and (y[:, :, 16:] - y2[:, :, 16:]).abs().max() > 1e-3
    h = torch.randn(1, 24, 256, generator=g, dtype=torch.float64)
    h2 = h.clone()
    h2[:, 20] += torch.randn(256, generator=g, dtype=torch.float64)      # a constant shift would vanish in the LayerNorm
    tp = "decoder.estimator.down_blocks.0.1.0."
    assert (O.transformer_block(weights, tp, h) - O.transformer_block(weights, tp, h2))[:, :8].abs().max() > 1e-6


def test_guidance_and_time_grid(weights):
    L = 16
    inp = CW.synthetic_inputs(L, 6)
    ts = CW.t_span(10)
    assert ts.dtype == torch.float32 and ts[0] == 0 and abs(ts[-1].item() - 1.0) < 1e-6
    assert abs(ts[5].item() - (1 - math.cos(0.25 * math.pi))) < 1e-6
    # one step by hand: x1 = z + dt * (1.7 d_cond - 0.7 d_uncond)  (flow_matching.py:125-127)
    x, mu, sp, c = _rows(inp, L)
    t0 = torch.zeros(2, dtype=torch.float64)
    d = O.estimator(weights, x, mu, t0, sp, c)
    want = x[0:1] + (ts[1] - ts[0]).double() * (1.7 * d[0:1] - 0.7 * d[1:2])
    sol = []
    O.solve_euler(weights, inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"], dtype=torch.float64, sol=sol)
    assert len(sol) == 10 and torch.allclose(sol[0], want, atol=1e-12)
    # the two rows differ only through mu, spks and cond
    d_same = O.estimator(weights, x, torch.cat([mu[1:], mu[1:]]), t0, torch.cat([sp[1:], sp[1:]]), torch.cat([c[1:], c[1:]]))
    assert torch.allclose(d_same[0], d_same[1], atol=1e-12) and torch.allclose(d_same[1], d[1], atol=1e-12)


def test_fp32_restatement_stays_close_to_fp64():
    sd = CW.synthetic_state_dict()
    inp = CW.synthetic_inputs(40, 12)
    a = O.cfm_forward(sd, inp["mu"], inp["spks"], inp["cond"], inp["noise"], n_timesteps=4)
    b = O.cfm_forward(sd, inp["mu"], inp["spks"], inp["cond"], inp["noise"], n_timesteps=4, dtype=torch.float64)
    assert (a.double() - b).abs().max().item() < 5e-5 * b.abs().max().item()


def test_flop_count():
    # per frame and row: 14 resnets, 56 transformer blocks (attention grows with the utterance), 3 tail / final convolutions
    f = CW.estimator_flop(1000)
    assert 4.5e11 < f < 5.5e11
