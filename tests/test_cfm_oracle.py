"""CPU tests of the flow-matching decoder's restatement (oracle/cfm_oracle.py) and of the package's host side for it.

The restatement is pinned by tests/golden/cfm_fixture.npz (made by tests/golden/make_cfm_fixtures.py from the reference's
own decoder.py / matcha/decoder.py / matcha/transformer.py / flow_matching.py): part A holds outputs of the reference's own
arithmetic alone, part B the whole estimator and solve with the three diffusers classes as labelled stand-ins, so what stays
unpinned is the inside of those three classes.  The remaining tests hold the restatement to properties the source implies."""
import math

import os

import numpy as np
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


# ---- the restatement against outputs of the reference's own code (tests/golden/cfm_fixture.npz) ----------------------
@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfm_fixture.npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


def _close(got, want, tol):
    """|got - want| <= tol * max|want|."""
    err = (got.double() - want.double()).abs().max().item()
    assert err <= tol * want.abs().max().item(), f"{err:.3g} against a scale of {want.abs().max().item():.3g}"


def test_time_embedding_equals_the_reference(fx):
    # matcha/decoder.py:14-29 and 73-118 on this repository's seeded weights
    sd = CW.synthetic_state_dict()
    t = fx["a_posemb_t"]
    _close(O.sinusoidal_pos_emb(t), fx["a_posemb_out"], 1e-6)
    _close(O.time_embedding(sd, t), fx["a_temb_out"], 2e-6)


def test_causal_convolution_block_and_resnet_block_equal_the_reference(fx):
    # decoder.py:48-97, matcha/decoder.py:46-61.  The fixture's second row is padded from frame 29 on (mask 0): the
    # restatement is written for an all-ones mask, and the blocks are causal, so it must agree on every frame before 29.
    sd = CW.synthetic_state_dict()
    P = CW.PREFIX
    _close(O.causal_conv(sd, P + "down_blocks.0.2", fx["a_x256"]), fx["a_conv_out"], 2e-6)
    got = O.causal_block(sd, P + "final_block.", fx["a_x256"])
    _close(got[0], fx["a_block_out"][0], 2e-6)
    _close(got[1, :, :29], fx["a_block_out"][1, :, :29], 2e-6)
    assert fx["a_block_out"][1, :, 29:].abs().max() == 0
    got = O.resnet_block(sd, P + "down_blocks.0.0.", fx["a_x320"], fx["a_resnet_temb"])
    _close(got[0], fx["a_resnet_out"][0], 2e-6)
    _close(got[1, :, :29], fx["a_resnet_out"][1, :, :29], 2e-6)


def _closed_form(x, mu, t, spks, cond):
    # tests/golden/make_cfm_fixtures.py::ClosedFormEstimator with an all-ones mask
    return 0.5 * x - 0.25 * mu + torch.sin(3.0 * t)[:, None, None] + 0.125 * spks[:, :, None] * x + 0.0625 * cond


def test_euler_solve_and_forward_equal_the_reference_around_a_closed_form_estimator(fx):
    # flow_matching.py:82-133, 203-228: time grid, guidance batch (second row zeroed), rate 0.7, fp32 update order
    inp = CW.synthetic_inputs(24, 7)
    ts = CW.t_span(10)
    assert torch.equal(ts, fx["a_euler_t_span"])
    got = O.solve_euler(None, inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"], estimator_fn=_closed_form)
    _close(got, fx["a_euler_solve"], 1e-6)
    for n, temp in ((10, 1.0), (4, 0.8)):
        got = O.cfm_forward(None, inp["mu"], inp["spks"], inp["cond"], fx["a_euler_rand_noise"], n_timesteps=n,
                            temperature=temp, estimator_fn=_closed_form)
        _close(got, fx[f"a_euler_forward_{n}"], 1e-6)
    # the recorded solve does depend on which row is zeroed: conditioning both rows gives something else
    both = lambda x, mu, t, sp, c: _closed_form(x, torch.cat([mu[:1], mu[:1]]), t, torch.cat([sp[:1], sp[:1]]), torch.cat([c[:1], c[:1]]))
    other = O.solve_euler(None, inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"], estimator_fn=both)
    assert (other - fx["a_euler_solve"]).abs().max() > 0.05


def _fixture_rows(fx, tag):
    i = CW.synthetic_inputs(fx[f"{tag}_frames"], fx[f"{tag}_prompt"], seed=fx[f"{tag}_seed"])
    z = torch.zeros_like(i["mu"])
    t = torch.full((2,), fx[f"{tag}_t"], dtype=torch.float32)
    return (torch.cat([i["noise"], i["noise"]]), torch.cat([i["mu"], z]), t, torch.cat([i["spks"], torch.zeros(1, 80)]),
            torch.cat([i["cond"], z]))


@pytest.mark.parametrize("tag", ["b1", "b2"])
def test_whole_estimator_equals_the_reference_module_tree(fx, tag):
    # ConditionalDecoder.forward, decoder.py:233-317 (the three diffusers classes as the fixture's labelled stand-ins)
    sd = CW.synthetic_state_dict()
    x, mu, t, sp, c = _fixture_rows(fx, tag)
    _close(O.estimator(sd, x, mu, t, sp, c), fx[f"{tag}_estimator_out"], 2e-5)
    w = {k: v.double() for k, v in sd.items()}
    _close(O.estimator(w, x.double(), mu.double(), t.double(), sp.double(), c.double()), fx[f"{tag}_estimator_out"], 2e-5)


def test_ten_step_solve_equals_the_reference(fx):
    sd = CW.synthetic_state_dict()
    i = CW.synthetic_inputs(40, 12)
    _close(O.cfm_forward(sd, i["mu"], i["spks"], i["cond"], i["noise"]), fx["b_forward_10"], 5e-5)
