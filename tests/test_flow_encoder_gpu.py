"""Token -> mu half of S3Gen's flow stage on the GPU (C ABI include/s3flow_hip.h) against what the reference's own
CausalMaskedDiffWithXvec.inference hands to its decoder (tests/golden/flow_encoder_fixture.npz) and against the CPU
oracle (oracle/flow_encoder_oracle.py, held to the same fixture by tests/test_flow_encoder_oracle.py).  fp32 on both
sides; mu has values of order one (|mu| up to 4.4), tolerances are absolute."""
import os

import numpy as np
import pytest
import torch

from chatterbox_vllm_amd import flow_weights as FW

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(ROOT, "tests", "golden", "flow_encoder_fixture.npz"))


@pytest.fixture(scope="module")
def enc():
    from chatterbox_vllm_amd.flow_encoder import FlowEncoder
    e = FlowEncoder(FW.synthetic_state_dict(), max_tokens=260)
    yield e
    e.shutdown()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decoder_inputs_match_the_reference(enc, fx, tag):
    """mu, the speaker vector and the condition tensor against the reference's (2e-5 on values up to 4.4: ten blocks of
    fp32 sums in another order; observed ~2e-6)."""
    n_prompt, n_new = int(fx[f"{tag}_n_prompt"]), int(fx[f"{tag}_n_new"])
    prompt = FW.synthetic_tokens(n_prompt, seed=3).view(1, -1)
    new = FW.synthetic_tokens(n_new, seed=4).view(1, -1)
    got = enc.decoder_inputs(new, prompt, torch.from_numpy(fx[f"{tag}_prompt_feat"]), torch.from_numpy(fx[f"{tag}_xvec"]))
    ref_mu = torch.from_numpy(fx[f"{tag}_mu"])
    assert got["mu"].shape == ref_mu.shape
    err = (got["mu"].cpu() - ref_mu).abs().max().item()
    print(f"mu vs reference ({tag}): max |err| {err:.3g} on |mu| up to {ref_mu.abs().max().item():.3g}")
    assert err <= 2e-5
    assert (got["spks"].cpu() - torch.from_numpy(fx[f"{tag}_spks"])).abs().max().item() <= 1e-6
    assert torch.equal(got["cond"].cpu(), torch.from_numpy(fx[f"{tag}_cond"]))
    assert got["mask"].shape == (1, 1, ref_mu.shape[2]) and bool(got["mask"].all())


def test_encoder_output_before_the_projection(enc, fx):
    tok = torch.cat([FW.synthetic_tokens(7, seed=3), FW.synthetic_tokens(30, seed=4)])
    enc.token_to_mu(tok)
    h = enc.last_hidden(37).cpu()
    ref = torch.from_numpy(fx["a_encoder_out"])[0]
    assert (h - ref).abs().max().item() <= 2e-5


def test_longer_utterance_against_the_fp64_oracle(enc):
    """250 tokens (500 mel frames): several tiles in every launch, attention rows longer than one workgroup pass."""
    from oracle import flow_encoder_oracle as FO
    tok = FW.synthetic_tokens(250, seed=12)
    tok[5] = -3                                   # clamped to 0, as flow.py:250 does
    mu = enc.token_to_mu(tok).cpu()[0].t()
    w64 = {k: v.double() for k, v in FW.synthetic_state_dict().items()}
    want = FO.token_to_mu(w64, tok, dtype=torch.float64)
    err = (mu.double() - want).abs().max().item()
    print(f"mu vs fp64 oracle: max |err| {err:.3g} on |mu| up to {want.abs().max().item():.3g}")
    assert err <= 2e-5
    # one token, and a repeat of the long call: same bits
    assert enc.token_to_mu(tok[:1]).shape == (1, 80, 2)
    assert torch.equal(enc.token_to_mu(tok).cpu()[0].t(), mu)


def test_workspace_limit_is_an_error(enc):
    with pytest.raises(ValueError):
        enc.token_to_mu(FW.synthetic_tokens(261))


def test_token_to_waveform_chain_around_a_callers_decoder(fx):
    """S3Token2Wav: encoder -> (caller's decoder) -> vocoder -> trim fade, with a stand-in decoder that returns a fixed
    function of its inputs: the decoder sees the reference's tensors, the waveform is the vocoder's for the returned mel
    with the first 20 ms silenced and the next 20 ms faded in (s3gen.py:247-251, 313-315)."""
    from chatterbox_vllm_amd import hift_weights as HW
    from chatterbox_vllm_amd.s3gen import S3Token2Wav
    seen = {}

    def decoder(mu, mask, spks, cond, n_timesteps):
        seen.update(mu=mu.clone(), spks=spks.clone(), cond=cond.clone(), n=n_timesteps)
        return -4.0 + 0.5 * torch.tanh(mu) + 0.0 * cond, None

    s3 = S3Token2Wav(FW.synthetic_state_dict(), HW.synthetic_state_dict(), decoder, max_tokens=64)
    n_prompt, n_new = int(fx["a_n_prompt"]), int(fx["a_n_new"])
    ref = {"prompt_token": FW.synthetic_tokens(n_prompt, seed=3).view(1, -1), "prompt_feat": torch.from_numpy(fx["a_prompt_feat"]),
           "embedding": torch.from_numpy(fx["a_xvec"])}
    new = FW.synthetic_tokens(n_new, seed=4)
    wav, src = s3.inference(new, ref, seed=3)
    assert (seen["mu"].cpu() - torch.from_numpy(fx["a_mu"])).abs().max().item() <= 2e-5 and seen["n"] == 10
    assert torch.equal(seen["cond"].cpu(), torch.from_numpy(fx["a_cond"]))
    assert wav.shape == (1, 2 * n_new * 480) and src.shape == (1, 1, 2 * n_new * 480)
    mel = (-4.0 + 0.5 * torch.tanh(seen["mu"]))[:, :, 2 * n_prompt:]
    raw, _ = s3.hift_inference(mel, seed=3)
    assert not wav[0, :480].any()
    assert torch.equal(wav[0, 960:], raw[0, 960:])
    fade = (torch.cos(torch.linspace(torch.pi, 0, 480)) + 1) / 2
    assert torch.allclose(wav[0, 480:960].cpu(), raw[0, 480:960].cpu() * fade, atol=1e-7)
    s3.shutdown()
