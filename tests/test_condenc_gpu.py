"""GPU conditioning encoder (csrc/condenc.hip through t3_condenc_*) against the outputs of the reference's
own T3CondEnc (tests/golden/ref_plugin_fixture.npz, captured by make_reference_fixtures.py) and, for other
prompt lengths, against the CPU checker oracle/cond_enc_oracle.py.

fp32 throughout; the only difference allowed is fp32 summation order: |got - ref| <= 2e-5 + 2e-5 |ref|
on rows whose values are O(0.1) (observed ~1e-6)."""
import os

import numpy as np
import pytest
import torch

from chatterbox_vllm_amd import cond_enc as CE
from oracle import cond_enc_oracle as CEO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def weights():
    return {k[len("cond_enc."):]: v for k, v in CE.synthetic_weights(1234).items()}


def test_matches_reference_t3condenc_outputs(golden_dir, weights):
    d = np.load(os.path.join(golden_dir, "ref_plugin_fixture.npz"))
    g = torch.Generator().manual_seed(31)
    spk = torch.randn(1, 256, generator=g)
    prompt = torch.randn(150, 1024, generator=g) * 0.05
    enc = CE.CondEncoder(weights)
    for e in (0.5, 0.8):
        got = enc.encode(spk, prompt, e)
        ref = torch.from_numpy(d[f"condenc_out_{e}"])
        assert got.shape == (34, 1024)
        assert torch.allclose(got, ref, atol=2e-5, rtol=2e-5), (got - ref).abs().max()
    assert torch.allclose(enc.emotion_row(0.3), torch.from_numpy(d["condenc_emotion_row_0.3"]), atol=1e-7)
    # same voice twice: same bits
    assert torch.equal(enc.encode(spk, prompt, 0.5), enc.encode(spk, prompt, 0.5))
    enc.close()


@pytest.mark.parametrize("n_prompt", [1, 37, 150, 333])
def test_other_prompt_lengths_and_weights_match_checker(n_prompt):
    g = torch.Generator().manual_seed(500 + n_prompt)
    w = {k[len("cond_enc."):]: v for k, v in CE.synthetic_weights(77).items()}
    # non-trivial LayerNorm gains and biases everywhere
    for k in w:
        if k.endswith("bias"):
            w[k] = torch.randn(w[k].shape, generator=g) * 0.05
        if k.endswith("norm.weight"):
            w[k] = 1 + 0.1 * torch.randn(w[k].shape, generator=g)
    spk = torch.randn(256, generator=g)
    prompt = torch.randn(n_prompt, 1024, generator=g) * 0.3
    enc = CE.CondEncoder(w)
    got = enc.encode(spk, prompt, 0.65)
    enc.close()
    ref = CEO.encode(w, spk, prompt, 0.65)
    assert torch.allclose(got, ref, atol=2e-5, rtol=2e-5), (got - ref).abs().max()


def test_missing_tensor_is_reported(weights):
    w = dict(weights)
    del w["perceiver.attn.to_k.bias"]
    with pytest.raises(KeyError):
        CE.CondEncoder(w)


def test_facade_builds_the_prefix_on_the_gpu(weights):
    """ChatterboxTTS.get_audio_conditionals / update_exaggeration (tts.py:215-257) through the GPU encoder."""
    from chatterbox_vllm_amd.tts import ChatterboxTTS
    g = torch.Generator().manual_seed(9)
    conds = {"t3": {"speaker_emb": torch.randn(1, 256, generator=g),
                    "cond_prompt_speech_tokens": torch.randint(0, 6561, (1, 150), generator=g)}, "gen": {"k": 1}}
    tts = ChatterboxTTS.from_synthetic(n_layers=2, max_model_len=128, max_batch_size=2, default_conds=conds)
    ref_dict, cond = tts.get_audio_conditionals(None)
    assert ref_dict == {"k": 1} and cond.shape == (34, 1024) and cond.device.type == "cpu"
    sd_w = {k: v.float() for k, v in tts.cond_enc_weights.items()}
    emb = tts._speech_emb[conds["t3"]["cond_prompt_speech_tokens"][0]] + tts._speech_pos_emb[:150]
    assert torch.allclose(cond, CEO.encode(sd_w, conds["t3"]["speaker_emb"], emb, 0.5), atol=2e-5, rtol=2e-5)
    hot = tts.update_exaggeration(cond, 0.9)
    assert torch.equal(hot[:-1], cond[:-1]) and torch.allclose(hot[-1], CEO.emotion_row(sd_w, 0.9), atol=1e-7)
    tts.shutdown()
