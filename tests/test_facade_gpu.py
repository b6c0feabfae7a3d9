"""The public facade on the GPU: the three prompts of the reference's example-tts.py (BASELINE config 1) through
``ChatterboxTTS.generate`` and one French prompt through ``ChatterboxMultilingualTTS.generate``, on a seeded
2-layer checkpoint.  Tokenisation comes from the ids the reference's own vocabularies produced for these very
prompts (tests/golden/ref_text_fixture.json), so the test needs no vocabulary file.

Checked: the request the facade builds (prompt string, ids, batch-wide max_tokens, sampling clamp), results in
input order, the facade's tensors == post-processing of the engine's ids, and the engine's ids against the oracle
(same bar as tests/test_engine_gpu.py)."""
import json
import os

import pytest
import torch

from chatterbox_vllm_amd import weights as W
from chatterbox_vllm_amd.text_utils import batch_max_tokens
from chatterbox_vllm_amd.tts import ChatterboxTTS, postprocess_speech_ids, STOP_OFFSET_ID
from chatterbox_vllm_amd.mtl_tts import ChatterboxMultilingualTTS
from oracle.t3_oracle import T3Oracle, SamplingSpec

pytestmark = pytest.mark.gpu

EXAMPLE_PROMPTS = [
    "You are listening to a demo of the Chatterbox TTS model running on VLLM.",
    "This is a separate prompt to test the batching implementation.",
    "And here is a third prompt. It's a bit longer than the first one, but not by much.",
]


@pytest.fixture(scope="module")
def text_fx(golden_dir):
    with open(os.path.join(golden_dir, "ref_text_fixture.json")) as f:
        return json.load(f)


def _lookup_tokenizer(pairs):
    table = {s: ids for s, ids in pairs}

    def tok(prompt, **kw):
        return table[prompt]
    return tok


def _oracle_ok(oracle, ids, text, cond, cfg_scale, max_model_len):
    r = oracle.generate_one(text, cond, SamplingSpec(max_tokens=len(ids)), max_model_len=max_model_len, forced=ids,
                            return_logits=True)
    tol = 2 * (1 + 2 * cfg_scale) * 2 ** -5
    for t, tok in enumerate(ids):
        L = r["logits"][t]
        assert L[tok].item() >= L.max().item() - tol, f"step {t}: id {tok} is not the oracle's choice"


def test_example_prompts_through_chatterbox_tts(text_fx, monkeypatch):
    monkeypatch.setenv("CHATTERBOX_DETERMINISTIC", "1")            # tts.py:358-362: temperature 0, top_p 1
    monkeypatch.setenv("CHATTERBOX_CFG_SCALE", "0.5")
    tts = ChatterboxTTS.from_synthetic(tokenizer=_lookup_tokenizer(text_fx["en"]), n_layers=2, max_model_len=400,
                                       max_batch_size=3)
    assert tts.sr == 24000 and tts.get_supported_languages() == {"en": "English"}
    out = tts.generate(EXAMPLE_PROMPTS, max_tokens=48)
    assert len(out) == 3 and all(o.dtype == torch.int64 and o.dim() == 1 for o in out)
    ros = tts.last_request_outputs
    assert [ro.prompt for ro in ros] == [s for s, _ in text_fx["en"]]                   # "[START]...[STOP]", input order
    assert [ro.prompt_token_ids for ro in ros] == [ids for _, ids in text_fx["en"]]
    cap = batch_max_tokens([ro.prompt for ro in ros], 48, 400)
    assert cap == 48 and all(len(ro.outputs[0].token_ids) <= cap for ro in ros)
    for o, ro in zip(out, ros):
        assert torch.equal(o, postprocess_speech_ids(ro.outputs[0].token_ids, ro.prompt))
        assert int(o.max()) < 6561 if o.numel() else True
    cond = tts.get_audio_conditionals(None)[1]
    oracle = T3Oracle(W.synthetic_state_dict(tts.t3_config), n_layers=2, cfg_scale=0.5)
    _oracle_ok(oracle, ros[1].outputs[0].token_ids[:24], ros[1].prompt_token_ids, cond, 0.5, 400)
    # one prompt as a plain string; an unsupported language is refused (tts.py:317-322)
    one = tts.generate(EXAMPLE_PROMPTS[1], max_tokens=48)
    assert len(one) == 1 and torch.equal(one[0], out[1])
    with pytest.raises(ValueError):
        tts.generate(EXAMPLE_PROMPTS[0], language_id="fr")
    tts.shutdown()


def test_french_prompt_through_multilingual_tts(text_fx, monkeypatch):
    monkeypatch.setenv("CHATTERBOX_DETERMINISTIC", "1")
    monkeypatch.setenv("CHATTERBOX_CFG_SCALE", "0.5")
    (s, ids), = text_fx["mtl_fr"]
    seen = {}

    def tok(prompt, language_id=None, **kw):                         # mtl_tts.py:423-435: "[fr]" prefix + hint
        seen["prompt"], seen["lang"] = prompt, language_id
        return ids
    tts = ChatterboxMultilingualTTS.from_synthetic(tokenizer=tok, n_layers=2, max_model_len=400, max_batch_size=2)
    assert "fr" in tts.get_supported_languages() and len(tts.get_supported_languages()) == 23
    out = tts.generate("Bonjour, comment ça va? Ceci est le modèle de synthèse vocale multilingue Chatterbox.",
                       language_id="fr", max_tokens=30)
    assert seen == {"prompt": "[fr]" + s, "lang": "fr"}
    ro = tts.last_request_outputs[0]
    assert ro.prompt_token_ids == ids and ids[0] == text_fx["mtl_special_ids"]["[fr]"]
    assert torch.equal(out[0], postprocess_speech_ids(ro.outputs[0].token_ids, ro.prompt))
    cond = tts.get_audio_conditionals(None)[1]
    oracle = T3Oracle(W.synthetic_state_dict(tts.t3_config), n_layers=2, cfg_scale=0.5)
    _oracle_ok(oracle, ro.outputs[0].token_ids[:20], ids, cond, 0.5, 400)
    with pytest.raises(ValueError):
        tts.generate(["a", "b"], language_id=["fr"])                 # mtl_tts.py:412-413
    with pytest.raises(ValueError):
        tts.generate("a", language_id="xx")
    tts.shutdown()


def test_from_local_on_a_checkpoint_directory(tmp_path, golden_dir, monkeypatch):
    """ChatterboxTTS.from_local (reference: tts.py:91-158) on a directory holding a 2-layer seeded checkpoint as
    safetensors, a tokenizer.json (tiny BPE vocabulary of the tests) and conds.pt: the real engine is built from
    what the files hold, the default voice is encoded on the GPU from conds.pt, and generate() gives the ids that
    an engine built from the same tensors in memory gives for the same token ids and conditioning."""
    from test_host_cpu import _write_checkpoint_dir
    from chatterbox_vllm_amd.engine import T3Engine, SamplingParams
    monkeypatch.setenv("CHATTERBOX_DETERMINISTIC", "1")
    monkeypatch.setenv("CHATTERBOX_CFG_SCALE", "0.5")
    cfg, sd = _write_checkpoint_dir(str(tmp_path), golden_dir, n_layers=2)
    tts = ChatterboxTTS.from_local(str(tmp_path), max_model_len=200, max_batch_size=2)
    assert tts.t3_config.n_layers == 2 and sorted(tts.t3.skipped_tensors)[-1] == "text_head.weight"
    out = tts.generate(["Hi there.", "the inn"], max_tokens=20)
    ros = tts.last_request_outputs
    assert ros[0].prompt == "[START]Hi there.[STOP]" and ros[0].prompt_token_ids == [255, 36, 11, 2, 62, 20, 7, 55, 0]
    cond = tts.get_audio_conditionals(None)[1]
    assert cond.shape == (34, 1024) and cond.dtype == torch.float32 and cond.device.type == "cpu"
    tts.shutdown()
    eng = T3Engine(cfg, {k: v for k, v in sd.items() if not k.startswith("cond_enc.")}, max_model_len=200,
                   max_num_seqs=2, cfg_scale=0.5)
    want = eng.generate([{"prompt_token_ids": ro.prompt_token_ids, "multi_modal_data": {"conditionals": [cond]}} for ro in ros],
                        SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=len(ros[0].outputs[0].token_ids),
                                       stop_token_ids=[STOP_OFFSET_ID]))
    eng.shutdown()
    assert [w.outputs[0].token_ids for w in want] == [ro.outputs[0].token_ids for ro in ros]
    assert all(torch.equal(o, postprocess_speech_ids(ro.outputs[0].token_ids, ro.prompt)) for o, ro in zip(out, ros))


def test_text_to_waveform_with_the_token_to_waveform_chain_attached(text_fx, monkeypatch):
    """``generate`` with the S3Gen stages attached returns waveforms, as the reference's does (tts.py:489-493): per prompt
    the waveform is the chain's for exactly the speech tokens the token-only facade returns, 2 mel frames and 960 samples
    per token, 24 kHz.  The chain takes the prompts of a call through the flow-matching stage together."""
    monkeypatch.setenv("CHATTERBOX_DETERMINISTIC", "1")
    monkeypatch.setenv("CHATTERBOX_CFG_SCALE", "0.5")
    torch.manual_seed(5)                                           # the flow-matching decoder draws its fixed noise once
    tts = ChatterboxTTS.from_synthetic(tokenizer=_lookup_tokenizer(text_fx["en"]), n_layers=2, max_model_len=400,
                                       max_batch_size=3, with_s3gen=True)
    wavs = tts.generate(EXAMPLE_PROMPTS[:2], max_tokens=40, diffusion_steps=4)
    ref, _ = tts.get_audio_conditionals(None)
    s3 = tts.s3gen
    tts.s3gen = None
    toks = tts.generate(EXAMPLE_PROMPTS[:2], max_tokens=40)
    assert len(wavs) == 2 and all(t.numel() > 0 for t in toks)
    from chatterbox_vllm_amd.tts import postprocess_waveform
    # the facade hands the utterances to the chain longest first, max_batch at a time (here both in one call)
    assert s3.max_batch == 3
    order = sorted(range(2), key=lambda i: -toks[i].numel())
    again = dict(zip(order, s3.inference_batch([toks[i].cuda() for i in order], ref, n_timesteps=4)))
    for i, (wav, t) in enumerate(zip(wavs, toks)):
        assert again[i].shape == (1, 960 * t.numel())
        # the facade returns the chain's waveform after the reference's alignment cap and RMS tail trim (tts.py:495-541)
        assert wav.dtype == torch.float32 and torch.isfinite(wav).all() and 0 < wav.shape[1] <= 960 * t.numel()
        assert torch.equal(postprocess_waveform(again[i], t.numel()).cpu(), wav)
        assert not wav[0, :480].any() and wav.abs().max() > 0
    tts.s3gen = s3
    tts.shutdown()


def test_from_local_attaches_the_waveform_chain_when_the_directory_holds_s3gen(tmp_path, golden_dir, monkeypatch):
    """A checkpoint directory that also holds s3gen.safetensors (the file the reference loads into S3Gen, tts.py:146-148) and
    a conds.pt with the vocoder reference in its `gen` half: from_local builds the token -> waveform chain from the
    `flow.*` and `mel2wav.*` entries (other entries, like the reference-audio side's, are ignored) and generate() returns
    24 kHz waveforms, at most 960 samples per speech token (the tail trim may shorten them)."""
    from safetensors.torch import save_file
    from test_host_cpu import _write_checkpoint_dir
    from chatterbox_vllm_amd import cfm_weights as CW, flow_weights as FW, hift_weights as HW
    from chatterbox_vllm_amd.s3gen import S3Token2Wav
    monkeypatch.setenv("CHATTERBOX_DETERMINISTIC", "1")
    monkeypatch.setenv("CHATTERBOX_CFG_SCALE", "0.5")
    _write_checkpoint_dir(str(tmp_path), golden_dir, n_layers=2)
    s3 = {"flow." + k: v for k, v in FW.synthetic_state_dict().items()}
    s3.update({"flow." + k: v for k, v in CW.synthetic_state_dict().items()})
    s3.update({"mel2wav." + k: v for k, v in HW.synthetic_state_dict().items()})
    s3["speaker_encoder.head.weight"] = torch.zeros(4, 4)          # not taken
    save_file({k: v.contiguous() for k, v in s3.items()}, os.path.join(str(tmp_path), "s3gen.safetensors"))
    g = torch.Generator().manual_seed(3)
    conds = torch.load(os.path.join(str(tmp_path), "conds.pt"), weights_only=True)
    conds["gen"] = {"prompt_token": torch.randint(0, 6561, (1, 40), generator=g), "prompt_token_len": torch.tensor([40]),
                    "prompt_feat": 2.0 * torch.randn(1, 80, 80, generator=g) - 4.0, "embedding": torch.randn(1, 192, generator=g)}
    torch.save(conds, os.path.join(str(tmp_path), "conds.pt"))
    tts = ChatterboxTTS.from_local(str(tmp_path), max_model_len=200, max_batch_size=2)
    assert isinstance(tts.s3gen, S3Token2Wav)
    wavs = tts.generate(["Hi there."], max_tokens=20, diffusion_steps=3)
    n_tok = postprocess_speech_ids(tts.last_request_outputs[0].outputs[0].token_ids, tts.last_request_outputs[0].prompt).numel()
    assert n_tok > 0 and 0 < wavs[0].shape[1] <= 960 * n_tok and torch.isfinite(wavs[0]).all() and wavs[0].abs().max() > 0
    tts.shutdown()
