"""The speech endpoint (chatterbox_vllm_amd/server.py) over the real engine with the token -> waveform chain attached, on
seeded weights: a streamed request returns, byte for byte, what the reference's server returns for it when it takes the
chunks one after the other through ``generate`` (openai_tts_server.py:494-528), although here all chunks but the first
share one batched T3 call; a non-streamed request returns ``generate``'s waveform as a WAV file."""
import io
import wave

import pytest
import torch

from chatterbox_vllm_amd import server as S
from chatterbox_vllm_amd.text_utils import punc_norm
from chatterbox_vllm_amd.tts import ChatterboxTTS

pytestmark = pytest.mark.gpu

TEXT = ("Hello there, friend. This is the second sentence of the request; it is longer than the first! "
        "And a third one follows, to fill the batch.")


def _char_tokenizer(prompt, **kw):
    """[START] + one id per character + [STOP] in the English vocabulary's range (ids 3..602)."""
    body = prompt.replace("[START]", "").replace("[STOP]", "")
    return [255] + [3 + (ord(c) * 7) % 600 for c in body] + [0]


def test_streamed_request_equals_chunk_by_chunk_generate_calls(monkeypatch):
    from fastapi.testclient import TestClient
    monkeypatch.setenv("CHATTERBOX_DETERMINISTIC", "1")            # greedy: results do not depend on batching
    monkeypatch.setenv("CHATTERBOX_CFG_SCALE", "0.5")
    torch.manual_seed(11)                                          # the flow-matching decoder draws its fixed noise once
    tts = ChatterboxTTS.from_synthetic(tokenizer=_char_tokenizer, n_layers=2, max_model_len=400, max_batch_size=4,
                                       with_s3gen=True)
    svc = S.SpeechService(tts)
    chunks = S.plan_chunks(TEXT, 30, 60)
    assert len(chunks) == 4
    with TestClient(S.create_app(svc, warm_up=False)) as client:
        r = client.post("/v1/audio/speech", json={"input": TEXT, "stream": True, "chunk_chars": 60, "frame_ms": 20,
                                                  "first_chunk_diff_steps": 2, "diffusion_steps": 3,
                                                  "primer_silence_ms": 10})
        assert r.status_code == 200 and r.headers["content-type"] == "audio/pcm;rate=24000;channels=1"
        # the batched token call gave every chunk its own cap and ran them together
        ros = tts.last_request_outputs
        assert [ro.prompt for ro in ros] == ["[START]" + punc_norm(c) + "[STOP]" for c in chunks[1:]]
        lens = [len(ro.outputs[0].token_ids) for ro in ros]
        assert len(set(lens)) > 1 and all(ro.outputs[0].finish_reason == "length" for ro in ros)
        want = bytes(2 * 240)
        for k, chunk in enumerate(chunks):                             # what the reference's loop does, one call per chunk
            wav = tts.generate(chunk, max_tokens=tts.max_model_len, diffusion_steps=2 if k == 0 else 3)[0]
            if k:
                assert len(tts.last_request_outputs[0].outputs[0].token_ids) == lens[k - 1]
            assert wav.shape[0] == 1 and wav.shape[1] > 2400 and torch.isfinite(wav).all() and wav.abs().max() > 0
            want += S.float32_to_pcm16_bytes(S.tail_ramp(wav, 24000))
        assert r.content == want
        # not streamed: the whole text in one generate call, as a WAV file
        r = client.post("/v1/audio/speech", json={"input": chunks[0], "diffusion_steps": 2})
        assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
        whole = tts.generate(chunks[0], max_tokens=tts.max_model_len, diffusion_steps=2)[0]
        assert r.content == S.pack_wav_bytes(whole, 24000)
        with wave.open(io.BytesIO(r.content)) as w:
            assert w.getframerate() == 24000 and w.getnframes() == whole.shape[1]
        assert client.post("/v1/audio/speech", json={"input": "x", "language_id": "fr"}).status_code == 400
    svc.shutdown()


def test_multilingual_service_detects_the_language_and_streams_the_same_waveforms(monkeypatch):
    """The multilingual engine behind the service: a request without language_id gets the script-based guess
    (openai_tts_server.py:406-411), every chunk carries the "[zh]" tag and tokenization hint (mtl_tts.py:419-435), and the
    streamed waveforms equal one ``generate(chunk, language_id="zh")`` call per chunk."""
    from chatterbox_vllm_amd.mtl_tts import ChatterboxMultilingualTTS
    monkeypatch.setenv("CHATTERBOX_DETERMINISTIC", "1")
    monkeypatch.setenv("CHATTERBOX_CFG_SCALE", "0.5")
    torch.manual_seed(12)
    seen = []

    def tok(prompt, language_id=None, **kw):
        seen.append((prompt, language_id))
        body = prompt.replace("[START]", "").replace("[STOP]", "")
        return [255] + [3 + (ord(c) * 11) % 2400 for c in body] + [0]       # ids up to the multilingual vocabulary's range
    tts = ChatterboxMultilingualTTS.from_synthetic(tokenizer=tok, n_layers=2, max_model_len=400, max_batch_size=4,
                                                   with_s3gen=True)
    svc = S.SpeechService(tts)
    text = "你好，世界。今天天气很好！我们去公园散步吧；然后去吃午饭、喝茶。"
    chunks = S.plan_chunks(text, 10, 40)
    assert len(chunks) >= 2
    got = list(svc.stream_waveforms(text, language_id="", temperature=0.4, diffusion_steps=2, first_chunk_diff_steps=2,
                                    first_chunk_chars=10, chunk_chars=40))
    assert len(got) == len(chunks)
    assert all(p.startswith("[zh][START]") and lang == "zh" for p, lang in seen)
    for chunk, wav in zip(chunks, got):
        want = tts.generate(chunk, language_id="zh", temperature=0.4, max_tokens=tts.max_model_len, diffusion_steps=2)[0]
        assert torch.equal(wav, S.tail_ramp(want, 24000)) and wav.shape[1] > 0
    svc.shutdown()
