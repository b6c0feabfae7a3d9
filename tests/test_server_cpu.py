"""The speech endpoint's pure helpers (chatterbox_vllm_amd/server.py) against outputs of the reference's own functions in
openai_tts_server.py, captured by tests/golden/make_server_fixtures.py into ref_server_fixture.json; and the HTTP shell
around a stand-in engine (no GPU)."""
import base64
import io
import json
import os
import wave

import pytest
import torch

from chatterbox_vllm_amd import server as S


@pytest.fixture(scope="module")
def fx(golden_dir):
    with open(os.path.join(golden_dir, "ref_server_fixture.json"), encoding="utf-8") as f:
        return json.load(f)


def test_chunker_equals_the_reference_on_every_recorded_text(fx):
    assert len(fx["chunker"]) >= 70
    for c in fx["chunker"]:
        split = S.split_text_for_low_latency(c["text"], c["max_chars"])
        assert split == c["split"], (c["text"][:40], c["max_chars"])
        assert S.repair_chunk_boundaries(list(split)) == c["repaired"]
        assert S.avoid_weak_endings(list(c["repaired"])) == c["final"]
    # some of the recorded cases do exercise the boundary passes
    assert any(c["split"] != c["final"] for c in fx["chunker"])


def test_boundary_passes_on_hand_made_chunk_lists(fx):
    for b in fx["boundaries"]:
        assert S.repair_chunk_boundaries(list(b["chunks"])) == b["repaired"], b["chunks"]
        assert S.avoid_weak_endings(list(b["chunks"])) == b["weak"], b["chunks"]


def test_stream_plan_equals_the_reference(fx):
    for p in fx["plans"]:
        assert S.plan_chunks(p["text"], p["first_chunk_chars"], p["chunk_chars"]) == p["chunks"], p["text"][:40]
    assert S.plan_chunks("") == [] and S.plan_chunks("   ") == []


def test_chunks_respect_the_limit_and_keep_every_word():
    text = ("It was the best of times, it was the worst of times; it was the age of wisdom: it was the age of foolishness! "
            "Was it the epoch of belief? It was the epoch of incredulity... " * 3)
    for limit in (30, 57, 120):
        chunks = S.split_text_for_low_latency(text, limit)
        assert all(0 < len(c) <= limit for c in chunks)
        assert " ".join(chunks).split() == [w for w in text.replace("...", ".").split()]


def test_language_fallback_equals_the_reference(fx):
    for c in fx["languages"]:
        assert S.detect_language(c["text"]) == c["lang"], c["text"][:20]


def test_pcm_and_wav_bytes_equal_the_reference(fx):
    a = fx["audio"]
    g = torch.Generator().manual_seed(a["seed"])
    wav = (torch.randn(1, a["n"], generator=g) * a["scale"]).float()
    wav[0, :4] = torch.tensor(a["head"])
    pcm = S.float32_to_pcm16_bytes(wav)
    assert pcm == base64.b64decode(a["pcm16_b64"])
    assert S.float32_to_pcm16_bytes(wav[0]) == pcm and S.float32_to_pcm16_bytes(wav.tolist()) == pcm
    assert S.pack_wav_bytes(wav, 24000) == base64.b64decode(a["wav_b64"])
    for ms, lengths in a["frame_lengths"].items():
        assert [len(f) for f in S.frame_chunk_bytes(pcm, 24000, int(ms))] == lengths
    assert b"".join(S.frame_chunk_bytes(pcm, 24000, 20)) == pcm


def test_tail_ramp():
    wav = torch.ones(1, 1000)
    out = S.tail_ramp(wav, 24000)
    assert torch.equal(out[:, :880], wav[:, :880]) and out[0, -1].item() == pytest.approx(0.95) and out[0, 880].item() == 1.0
    assert torch.equal(wav, torch.ones(1, 1000))                    # the input is left alone
    short = torch.ones(1, 120)
    assert S.tail_ramp(short, 24000) is short                      # 120 samples = the ramp's length: untouched (:514)


class _StandInEngine:
    """Duck-typed engine: a waveform whose samples encode the chunk (length 240 per character, value by call order)."""
    sr = 24000
    max_model_len = 1000
    variant = "english"
    s3gen = object()

    def __init__(self):
        self.calls = []
        self.shut = False

    def get_audio_conditionals(self, path=None):
        return {"ref": path}, torch.zeros(34, 1024)

    def _wave(self, text, steps):
        return torch.full((1, 240 * len(text)), 0.001 * steps + 0.01 * (len(text) % 7))

    def generate(self, prompts, audio_prompt_path=None, language_id="en", temperature=0.8, exaggeration=0.5,
                 diffusion_steps=10, max_tokens=1000, top_p=0.8, **kw):
        if language_id == "xx":
            raise ValueError("Unsupported language_id 'xx'")
        if prompts == "boom":
            raise RuntimeError("device lost")
        self.calls.append(("generate", prompts, diffusion_steps, max_tokens, language_id))
        return [self._wave(prompts, diffusion_steps)]

    def generate_with_conds(self, prompts, s3gen_ref, cond_emb, language_id="en", temperature=0.8, exaggeration=0.5,
                            max_tokens=1000, top_p=1.0, per_prompt_max_tokens=False, tokens_only=False, per_prompt_clamp=False,
                            **kw):
        assert tokens_only and per_prompt_max_tokens and top_p == 0.8
        assert per_prompt_clamp == isinstance(language_id, list)
        self.calls.append(("tokens", list(prompts), max_tokens, language_id))
        return [torch.tensor([ord(ch) for ch in p]) for p in prompts]

    def render_waveforms(self, toks, ref, diffusion_steps=10):
        self.calls.append(("render", [len(t) for t in toks], diffusion_steps))
        return [self._wave("".join(chr(int(v)) for v in t), diffusion_steps) for t in toks]

    def shutdown(self):
        self.shut = True


TEXT = "Hello there, friend. This is the second sentence of the request; it is longer than the first! And a third one follows."


def test_service_streams_first_chunk_alone_then_one_batched_token_call():
    eng = _StandInEngine()
    svc = S.SpeechService(eng)
    chunks = S.plan_chunks(TEXT, 30, 60)
    assert len(chunks) >= 3
    it = svc.stream_waveforms(TEXT, diffusion_steps=7, first_chunk_diff_steps=3, first_chunk_chars=30, chunk_chars=60)
    first = next(it)
    assert eng.calls == [("generate", chunks[0], 3, 1000, "en")]      # nothing else has run when the first audio is out
    assert first.shape == (1, 240 * len(chunks[0])) and first[0, -1].item() == pytest.approx(0.95 * first[0, 0].item())
    rest = list(it)
    assert eng.calls[1] == ("tokens", chunks[1:], 1000, "en")
    assert eng.calls[2:] == [("render", [len(c)], 7) for c in chunks[1:]]
    assert [w.shape[1] for w in rest] == [240 * len(c) for c in chunks[1:]]
    frames = list(svc.stream_pcm_frames(TEXT, frame_ms=20, primer_silence_ms=50, diffusion_steps=7,
                                        first_chunk_diff_steps=3, first_chunk_chars=30, chunk_chars=60))
    assert frames[0] == bytes(2 * 1200)
    assert all(0 < len(f) <= 960 for f in frames[1:])
    assert b"".join(frames[1:]) == b"".join(S.float32_to_pcm16_bytes(w) for w in [first] + rest)
    assert list(svc.stream_pcm_frames("   ")) == []
    with pytest.raises(ValueError):
        S.SpeechService(type("NoChain", (), {"s3gen": None})())


def test_http_endpoint_over_a_stand_in_engine(fx):
    from fastapi.testclient import TestClient
    eng = _StandInEngine()
    app = S.create_app(S.SpeechService(eng), warm_up=True)
    # the request schema carries the reference's fields and defaults
    fields = S.SpeechRequest.model_fields
    assert {k: f.default for k, f in fields.items() if k != "input"} == fx["request_defaults"]
    with TestClient(app) as client:
        assert eng.calls[0] == ("generate", "Hello", 2, 32, "en")     # warm-up at start-up
        r = client.post("/v1/audio/speech", json={"input": "Hi all."})
        assert r.status_code == 200 and r.headers["content-type"] == "audio/wav"
        with wave.open(io.BytesIO(r.content)) as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 24000, 240 * 7)
        r = client.post("/v1/audio/speech", json={"input": "Hi all.", "response_format": "pcm"})
        assert r.headers["content-type"] == "audio/pcm;rate=24000;channels=1" and len(r.content) == 2 * 240 * 7
        r = client.post("/v1/audio/speech", json={"input": TEXT, "stream": True, "chunk_chars": 60, "frame_ms": 40,
                                                  "first_chunk_diff_steps": 3, "diffusion_steps": 7})
        assert r.status_code == 200 and r.headers["content-type"] == "audio/pcm;rate=24000;channels=1"
        assert r.headers["x-accel-buffering"] == "no" and r.headers["cache-control"] == "no-cache"
        want = b"".join(S.SpeechService(_StandInEngine()).stream_pcm_frames(
            TEXT, frame_ms=40, diffusion_steps=7, first_chunk_diff_steps=3, chunk_chars=60))
        assert r.content == want and len(want) > 0
        assert client.post("/v1/audio/speech", json={"input": "x", "language_id": "xx"}).status_code == 400
        assert client.post("/v1/audio/speech", json={"input": "boom"}).status_code == 500
        assert client.post("/v1/audio/speech", json={"input": "x", "format": "mp3"}).status_code == 422
        assert client.post("/v1/audio/speech", json={"input": "x", "frame_ms": 5}).status_code == 422
    assert not eng.shut                                               # a service handed in is left running


def test_language_fallback_applies_to_the_multilingual_engine_only():
    """openai_tts_server.py:406-411: a multilingual engine with no language in the request gets the script-based guess;
    a named language is never overridden, and the English engine is never second-guessed."""
    eng = _StandInEngine()
    svc = S.SpeechService(eng)
    svc.synthesize("Привет, мир!", language_id=None)
    assert eng.calls[-1][4] is None
    eng.variant = "multilingual"
    svc.synthesize("Привет, мир!", language_id=None)
    assert eng.calls[-1][4] == "ru"
    svc.synthesize("Привет, мир!", language_id="  ")
    assert eng.calls[-1][4] == "ru"
    svc.synthesize("Привет, мир!", language_id="fr")
    assert eng.calls[-1][4] == "fr"
    svc.synthesize("plain ascii", language_id=None)
    assert eng.calls[-1][4] is None
    # the streamed form resolves the language chunk by chunk, as the reference's per-chunk _synthesize_one calls do
    # (openai_tts_server.py:406-411, 501-509): one language for a one-script text ...
    eng.calls.clear()
    list(svc.stream_waveforms("你好，世界。今天天气很好！我们去公园散步吧。", language_id="", first_chunk_chars=10, chunk_chars=40))
    first = [c for c in eng.calls if c[0] == "generate"][0]
    rest = [c for c in eng.calls if c[0] == "tokens"][0]
    assert first[-1] == "zh" and set(rest[-1]) == {"zh"} and len(rest[-1]) == len(rest[1])
    # ... and each chunk's own for mixed scripts: an English first chunk keeps no language, the Hebrew one gets "he"
    eng.calls.clear()
    list(svc.stream_waveforms("Hello there my friend. שלום עולם, מה שלומך היום?", language_id=None, first_chunk_chars=25, chunk_chars=60))
    first = [c for c in eng.calls if c[0] == "generate"][0]
    rest = [c for c in eng.calls if c[0] == "tokens"][0]
    assert first[-1] is None and rest[-1] == [S.detect_language(c) for c in rest[1]] and "he" in rest[-1]
