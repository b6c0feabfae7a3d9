#!/usr/bin/env python3
"""Capture golden vectors from the pure helper functions of the reference's ``openai_tts_server.py``.

Run in the build container only (needs /root/reference and fastapi):

    python tests/golden/make_server_fixtures.py

The server script imports ``chatterbox_vllm.tts`` / ``.mtl_tts`` (hence vLLM) at its top only to name the two engine
classes; both modules are replaced by empty stand-ins here, the script is loaded by path, and its text chunker
(``_split_text_for_low_latency`` :144-217, ``_repair_chunk_boundaries`` :220-259, ``_avoid_weak_endings`` :262-327, the
first-chunk plan of ``_synthesize_streaming_pcm_frames`` :473-489), ``_detect_language`` (:355-381), ``_float32_to_pcm16_bytes``
(:117-127), ``_pack_wav_bytes`` (:130-141) and ``_frame_chunk_bytes`` (:330-339) are called on the inputs below.  Only inputs
and the reference's OUTPUTS are stored, in ``ref_server_fixture.json``.
"""
import base64
import importlib.util
import json
import os
import sys
import types

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference_server():
    for name in ("chatterbox_vllm", "chatterbox_vllm.tts", "chatterbox_vllm.mtl_tts"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["chatterbox_vllm.tts"].ChatterboxTTS = type("ChatterboxTTS", (), {})
    sys.modules["chatterbox_vllm.mtl_tts"].ChatterboxMultilingualTTS = type("ChatterboxMultilingualTTS", (), {})
    spec = importlib.util.spec_from_file_location("ref_openai_tts_server", os.path.join(REF, "openai_tts_server.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _texts():
    with open(os.path.join(REF, "docs", "benchmark-text-1.txt"), encoding="utf-8") as f:
        en = f.read()
    with open(os.path.join(REF, "docs", "benchmark-text-fr-1.txt"), encoding="utf-8") as f:
        fr = f.read()
    with open(os.path.join(REF, "docs", "benchmark-text-zh-1.txt"), encoding="utf-8") as f:
        zh = f.read()
    return [
        "",
        "   ",
        "Hello.",
        "Hello there. This is a test of the chunker, and it goes on and on to the end of the line without a stop",
        "You are listening to a demo of the Chatterbox TTS model running on VLLM. This is a separate prompt to test the "
        "batching implementation. And here is a third prompt. It's a bit longer than the first one, but not by much.",
        "He went to the store and bought a loaf of bread for the children; then he walked to the park, sat on a bench and "
        "looked at the river for an hour or so. Nothing happened!",
        "Wait... what?! No: never; not once, not ever.",
        "...leading dots, then words. ,and a comma start",
        "A" * 95,
        "word " + "B" * 70 + " tail of the sentence goes here and on",
        "the of and to a in the of and to a in the of and to a in the of and to a",
        "To be or not to be, that is the question of the day for a man in a boat on the sea at dawn and at dusk",
        "no punctuation at all just a long run of words that keeps going past every limit the chunker has been given so far "
        "and then a little more",
        "Semi;colons:and;no:spaces;here:at;all:in;this:one;long:token;which:keeps;going:on",
        "שלום עולם. מה שלומך היום?",
        "مرحبا بالعالم، كيف حالك اليوم؟ أنا بخير؛ شكرا.",
        "Привет, мир! Как дела?",
        "안녕하세요. 오늘 날씨가 좋네요!",
        "こんにちは。今日はいい天気ですね！",
        "你好，世界。今天天气很好！我们去公园散步吧；然后去吃午饭、喝茶……",
        "Mixed: hello 你好 and bye.",
        en[:1500],
        en[4000:5200],
        fr[:900],
        zh[:400],
    ]


def main():
    srv = _load_reference_server()
    texts = _texts()
    cases = []
    for t in texts:
        for max_chars in (30, 60, 120):
            split = srv._split_text_for_low_latency(t, max_chars=max_chars)
            repaired = srv._repair_chunk_boundaries(list(split))
            final = srv._avoid_weak_endings(list(repaired))
            cases.append({"text": t, "max_chars": max_chars, "split": split, "repaired": repaired, "final": final})
    # the chunk plan of a streaming request (:473-489): a short first chunk, then the rest at the larger size
    plans = []
    for t in texts:
        for first, rest in ((30, 120), (0, 120), (40, 60)):
            chunks = []
            if first and first > 0:
                fchunks = srv._split_text_for_low_latency(t, max_chars=first)
                if fchunks:
                    chunks.append(fchunks[0])
                    remaining = t[len(fchunks[0]):].strip()
                    if remaining:
                        chunks.extend(srv._split_text_for_low_latency(remaining, max_chars=rest))
            else:
                chunks = srv._split_text_for_low_latency(t, max_chars=rest)
            chunks = srv._avoid_weak_endings(srv._repair_chunk_boundaries(chunks))
            plans.append({"text": t, "first_chunk_chars": first, "chunk_chars": rest, "chunks": chunks})
    # hand-made chunk lists for the two boundary passes (cases the splitter itself never produces)
    boundary_inputs = [
        ["Hello", ", world", "! and more", "  ", "—dash start"],
        ["", ".", "x"],
        ["go to", "the store"],
        ["and", "then what"],
        ["to the", "a", "b c"],
        ["We went to.", "Then home"],
        ["one of the", "of the", "best"],
        ["The", "end"],
        ["x the,", "y"],
        ["  spaced a  ", "  next  "],
        ["only"],
        [],
    ]
    boundaries = [{"chunks": c, "repaired": srv._repair_chunk_boundaries(list(c)),
                   "weak": srv._avoid_weak_endings(list(c))} for c in boundary_inputs]
    langs = [{"text": t, "lang": srv._detect_language(t)} for t in texts + ["Bonjour", "日本語のテキスト", "中文", "한글", "שלום", "Ж"]]
    # audio packing: a seeded waveform with values beyond [-1, 1]
    g = torch.Generator().manual_seed(77)
    wav = (torch.randn(1, 2500, generator=g) * 0.6).float()
    wav[0, :4] = torch.tensor([1.0, -1.0, 1.5, -2.0])
    pcm = srv._float32_to_pcm16_bytes(wav)
    wav_file = srv._pack_wav_bytes(wav, 24000)
    frames = {str(ms): [len(f) for f in srv._frame_chunk_bytes(pcm, 24000, frame_ms=ms)] for ms in (10, 20, 40, 60)}
    out = {
        "chunker": cases, "plans": plans, "boundaries": boundaries, "languages": langs,
        "audio": {"seed": 77, "n": 2500, "scale": 0.6, "head": [1.0, -1.0, 1.5, -2.0],
                  "pcm16_b64": base64.b64encode(pcm).decode(), "wav_b64": base64.b64encode(wav_file).decode(),
                  "frame_lengths": frames},
        "request_defaults": {k: f.default for k, f in srv.SpeechRequest.model_fields.items() if k != "input"},
    }
    path = os.path.join(HERE, "ref_server_fixture.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", len(cases), "chunker cases,", len(plans), "plans")


if __name__ == "__main__":
    main()
