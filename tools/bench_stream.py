"""Time to first audio and total time of one streamed speech request, in process (no HTTP), on the full-size seeded model
(30 layers, token -> waveform chain attached): the package's SpeechService (first chunk alone, all other chunks through one
batched T3 call) against the loop of the reference's server (one generate() call per chunk, openai_tts_server.py:494-528),
same chunks, same sampling, same engine.

    python tools/bench_stream.py [--text-chars 700] [--repeats 3] [--out profiles/r02_stream_latency.json]

Seeded weights never emit the stop token, so every chunk runs to its cap (tts.py:372-393: about 3.5 tokens per character
plus 16); a trained model stops earlier, which shortens both columns alike."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from chatterbox_vllm_amd import server as S                     # noqa: E402
from chatterbox_vllm_amd.tts import ChatterboxTTS                # noqa: E402

PARAGRAPH = ("The quick brown fox jumps over the lazy dog near the river bank. It was a bright cold day in April, and the clocks "
             "were striking thirteen; nobody in the street seemed to notice. She walked to the station, bought a ticket for the "
             "morning train, and waited on the platform with a cup of coffee! Where does the road go after the bridge? It bends "
             "to the north, climbs the hill behind the mill, and ends at the old lighthouse: a white tower above the rocks. "
             "Nothing else happened that day, or the next, but the story was told for years in every house of the village. ")


def char_tokenizer(prompt, **kw):
    body = prompt.replace("[START]", "").replace("[STOP]", "")
    return [255] + [3 + (ord(c) * 7) % 600 for c in body] + [0]


def timed_stream(frames):
    t0 = time.perf_counter()
    first, n_bytes = None, 0
    for f in frames:
        if first is None:
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
        n_bytes += len(f)
    torch.cuda.synchronize()
    return first, time.perf_counter() - t0, n_bytes


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--text-chars", type=int, default=700)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--layers", type=int, default=30)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    os.environ["CHATTERBOX_DETERMINISTIC"] = "1"                 # greedy: both columns produce the same bytes
    text = (PARAGRAPH * (1 + args.text_chars // len(PARAGRAPH)))[:args.text_chars].rsplit(" ", 1)[0]
    tts = ChatterboxTTS.from_synthetic(tokenizer=char_tokenizer, n_layers=args.layers, max_model_len=1000, max_batch_size=16,
                                       with_s3gen=True)
    svc = S.SpeechService(tts)
    chunks = S.plan_chunks(text, 30, 120)
    kw = dict(diffusion_steps=10, first_chunk_diff_steps=10, first_chunk_chars=30, chunk_chars=120)

    def reference_loop():
        for k, chunk in enumerate(chunks):
            wav = svc.synthesize(chunk, diffusion_steps=10)
            yield from S.frame_chunk_bytes(S.float32_to_pcm16_bytes(S.tail_ramp(wav, svc.sr)), svc.sr, 20)

    svc.warm_up()
    rows = {"service": [], "chunk_by_chunk": []}
    same = None
    for r in range(args.repeats + 1):                            # the first round warms graphs and workspaces up
        a = list(svc.stream_pcm_frames(text, frame_ms=20, **kw)) if r == 0 else None
        b = list(reference_loop()) if r == 0 else None
        if r == 0:
            same = b"".join(a) == b"".join(b)
            continue
        rows["service"].append(timed_stream(svc.stream_pcm_frames(text, frame_ms=20, **kw)))
        rows["chunk_by_chunk"].append(timed_stream(reference_loop()))
    rec = {"text_chars": len(text), "chunks": [len(c) for c in chunks], "layers": args.layers, "repeats": args.repeats,
           "same_bytes": same, "weights": "seeded synthetic, every chunk runs to its token cap"}
    for name, rs in rows.items():
        first = sorted(x[0] for x in rs)[len(rs) // 2]
        total = sorted(x[1] for x in rs)[len(rs) // 2]
        audio_s = rs[0][2] / 2 / svc.sr
        rec[name] = {"first_audio_ms": round(1e3 * first, 1), "total_s": round(total, 3), "audio_s": round(audio_s, 2),
                     "audio_s_per_s": round(audio_s / total, 1)}
    rec["total_speedup"] = round(rec["chunk_by_chunk"]["total_s"] / rec["service"]["total_s"], 2)
    svc.shutdown()
    line = json.dumps(rec, indent=1)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")
