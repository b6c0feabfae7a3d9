#!/usr/bin/env python3
"""Times the HiFT vocoder's decode() on the GPU:  python3 tools/bench_hift.py [--batch 4] [--frames 500] [--iters 5]
Prints one JSON line: audio seconds per second, milliseconds per call by stage, and the fp32 matrix rate reached
(algorithmic FLOP of the convolutions / time) against the 157.3 TFLOP/s fp32 MFMA peak of the chip."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatterbox_vllm_amd import hift_weights as HW          # noqa: E402
from chatterbox_vllm_amd.hift import HiFTGenerator           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--decode-only", action="store_true", help="one warm-up decode and `iters` timed ones, nothing else (counter passes)")
    args = ap.parse_args()
    B, T = args.batch, args.frames
    gen = HiFTGenerator(HW.synthetic_state_dict(), max_batch=B, max_frames=T)
    mel = HW.synthetic_mel(T, seed=3, batch=B).cuda()
    f0 = gen.f0_predictor(mel)
    s = gen.source(f0, seed=1)
    gen.decode(mel, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        wav = gen.decode(mel, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    if args.decode_only:
        print(json.dumps({"batch": B, "frames": T, "decode_ms": round(dt * 1e3, 3), "decode_calls": args.iters + 1}))
        gen.shutdown()
        return
    gen.set_timing(True)
    gen.decode(mel, s)
    stage_ms = gen.last_timing()
    gen.set_timing(False)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        gen.inference(mel, seed=2)
    torch.cuda.synchronize()
    dt_inf = (time.perf_counter() - t0) / args.iters
    pre, stages, post = HW.decode_flop(T)
    flop = B * (pre + sum(stages) + post)
    out = {"batch": B, "frames": T, "audio_s_per_call": B * T * 0.02, "decode_ms": round(dt * 1e3, 3),
           "inference_ms": round(dt_inf * 1e3, 3), "audio_s_per_s": round(B * T * 0.02 / dt, 1),
           "stage_ms": [round(x, 3) for x in stage_ms], "gflop_per_call": round(flop / 1e9, 1),
           "tflops": round(flop / dt / 1e12, 2), "frac_of_fp32_mfma_peak": round(flop / dt / 157.3e12, 3),
           "stage_tflops": [round(B * f / (ms * 1e-3) / 1e12, 2) for f, ms in zip(stages, stage_ms[1:4])],
           "wav_absmax": round(wav.abs().max().item(), 4)}
    print(json.dumps(out))
    gen.shutdown()


if __name__ == "__main__":
    main()
