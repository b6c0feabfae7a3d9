#!/usr/bin/env python3
"""Times the token -> mu half of the flow stage on the GPU:  python3 tools/bench_flow_encoder.py [--tokens 500]
One JSON line: milliseconds per call, audio seconds per second (a token is 40 ms), algorithmic GFLOP and TFLOP/s."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chatterbox_vllm_amd import flow_weights as FW          # noqa: E402
from chatterbox_vllm_amd.flow_encoder import FlowEncoder     # noqa: E402


def flop(T):
    d, f = FW.D_MODEL, FW.FFN
    lin = lambda n: n * (2 * d * d * 4 + 2 * d * f * 2)                     # q(x2 here), k, v, out ~ 5 d^2; w_1, w_2
    att = lambda n: 8 * (2 * n * n * 64 + 2 * n * (2 * n - 1) * 64 + 2 * n * n * 64) + 2 * (2 * n - 1) * d * d + 2 * n * d * d
    pre = 2 * d * d * (1 + 4 + 3) * T
    up = 2 * d * d * (5 + 1) * 2 * T + 2 * d * 80 * 2 * T
    return pre + up + 6 * (lin(T) + att(T)) + 4 * (lin(2 * T) + att(2 * T))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=500)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    T = args.tokens
    enc = FlowEncoder(FW.synthetic_state_dict(), max_tokens=T)
    tok = FW.synthetic_tokens(T, seed=2).cuda()
    enc.token_to_mu(tok)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        enc.token_to_mu(tok)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    print(json.dumps({"tokens": T, "mel_frames": 2 * T, "audio_s": T * 0.04, "ms_per_call": round(dt * 1e3, 3),
                      "audio_s_per_s": round(T * 0.04 / dt, 1), "gflop": round(flop(T) / 1e9, 1),
                      "tflops": round(flop(T) / dt / 1e12, 2)}))
    enc.shutdown()


if __name__ == "__main__":
    main()
