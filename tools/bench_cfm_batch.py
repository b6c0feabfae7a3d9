"""Time the flow-matching decoder on several utterances per call (s3cfm_solve_batch) against one per call.

    python tools/bench_cfm_batch.py --frames 500 1000 1500 --batch 1 2 4 8
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_vllm_amd import cfm_weights as CW                      # noqa: E402
from chatterbox_vllm_amd.cfm_decoder import CausalConditionalCFM      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="+", default=[500, 750, 1000, 1500])
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--ragged", action="store_true", help="utterance u of a batch gets frames * (1 - 0.05 u)")
    ap.add_argument("--split-bf16", type=int, nargs="?", const=1, default=0, help="1: attention launches on split-bf16 operands; 2: linear layers too")
    args = ap.parse_args()
    cfm = CausalConditionalCFM(CW.synthetic_state_dict(), max_frames=max(args.frames), max_batch=max(args.batch),
                               split_bf16=args.split_bf16)
    ts = CW.t_span(10)
    for L in args.frames:
        d = CW.synthetic_inputs(L, L // 2)
        for n in args.batch:
            lens = [max(1, int(L * (1 - 0.05 * u))) if args.ragged else L for u in range(n)]
            rep = lambda t: torch.cat([t] * n).cuda()
            z, mu, spks, cond = rep(d["noise"]), rep(d["mu"]), rep(d["spks"]), rep(d["cond"])
            for u in range(n):
                z[u, :, lens[u]:] = 0; mu[u, :, lens[u]:] = 0; cond[u, :, lens[u]:] = 0
            cfm.solve_euler_batch(z, ts, mu, spks, cond, lens)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                cfm.solve_euler_batch(z, ts, mu, spks, cond, lens)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.iters * 1e3
            flop = sum(CW.estimator_flop(v) for v in lens) * 10
            print(json.dumps({"frames": L, "batch": n, "lens": lens if args.ragged else None, "ms": round(ms, 2),
                              "ms_per_utt": round(ms / n, 2), "audio_s_per_s": round(sum(lens) / 50 / (ms / 1e3), 1),
                              "tflops": round(flop / (ms / 1e3) / 1e12, 1)}), flush=True)
    cfm.shutdown()


if __name__ == "__main__":
    main()
