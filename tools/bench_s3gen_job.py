"""The benchmark job's chunks through the token -> waveform chain with one or several utterances in flight (bench.py's
s3gen_job leg on chunk lengths like the job's): python tools/bench_s3gen_job.py [--variants 1x1 3x1 1x8 2x4] [--chunks 154]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def job_lengths(n, seed=5):
    """Chunk lengths like the benchmark job's (44 to 695 speech tokens, mean 377)."""
    g = torch.Generator().manual_seed(seed)
    return [int(x) for x in torch.clamp(377 + 130 * torch.randn(n, generator=g), 44, 695)]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="+", default=["1x1", "3x1", "1x8", "2x4"], help="<utterances in flight>x<utterances per call>")
    ap.add_argument("--chunks", type=int, default=154)
    args = ap.parse_args()
    # "2x4" = two chains x four utterances per call; "2x4s" = attention on split-bf16 operands, "2x4ss" = every matrix launch
    variants = tuple(tuple(int(v) for v in s.rstrip("s").split("x")) + ((len(s) - len(s.rstrip("s")),) if s.endswith("s") else ())
                     for s in args.variants)
    print(json.dumps(bench.s3gen_job_leg(0, job_lengths(args.chunks), variants=variants), indent=1))
