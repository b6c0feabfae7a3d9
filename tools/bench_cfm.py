"""Time the flow-matching decoder on the GPU: python tools/bench_cfm.py [--frames 1000] [--steps 10] [--iters 3]."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_vllm_amd import cfm_weights as CW
from chatterbox_vllm_amd.cfm_decoder import CausalConditionalCFM

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, nargs="+", default=[1000])
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--split-bf16", type=int, nargs="?", const=1, default=0, help="1: attention launches on split-bf16 operands; 2: linear layers too")
args = ap.parse_args()
m = CausalConditionalCFM(CW.synthetic_state_dict(), max_frames=max(args.frames), split_bf16=args.split_bf16)
for L in args.frames:
    inp = {k: v.cuda() for k, v in CW.synthetic_inputs(L, L // 3).items()}
    ts = CW.t_span(args.steps)
    m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        out = m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    fl = CW.estimator_flop(L) * args.steps
    print(f"frames {L}: {ms:.2f} ms per solve of {args.steps} steps, {fl / ms / 1e9:.1f} TFLOP/s, {L / 50 / (ms / 1e3):.0f} audio-s/s, "
          f"finite {bool(torch.isfinite(out).all())}", flush=True)
