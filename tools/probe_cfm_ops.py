"""Time single launches of the flow-matching decoder's kernels: python tools/probe_cfm_ops.py [--frames 1000]."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_vllm_amd.cfm_decoder import attention_op, linear_op

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, nargs="+", default=[1000])
ap.add_argument("--iters", type=int, default=50)
args = ap.parse_args()
for L in args.frames:
    ld = (L + 63) // 64 * 64
    if (ld // 64) % 2 == 0:
        ld += 64
    for name, Cin, Cout, K, pad, gelu, v_from in (("qkv", 256, 1536, 1, 0, False, 1024), ("out", 512, 256, 1, 0, False, None),
                                                  ("w1", 256, 1024, 1, 0, True, None), ("w2", 1024, 256, 1, 0, False, None),
                                                  ("conv3", 256, 256, 3, 2, False, None), ("conv3 in 512", 512, 256, 3, 2, False, None)):
        x = torch.randn(2, Cin, ld, device="cuda")[:, :, :L]
        w = torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
        res = None if v_from or gelu else torch.randn(2, Cout, L, device="cuda")
        _, _, us = linear_op(x, w, torch.zeros(Cout), pad=pad, gelu=gelu, res=res, v_from=v_from, iters=args.iters)
        fl = 2.0 * 2 * L * Cin * Cout * K
        print(f"L {L} {name:14s} {us:8.2f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
    q = torch.randn(2, 512, L, device="cuda")
    _, us = attention_op(q, torch.randn_like(q), torch.randn_like(q), iters=args.iters)
    fl = 2.0 * 2 * 2 * L * L * 512
    print(f"L {L} {'attention':14s} {us:8.2f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
