"""Timing of the flow-matching decoder's linear launches, fp32 matrix instruction against split-bf16 operands:
python3 tools/cfm_lin_probe.py [frames [form]]   (form: both, fp32 or split; 4000 frames make the launches of a call with four
utterances of 1000)"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from chatterbox_vllm_amd.cfm_decoder import linear_op  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
FORM = sys.argv[2] if len(sys.argv) > 2 else "both"
ld = (L + 63) // 64 * 64 + 64
for name, Cin, Cout, K, pad, ln, gelu in (("qkv (norm folded)", 256, 1536, 1, 0, True, False), ("ff1 (norm folded, GELU)", 256, 1024, 1, 0, True, True),
                                          ("ff2", 1024, 256, 1, 0, False, False), ("attention out", 512, 256, 1, 0, False, False),
                                          ("resnet conv 3 taps", 256, 256, 3, 2, False, False), ("resnet 1x1", 256, 256, 1, 0, False, False)):
    g = torch.Generator().manual_seed(Cin + Cout + K)
    x = torch.randn(2, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    xd = torch.zeros(2, Cin, ld, device="cuda")
    xd[:, :, :L] = x.cuda()
    kw = dict(pad=pad, gelu=gelu)
    if ln:
        kw["ln"] = (1.0 + 0.1 * torch.randn(Cin, generator=g), 0.1 * torch.randn(Cin, generator=g))
        if Cout == 1536:
            kw["v_from"] = 1024
    gf = 2.0 * 2 * Cout * Cin * K * L / 1e9
    if FORM != "both":
        _, _, us = linear_op(xd[:, :, :L], w, b, iters=50, split=FORM == "split", **kw)
        print(f"{name:26s} {FORM} {us:6.1f} us ({gf / us * 1e3:5.1f} TFLOP/s)", flush=True)
        continue
    y0, _, us0 = linear_op(xd[:, :, :L], w, b, iters=50, **kw)
    y1, _, us1 = linear_op(xd[:, :, :L], w, b, iters=50, split=True, **kw)
    print(f"{name:26s} fp32 {us0:6.1f} us ({gf / us0 * 1e3:5.1f} TFLOP/s)   split {us1:6.1f} us ({gf / us1 * 1e3:5.1f} TFLOP/s)   "
          f"max difference {(y0 - y1).abs().max().item():.2e}", flush=True)
