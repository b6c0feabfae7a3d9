"""Timing of the flow-matching decoder's attention launch, fp32 matrix instruction against split-bf16 operands, on the GPU:
python3 tools/cfm_split_probe.py [frames ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from chatterbox_vllm_amd.cfm_decoder import attention_op  # noqa: E402

for L in [int(a) for a in sys.argv[1:]] or [500, 1000, 1100, 2000]:
    g = torch.Generator().manual_seed(L)
    q, k, v = (torch.randn(2, 512, L, generator=g).cuda() for _ in range(3))
    o32, us32 = attention_op(q, k, v, iters=50)
    osp, (us_att, us_split) = attention_op(q, k, v, iters=50, split=True)
    qh, kh, vh = (t.double().view(2, 8, 64, L) for t in (q, k, v))
    want = (vh @ torch.softmax(qh.transpose(2, 3) @ kh / 8.0, dim=-1).transpose(2, 3)).reshape(2, 512, L)
    e32, esp = (o32.double() - want).abs().max().item(), (osp.double() - want).abs().max().item()
    gflop = 2 * 8 * 4.0 * L * L * 64 / 1e9
    print(f"L {L:5d}: fp32 {us32:7.1f} us ({gflop / us32 * 1e3:6.1f} TFLOP/s, err {e32:.2e})   split {us_att:7.1f} us + re-arrangement "
          f"{us_split:5.1f} us ({gflop / us_att * 1e3:6.1f} TFLOP/s, err {esp:.2e})", flush=True)
