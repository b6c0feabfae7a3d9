"""Cycle stamps of wave 0 of workgroup 0 in the split attention loop (library built with -DS3_SPLIT_TIMELINE):
S3HIFT_LIB=tools/_build/libs3hift_tl.so python3 tools/cfm_split_timeline.py [frames]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from chatterbox_vllm_amd.cfm_decoder import attention_op, _lib  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = torch.Generator().manual_seed(L)
q, k, v = (torch.randn(2, 512, L, generator=g).cuda() for _ in range(3))
attention_op(q, k, v, iters=0, split=True)
torch.cuda.synchronize()
n = 3 * ((L + 31) // 32 // 4 - 1)
buf = (C.c_longlong * n)()
lib = _lib()
lib.s3cfm_read_timeline.argtypes = [C.c_void_p, C.c_int32]
assert lib.s3cfm_read_timeline(buf, n) == 0
st = list(buf)
print("iterations", n // 3)
for i in range(0, n - 3, 3):
    a, b, c, d = st[i], st[i + 1], st[i + 2], st[i + 3]
    print(f"  scores(next) + softmax {b - a:6d}   load_k + second product {c - b:6d}   load_v + loop {d - c:6d}   total {d - a:6d}")
