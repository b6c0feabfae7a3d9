"""Cycle stamps inside the split attention loop (library built with -DS3_SPLIT_TIMELINE; every wave stores its stamps without a
branch, the last writer of a slot stays): S3HIFT_LIB=tools/_build/libs3hift_tl.so python3 tools/cfm_split_timeline.py [frames]"""
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
n = 64 * 64 * 8
buf = (C.c_longlong * n)()
lib = _lib()
lib.s3cfm_read_timeline.argtypes = [C.c_void_p, C.c_int32]
assert lib.s3cfm_read_timeline(buf, n) == 0
names = ["scores(next) | finish", "load_k", "second product | stats(next)", "load_v + copies"]
for it in range(0, 14):
    st = [buf[(it * 8 + k) * 64] for k in range(5)]
    if st[4] == 0:
        break
    print(f"iteration {it:2d}: " + "   ".join(f"{nm} {st[k + 1] - st[k]:5d}" for k, nm in enumerate(names)) + f"   total {st[4] - st[0]}")
