#!/usr/bin/env python3
"""Micro-benchmark of the prompt-row projection kernels (gemm_prompt.hip) through the C ABI: per-launch time of
back-to-back launches replayed from a HIP graph, rotating over weight copies, for several row counts."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_vllm_amd import _capi  # noqa: E402


def main():
    lib = _capi.load()
    dev = "cuda:0"
    S = lib.t3_op_slab_split()
    rows = [int(x) for x in sys.argv[1:] if x.isdigit()] or [128, 512, 1024]
    forms = [0, 1] if "--both" in sys.argv else [0]
    cases = {"gateup_silu N=8192 K=1024": (8192, 1024, "silu"), "qkv-shaped silu N=3072 K=1024": (3072, 1024, "silu"),
             "down slab N=1024 K=4096": (1024, 4096, "slab"), "o slab N=1024 K=1024": (1024, 1024, "slab")}
    for name, (N, K, kind) in cases.items():
        ncopies = 8
        ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopies)]
        for M, form in ((m, f) for m in rows for f in forms):
            _capi.check(lib, lib.t3_op_set_prompt_gemm_form(form))
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            slab = torch.empty(S, M, N, device=dev, dtype=torch.float32)
            L = 32
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            with torch.cuda.graph(g, stream=st):
                h = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for i in range(L):
                    w = ws[i % ncopies]
                    if kind == "silu":
                        _capi.check(lib, lib.t3_op_gemm_prompt_silu(_capi.ptr(x), _capi.ptr(w), _capi.ptr(out), M, N // 2, K, h))
                    else:
                        _capi.check(lib, lib.t3_op_gemm_prompt_slab(_capi.ptr(x), _capi.ptr(w),
                                                                    C.cast(_capi.ptr(slab), C.POINTER(C.c_float)), M, N, K, h))
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / (10 * L) * 1e6
            print(f"{name:32s} {'row-walking' if form else 'tile per workgroup':18s} rows={M:5d}  {us:7.2f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
