#!/usr/bin/env python3
"""Micro-benchmark of the decode projection kernels through the C ABI: back-to-back launches replayed from a
HIP graph, once over ONE weight matrix (it then lives in the Infinity Cache) and once rotating over enough
copies to exceed that cache (every launch streams from HBM, as inside a decode step)."""
import argparse
import ctypes as C
import sys, os, time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_vllm_amd import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=52)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = _capi.load()
    dev = "cuda:0"
    M = a.rows
    S = lib.t3_op_slab_split()

    def chk(rc):
        _capi.check(lib, rc)

    cases = {
        # name: (N_rows_of_W, K, launcher)
        "gateup_silu N=8192 K=1024": (8192, 1024, "silu"),
        "qkv-shaped silu N=3072 K=1024": (3072, 1024, "silu"),
        "down slab N=1024 K=4096": (1024, 4096, "slab"),
        "o slab N=1024 K=1024": (1024, 1024, "slab"),
    }
    for name, (N, K, kind) in cases.items():
        bytes_w = N * K * 2
        ncopies = max(2, int(600e6 // bytes_w))
        ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(ncopies)]
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        slab = torch.empty(S, M, N, device=dev, dtype=torch.float32)
        L = 64
        res = {}
        for mode in ("hot", "cold"):
            g = torch.cuda.CUDAGraph()
            st = torch.cuda.Stream()
            with torch.cuda.graph(g, stream=st):
                h = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                for i in range(L):
                    w = ws[0] if mode == "hot" else ws[i % ncopies]
                    if kind == "silu":
                        chk(lib.t3_op_gemm_silu(_capi.ptr(x), _capi.ptr(w), _capi.ptr(out), M, N // 2, K, h))
                    else:
                        chk(lib.t3_op_gemm_slab(_capi.ptr(x), _capi.ptr(w), C.cast(_capi.ptr(slab), C.POINTER(C.c_float)),
                                                M, N, K, h))
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                g.replay()
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / (a.reps * L) * 1e6
        print(f"{name:34s} rows={M} W={bytes_w/1e6:5.1f} MB  hot {res['hot']:6.2f} us  cold {res['cold']:6.2f} us  "
              f"(cold = {bytes_w/res['cold']/1e3:6.0f} GB/s)", flush=True)
        del ws


if __name__ == "__main__":
    main()
