#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two --pmc pass summaries of tools/profile_round.sh:
   python3 tools/make_pmc_traffic.py gpurun_out/r01 > profiles/pmc_traffic.json
Keys are the engine's kernel classes; bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), the
mean over every launch of that kernel in one pass of the benchmark job."""
import hashlib, json, os, sys
tag = sys.argv[1]


def kernel_source_sha():
    """sha256 (16 hex digits) over the kernel sources: bench.py quotes the traffic only for the build it was measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chatterbox_vllm_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


f = json.load(open(tag + "_pmc_fetch.json"))["FETCH_SIZE"]
w = json.load(open(tag + "_pmc_write.json"))["WRITE_SIZE"]
# gemm_stream_kernel<NB, MT, KW, KT, EPI, W8, FN>: the full-batch (33-64 row) bf16 variants of the decode step
classes = {
    "attn_decode": ["void attn_decode_kernel<1, false, false>"],
    "gemm_gateup_silu": ["void gemm_stream_kernel<2, 4, 8, 4, 2, false, -1, false>"],
    "gemm_qkv_rope": ["void gemm_stream_kernel<2, 2, 8, 4, 3, false, -1, false>"],
    "gemm_down_resid": ["void gemm_stream_kernel<2, 4, 8, 2, 4, false, -1, false>"],
    "gemm_o_resid": ["void gemm_stream_kernel<2, 4, 2, 2, 4, false, -1, false>"],
    "add_rmsnorm": ["void slab_add_rmsnorm_kernel<8>"],
    "gemm_head": ["void gemm_stream_kernel<3, 4, 8, 4, 0, false, -1, false>"],
}
out = {"kernel_source_sha16": kernel_source_sha()}
for cls, names in classes.items():
    for n in names:
        if n in f:
            fb, wb = f[n]["bytes_corrected"], w.get(n, {}).get("bytes_corrected", 0.0)
            out[cls] = {"bytes_per_launch": round(fb + wb), "fetch_bytes": round(fb), "write_bytes": round(wb),
                        "dispatches": f[n]["dispatches"], "kernel": n,
                        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over one pass of the "
                                  "default bench.py job, graphs off; FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B); "
                                  "33-64 row variants for the projections"}
json.dump(out, sys.stdout, indent=1)
print()
