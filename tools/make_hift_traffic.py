#!/usr/bin/env python3
"""profiles/hift_pmc_traffic.json from the two --pmc pass summaries of tools/profile_hift.sh:
   python3 tools/make_hift_traffic.py gpurun_out/r02 > profiles/hift_pmc_traffic.json
HBM bytes per decode() call of the vocoder (8 utterances x 500 frames) = sum over the convolution launches of
2 x FETCH_SIZE + WRITE_SIZE, divided by the decode calls of the pass; tied to the kernel sources by hash."""
import hashlib, json, os, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "chatterbox_vllm_amd", "csrc_s3gen")
    for name in ("hift_engine.hip", "hift_kernels.hip"):     # the vocoder's kernels and their launch logic
        if True:
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


f = json.load(open(tag + "_hift_pmc_fetch.json"))["FETCH_SIZE"]
w = json.load(open(tag + "_hift_pmc_write.json"))["WRITE_SIZE"]
calls = json.loads(open(tag + "_hift_pmc_fetch_bench.json").read().strip().splitlines()[-1])["decode_calls"]
fetch = sum(v["total_bytes_corrected"] for k, v in f.items() if "conv1d_kernel" in k) / calls
write = sum(v["total_bytes_corrected"] for k, v in w.items() if "conv1d_kernel" in k) / calls
json.dump({"kernel_source_sha16": source_sha(), "workload": "decode() of 8 utterances x 500 frames",
           "conv_launch_fetch_bytes_per_call": round(fetch), "conv_launch_write_bytes_per_call": round(write),
           "bytes_per_call": round(fetch + write),
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950 counts 128-B requests "
                     "as 64 B), summed over the conv1d_kernel launches of the pass and divided by its decode calls"}, sys.stdout, indent=1)
print()
