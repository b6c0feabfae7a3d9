#!/usr/bin/env python3
"""A/B runs of bench.py over alternative builds of the library (T3_HIP_LIB): one line per build with the
job's tokens/s and the per-launch time of every decode kernel class.
    python tools/ab_bench.py [lib.so ...]        ("default" = the in-tree library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    libs = sys.argv[1:] or ["default"]
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["T3_HIP_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "1", "--warmup", "1"],
                           env=env, capture_output=True, text=True, timeout=400)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if r.returncode != 0 or not line:
            print(f"{lib}: FAILED rc={r.returncode}\n{r.stderr[-2000:]}", flush=True)
            return 1
        d = json.loads(line[-1])
        kd = d["kernel_breakdown"]["decode"]
        per = " ".join(f"{k}={1000 * v['ms'] / max(1, v['launches']):.2f}" for k, v in kd.items())
        pf = sum(v["ms"] for v in d["kernel_breakdown"]["prefill"].values())
        print(f"{os.path.basename(lib):24s} {d['value']:9.1f} tok/s | decode us/launch: {per} | prefill {pf:.0f} ms", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
