#!/usr/bin/env python3
"""Print the kernel breakdown of bench.py JSON lines found in the given log files."""
import json, sys
for f in sys.argv[1:]:
    l = [x for x in open(f) if x.startswith("{")][-1]
    d = json.loads(l)
    print(f, d["value"], "tok/s", d.get("decode"))
    for ph in ("decode", "prefill"):
        print(" ", ph)
        for k, v in d["kernel_breakdown"][ph].items():
            print("    %-18s %8.2f ms %7d launches %7.2f us/launch %8.1f GB/s" % (k, v["ms"], v["launches"], 1000 * v["ms"] / max(1, v["launches"]), v["GBps"]))
