#!/usr/bin/env python3
"""Where the time between kernels goes: reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints,
per kernel name, launches, mean duration and the mean idle gap that FOLLOWS it on the device, plus the
split of the whole span into busy time, short gaps (dependent-launch latency) and long gaps (host)."""
import csv, glob, sys
from collections import defaultdict

path = sys.argv[1]
long_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
for f in files:
    ev = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    if not ev:
        continue
    span = (max(e[1] for e in ev) - ev[0][0]) / 1e3
    busy = sum(e[1] - e[0] for e in ev) / 1e3
    dur, gap, n = defaultdict(float), defaultdict(float), defaultdict(int)
    short = longg = 0.0
    nlong = 0
    for i, (s, e, name) in enumerate(ev):
        k = name.split("(")[0][:72]
        n[k] += 1
        dur[k] += (e - s) / 1e3
        if i + 1 < len(ev):
            g = max(0.0, (ev[i + 1][0] - e) / 1e3)
            if g > long_us:
                longg += g
                nlong += 1
            else:
                short += g
                gap[k] += g
    print(f"{f}\n span {span/1e3:.1f} ms  busy {busy/1e3:.1f} ms ({100*busy/span:.1f}%)  short gaps {short/1e3:.1f} ms "
          f"({100*short/span:.1f}%)  long gaps(>{long_us:.0f}us) {longg/1e3:.1f} ms in {nlong} ({100*longg/span:.1f}%)")
    for k in sorted(n, key=lambda k: -dur[k]):
        print(f"  {k:72s} n={n[k]:7d} dur_us={dur[k]/n[k]:8.2f} gap_after_us={gap[k]/n[k]:6.2f} total_ms={dur[k]/1e3:8.1f}")
