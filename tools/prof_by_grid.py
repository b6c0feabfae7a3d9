#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace run (…_kernel_trace.csv) by kernel name and grid size: calls, average and total time.
Usage: prof_by_grid.py <dir or csv> [name filter]"""
import collections, csv, glob, sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
for f in files:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        a = acc[key]
        a[0] += 1
        a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for key, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {key[0]:40s} grid={key[1]:>7s},{key[2]:>4s},{key[3]:>2s} calls={n:6d} avg_us={t / n / 1e3:8.2f} total_ms={t / 1e6:8.1f}")
