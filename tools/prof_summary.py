#!/usr/bin/env python3
"""Print the per-kernel table of a rocprofv3 --kernel-trace --stats run (…_kernel_stats.csv)."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
for f in files:
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{f}: total kernel time {tot/1e6:.1f} ms")
    for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
        print(f"  {r['Name'][:64]:64s} calls={int(r['Calls']):7d} avg_us={float(r['AverageNs'])/1e3:8.2f} "
              f"min_us={float(r['MinNs'])/1e3:7.2f} total_ms={float(r['TotalDurationNs'])/1e6:9.1f} {float(r['Percentage']):5.1f}%")
