#!/usr/bin/env python3
"""Per-kernel sums of every counter of one rocprofv3 --pmc pass, divided by the dispatches: python3 tools/pmc_generic.py DIR [filter]"""
import csv, glob, sys
from collections import defaultdict

val, n, seen = defaultdict(lambda: defaultdict(float)), defaultdict(int), set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k))
            n[k] += 1
for k in val:
    print(k, "dispatches", n[k])
    for c in sorted(val[k]):
        print(f"   {c:32s} {val[k][c] / n[k]:16.1f}")
