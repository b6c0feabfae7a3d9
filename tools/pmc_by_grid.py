#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters (…_counter_collection.csv) per kernel name and grid size, averaged per dispatch.
Usage: pmc_by_grid.py <dir> [name filter]"""
import collections, csv, glob, sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:36], r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key].add(r["Dispatch_Id"])
    for key in sorted(acc, key=lambda k: -len(disp[k])):
        n = len(disp[key])
        print(f"{key[0]:36s} grid={key[1]:>8s} dispatches={n:5d} " + " ".join(f"{c}={v / n:.3g}" for c, v in sorted(acc[key].items())))
