#!/usr/bin/env python3
"""Per-kernel mean of one rocprofv3 --pmc counter (…_counter_collection.csv) -> JSON on stdout.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch; FETCH_SIZE on gfx950 counts a
128-byte request as 64 bytes for wide coalesced reads, so `bytes_corrected` doubles it (WRITE_SIZE is exact)."""
import csv, glob, json, sys
from collections import defaultdict

path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True) if not path.endswith(".csv") else [path]
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:72]
            c = r["Counter_Name"]
            tot[c][k] += float(r["Counter_Value"])
            cnt[c][k] += 1
out = {}
for c in tot:
    out[c] = {}
    for k in sorted(tot[c], key=lambda k: -tot[c][k]):
        mean_kib = tot[c][k] / cnt[c][k]
        b = mean_kib * 1024.0
        out[c][k] = {"dispatches": cnt[c][k], "mean_KiB": round(mean_kib, 3),
                     "bytes_corrected": round(b * (2.0 if c == "FETCH_SIZE" else 1.0), 1),
                     "total_bytes_corrected": round(b * cnt[c][k] * (2.0 if c == "FETCH_SIZE" else 1.0), 1)}
json.dump(out, sys.stdout, indent=1)
print()
