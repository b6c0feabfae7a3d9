#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from one rocprofv3 pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace ...
(…_counter_collection.csv + …_kernel_trace.csv in one directory) -> JSON on stdout.

For every kernel: dispatches, mean duration, and
  mfma_util      = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of SIMD-cycles in which
                   the matrix pipe was busy while the dispatch ran (rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs);
  tflops         = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 flop / duration (dense bf16 peak of the chip: ~2500 TFLOP/s);
  frac_of_peak   = tflops / 2500.
With SQ_INSTS_VALU_MFMA_MOPS_F32 in the pass (the fp32 vocoder), the same against the 157.3 TFLOP/s fp32 matrix peak."""
import csv, glob, json, sys
from collections import defaultdict

path = sys.argv[1]
cc = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
val = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
seen = set()
for f in cc:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:72]
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            n[k] += 1
dur = defaultdict(float)
nd = defaultdict(int)
for f in kt:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:72]
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        nd[k] += 1
out = {}
for k in sorted(val, key=lambda k: -val[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    v = val[k]
    busy, mops, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
    if busy <= 0:
        continue
    rec = {"dispatches": n[k], "mfma_busy_cycles_per_dispatch": round(busy / n[k], 1),
           "mfma_util": round(busy / (gui / 8.0 * 1024.0), 4) if gui > 0 else None}
    if nd[k]:
        s = dur[k] * 1e-9
        rec["mean_us_under_profiler"] = round(dur[k] / nd[k] / 1e3, 2)
        rec["tflops"] = round(mops * 512.0 / s / 1e12, 2)
        rec["frac_of_dense_bf16_peak"] = round(mops * 512.0 / s / 1e12 / 2500.0, 4)
        f32 = v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        if f32 > 0:                                       # fp32-input matrix instructions (peak 157.3 TFLOP/s)
            rec["tflops_f32"] = round(f32 * 512.0 / s / 1e12, 2)
            rec["frac_of_f32_matrix_peak"] = round(f32 * 512.0 / s / 1e12 / 157.3, 4)
    out[k] = rec
json.dump({"counters": "SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16 (x512 flop), GRBM_GUI_ACTIVE (sum over 8 XCDs)",
           "kernels": out}, sys.stdout, indent=1)
print()
