#!/bin/bash
# Hardware counters of the vocoder's convolution kernel on one layer shape: bash tools/pmc_probe.sh "8 64 64 3 1 60001 0"
SPEC=${1:-"8 64 64 3 1 60001 0"}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp1 /tmp/pp2
timeout -s KILL 60 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pp1 -o p --output-format csv -- $R/tools/_build/conv_probe $SPEC 2 || exit 1
timeout -s KILL 60 rocprofv3 --pmc SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY GRBM_GUI_ACTIVE -d /tmp/pp2 -o p --output-format csv -- $R/tools/_build/conv_probe $SPEC 2 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pp1", "/tmp/pp2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv1d" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
