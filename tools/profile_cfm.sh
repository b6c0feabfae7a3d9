#!/bin/bash
# Kernel-time profile of the flow-matching decoder, run on the GPU box from the repository root:
#   bash tools/profile_cfm.sh r02 1000        (tag, mel frames)
# rocprofv3 --kernel-trace --stats over tools/bench_cfm.py; the summary lands in gpurun_out/<tag>_cfm_kernel_stats_<frames>.txt.
set -o pipefail
TAG=${1:-r00}
FRAMES=${2:-1000}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pc_kt -o kt --output-format csv -- \
    python3 $R/tools/bench_cfm.py --frames $FRAMES --iters 2 > $OUT/${TAG}_cfm_bench_under_rocprof_${FRAMES}.txt 2> $OUT/${TAG}_cfm_rocprof_stderr.log || exit 1
python3 $R/tools/prof_summary.py /tmp/pc_kt 20 > $OUT/${TAG}_cfm_kernel_stats_${FRAMES}.txt || exit 1
python3 $R/tools/prof_by_grid.py /tmp/pc_kt > $OUT/${TAG}_cfm_by_grid_${FRAMES}.txt || exit 1
echo done
