#!/bin/bash
# Kernel-time profile of the flow-matching decoder, run on the GPU box from the repository root:
#   bash tools/profile_cfm.sh r02 1000        (tag, mel frames)
# rocprofv3 --kernel-trace --stats over tools/bench_cfm.py, then two counter passes; summaries land in gpurun_out/<tag>_cfm_*_<frames>.*;
# copy the ones to keep into profiles/.
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
#   2. one --pmc pass with the matrix-pipe counters over the same command
rm -rf /tmp/pc_m
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
    -d /tmp/pc_m -o m --output-format csv -- \
    python3 $R/tools/bench_cfm.py --frames $FRAMES --iters 1 > $OUT/${TAG}_cfm_pmc_bench_${FRAMES}.txt 2>> $OUT/${TAG}_cfm_rocprof_stderr.log || exit 1
python3 $R/tools/mfma_summary.py /tmp/pc_m > $OUT/${TAG}_cfm_pmc_mfma_${FRAMES}.json || exit 1
#   3. wave-state counters of single launches (tools/probe_cfm_ops.py)
rm -rf /tmp/pc_w
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
    -d /tmp/pc_w -o p --output-format csv -- python3 $R/tools/probe_cfm_ops.py --frames $FRAMES --iters 5 > $OUT/${TAG}_cfm_pmc_ops_bench_${FRAMES}.txt 2>> $OUT/${TAG}_cfm_rocprof_stderr.log || exit 1
python3 $R/tools/pmc_by_grid.py /tmp/pc_w _kernel > $OUT/${TAG}_cfm_pmc_ops_${FRAMES}.txt || exit 1
echo done
