#!/bin/bash
# Profiles of the HiFT vocoder, run on the GPU box from the repository root:  bash tools/profile_hift.sh r02
#   1. rocprofv3 --kernel-trace --stats over tools/bench_hift.py (8 utterances x 500 frames)
#   2. one --pmc pass with the matrix-pipe counters over the same command
# Summaries land in gpurun_out/<tag>_hift_*; copy the ones to keep into profiles/.
set -o pipefail
TAG=${1:-r00}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph_kt /tmp/ph_m
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/ph_kt -o kt --output-format csv -- \
    python3 $R/tools/bench_hift.py --batch 8 --frames 500 --iters 5 > $OUT/${TAG}_hift_bench_under_rocprof.json 2> $OUT/${TAG}_hift_rocprof_stderr.log || exit 1
python3 $R/tools/prof_summary.py /tmp/ph_kt 20 > $OUT/${TAG}_hift_kernel_stats.txt || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
    -d /tmp/ph_m -o m --output-format csv -- \
    python3 $R/tools/bench_hift.py --batch 8 --frames 500 --iters 2 > $OUT/${TAG}_hift_pmc_bench.json 2>> $OUT/${TAG}_hift_rocprof_stderr.log || exit 1
python3 $R/tools/mfma_summary.py /tmp/ph_m > $OUT/${TAG}_hift_pmc_mfma.json || exit 1
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one) over one warm-up + one decode call
rm -rf /tmp/ph_f /tmp/ph_w
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/ph_f -o f --output-format csv -- \
    python3 $R/tools/bench_hift.py --batch 8 --frames 500 --iters 1 --decode-only > $OUT/${TAG}_hift_pmc_fetch_bench.json 2>> $OUT/${TAG}_hift_rocprof_stderr.log || exit 1
python3 $R/tools/pmc_summary.py /tmp/ph_f > $OUT/${TAG}_hift_pmc_fetch.json || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/ph_w -o w --output-format csv -- \
    python3 $R/tools/bench_hift.py --batch 8 --frames 500 --iters 1 --decode-only > $OUT/${TAG}_hift_pmc_write_bench.json 2>> $OUT/${TAG}_hift_rocprof_stderr.log || exit 1
python3 $R/tools/pmc_summary.py /tmp/ph_w > $OUT/${TAG}_hift_pmc_write.json || exit 1
echo done
