#!/bin/bash
# Profiles of one round, run on the GPU box from the repository root:  bash tools/profile_round.sh r01
#   1. rocprofv3 --kernel-trace --stats over `python3 bench.py --no-extra-legs`: the benchmark command without the short
#      legs of the other configurations, which run other workloads through the same kernels and would blur the
#      per-kernel averages that bench.py's roofline object is compared with
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) over one pass of the same job
#   3. one --pmc pass with the matrix-pipe counters (MFMA busy cycles, bf16 MFMA ops) over a prefill-heavy leg
# Summaries land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
set -o pipefail
TAG=${1:-r00}
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w /tmp/prof_m
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- \
    python3 $R/bench.py --no-extra-legs > $OUT/${TAG}_bench_under_rocprof.json.log 2> $OUT/${TAG}_rocprof_stderr.log || exit 1
python3 $R/tools/prof_summary.py /tmp/prof_kt 40 > $OUT/${TAG}_kernel_stats.txt || exit 1
python3 $R/tools/trace_gaps.py /tmp/prof_kt > $OUT/${TAG}_trace_gaps.txt || exit 1
cp $(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv || exit 1
export T3_USE_GRAPHS=0
timeout -s KILL 500 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f --output-format csv -- \
    python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra-legs > $OUT/${TAG}_pmc_fetch_bench.json.log 2>> $OUT/${TAG}_rocprof_stderr.log || exit 1
python3 $R/tools/pmc_summary.py /tmp/prof_f > $OUT/${TAG}_pmc_fetch.json || exit 1
timeout -s KILL 500 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w --output-format csv -- \
    python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra-legs > $OUT/${TAG}_pmc_write_bench.json.log 2>> $OUT/${TAG}_rocprof_stderr.log || exit 1
python3 $R/tools/pmc_summary.py /tmp/prof_w > $OUT/${TAG}_pmc_write.json || exit 1
# prefill-heavy leg: 48 requests of the job, 16 output tokens each would be another workload; the job's own first 64
# requests with graphs off keep every prompt kernel of the benchmark in the trace
timeout -s KILL 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
    -d /tmp/prof_m -o m --output-format csv -- \
    python3 $R/bench.py --steps 1 --warmup 0 --requests 64 --no-cpu-baseline --no-roofline --no-extra-legs > $OUT/${TAG}_pmc_mfma_bench.json.log 2>> $OUT/${TAG}_rocprof_stderr.log || exit 1
python3 $R/tools/mfma_summary.py /tmp/prof_m > $OUT/${TAG}_pmc_mfma.json || exit 1
echo done
