#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats, then separate PMC passes, of the bench command.
# usage: tools/profile_gpu.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-latency --no-convergence $*"
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- $BENCH > $OUT/trace_bench.log 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $BENCH --calibrate-hbm > $OUT/pmc_fetch_bench.log 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $BENCH --calibrate-hbm > $OUT/pmc_write_bench.log 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq -o run -- $BENCH > $OUT/pmc_sq_bench.log 2>&1
find $OUT -name "*.csv" | head -30
# the raw per-dispatch traces are large: keep stats + counter CSVs, summarised by tools/prof_summarize.py
find $OUT -type f -size +6M -print -delete
du -sh $OUT
