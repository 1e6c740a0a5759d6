#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the widening rows (end-effector cost, MPC cycles).
# usage: tools/profile_widening.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python $ROOT/tools/bench_widening.py > $OUT/trace_bench.log 2>&1
find $OUT -type f -size +6M -print -delete
du -sh $OUT
