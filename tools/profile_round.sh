#!/bin/bash
# Round profile on the GPU box (through gpurun): rocprofv3 --kernel-trace --stats of the bench command, the plain bench line, then the counter passes.
# usage: tools/profile_round.sh <tag>      outputs: gpurun_out/prof_<tag>/{kernel_stats.csv, bench_under_rocprof.log, bench_line.json}, gpurun_out/pmc_<tag>/
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python $ROOT/bench.py --no-cpu-baseline --no-latency --no-convergence --no-lean-row > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/run_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
cd $ROOT && python bench.py > $OUT/bench_full.log 2>&1; grep "^{" $OUT/bench_full.log | tail -1 > $OUT/bench_line.json
$ROOT/tools/pmc_pass.sh $TAG > $OUT/pmc.log 2>&1
python $ROOT/tools/kernel_times.py 1 64 256 1024 2048 4096 8192 16384 > $OUT/kernel_times.txt 2>&1
head -12 $OUT/kernel_stats.csv | cut -c1-200; cut -c1-400 $OUT/bench_line.json; tail -8 $OUT/kernel_times.txt
