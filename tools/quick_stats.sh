#!/bin/bash
# quick rocprofv3 --stats of bench.py at the given args; prints the kernel table.  usage: tools/quick_stats.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/qs_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o run -- python $ROOT/bench.py --no-cpu-baseline --no-latency $* > $OUT/bench.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/run_kernel_stats.csv")):
    n=r["Name"].replace("void pddp::","").split("(")[0][:60]
    if n.startswith("k_"): print(f"{n:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.2f} pct {float(r['Percentage']):6.2f}")
PY
grep -o '"value": [0-9.]*' $OUT/bench.log | head -1
rm -f $OUT/run_kernel_trace.csv
