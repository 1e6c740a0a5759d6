#!/bin/bash
# One ad-hoc counter pass of the headline bench: usage tools/pmc_adhoc.sh <tag> "<counters>" [bench args].  Summary -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; CTR=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc $CTR -d $OUT/a -o run -- python $ROOT/bench.py --no-cpu-baseline --no-latency --no-convergence --steps 10 "$@" > $OUT/a.log 2>&1
python - <<PY > $OUT/summary.txt
import csv, collections, statistics, glob
for f in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("void ","").replace("pddp::","").split("(")[0]
        if k.startswith("k_"): d[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    names=sorted({c for _,c in d})
    for k in sorted({k for k,_ in d}):
        print(k, " ".join(f"{c}={statistics.mean(d[(k,c)]):.4g}" for c in names if (k,c) in d))
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -delete
