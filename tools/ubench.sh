#!/bin/bash
# usage: tools/ubench.sh   (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
run() { # tag grid what count
  local out=$ROOT/gpurun_out/ub_$1; mkdir -p $out
  PDDP_EVAL_GRID=$2 timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $out -o run -- python $ROOT/tools/ubench_eval.py $3 $4 > $out/log 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("$out/run_kernel_stats.csv")):
    if "plant_eval" in r["Name"]: print("$1 grid=$2 what=$3 count=$4", r["Name"].split("<")[0].replace("void pddp::",""), "avg_us", float(r["AverageNs"])/1e3)
PY
  rm -f $out/run_kernel_trace.csv
}
run lat_lg 1 4 512        # 64 serial evaluations of 8 instances in one wave
run lat_coop 1 0 64       # 64 serial evaluations in one wave (cooperative)
run thr_lg 0 4 262144     # throughput: 4096 waves x 8 iterations
run thr_coop 0 0 32768
run lat_grad 1 1 64
run thr_grad 0 1 32768
