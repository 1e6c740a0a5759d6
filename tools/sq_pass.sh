#!/bin/bash
# SQ counter pass of the bench command: usage tools/sq_pass.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/sq_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT -o run -- python $ROOT/bench.py --no-cpu-baseline --no-latency $* > $OUT/bench.log 2>&1
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -d $OUT/b -o run -- python $ROOT/bench.py --no-cpu-baseline --no-latency $* > $OUT/bench2.log 2>&1
python - <<PY
import csv, collections, statistics
for f in ("$OUT/run_counter_collection.csv", "$OUT/b/run_counter_collection.csv"):
    d=collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("void pddp::","").split("<")[0].split("(")[0]
            if k.startswith("k_"): d[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    except Exception as e: print(e); continue
    names=sorted({c for _,c in d})
    for k in sorted({k for k,_ in d}):
        print(k, " ".join(f"{c.replace('SQ_','')}={statistics.mean(d[(k,c)]):.3g}" for c in names if (k,c) in d))
PY
rm -f $OUT/run_kernel_trace.csv $OUT/b/run_kernel_trace.csv
