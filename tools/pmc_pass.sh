#!/bin/bash
# Counter passes of the bench command (separate --pmc runs, kernel-trace only): usage tools/pmc_pass.sh <tag> [bench args]
# pass 1-2: SQ issue / wait counters; pass 3-4: HBM traffic (FETCH_SIZE, WRITE_SIZE) as the MI355X guide prescribes.  Summary -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { d=$1; shift; timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $OUT/$d -o run -- python $ROOT/bench.py --no-cpu-baseline --no-latency --no-convergence --no-lean-row --steps 20 $BARGS > $OUT/$d.log 2>&1; }
BARGS="$*"
python -c "import sys, json; sys.path.insert(0, '$ROOT/parallel-ddp_amd'); import pyddp; json.dump(pyddp.build_id(), open('$OUT/build_id.json', 'w'))"      # what the counters belong to
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM
run c FETCH_SIZE
run d WRITE_SIZE
run e TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
python - <<PY > $OUT/summary.txt
import csv, collections, statistics, glob
for f in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    d=collections.defaultdict(list)
    try:
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("void ","").replace("pddp::","").split("(")[0]
            if k.startswith("k_"): d[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    except Exception as e: print(f, e); continue
    names=sorted({c for _,c in d})
    for k in sorted({k for k,_ in d}):
        print(k, " ".join(f"{c}={statistics.mean(d[(k,c)]):.4g}" for c in names if (k,c) in d))
PY
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete
# keep only the rows of the library's kernels (the copy kernels of the host API dominate the files; gpurun merges at most 64 MiB back)
for f in $OUT/*/run_counter_collection.csv; do head -1 $f > $f.tmp; grep "pddp::k_" $f >> $f.tmp; mv $f.tmp $f; done
