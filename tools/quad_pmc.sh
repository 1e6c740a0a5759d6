#!/bin/bash
# Counter passes over the quadrotor sweeps: usage tools/quad_pmc.sh <tag> [batch] [dtype].  -> gpurun_out/quad_<tag>/summary.txt
TAG=$1; B=${2:-16384}; DT=${3:-0}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/quad_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { d=$1; shift; timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $OUT/$d -o run -- python $ROOT/tools/quad_probe.py $B $DT 3 > $OUT/$d.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT
run b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run c FETCH_SIZE WRITE_SIZE
run d SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32
python - <<PY > $OUT/summary.txt
import csv, collections, statistics, glob
for f in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("void ","").replace("pddp::","").split("(")[0]
        if k.startswith("k_"): d[(k,r["Counter_Name"])].append(float(r["Counter_Value"]))
    names=sorted({c for _,c in d})
    for k in sorted({k for k,_ in d}):
        print(k[:60], " ".join(f"{c}={statistics.mean(d[(k,c)]):.4g}" for c in names if (k,c) in d))
PY
cat $OUT/*.log | grep "^quad" | head -3; cat $OUT/summary.txt
find $OUT -name "*.csv" -delete
