#!/bin/bash
# Counter passes (and a kernel trace) over the bench rows BESIDE the headline: BASELINE configs[1] (cart-pole), configs[4] (quadrotor, float32 / float64), configs[3]'s
# end-effector cost family at 64 and 4096 problems, the MPC control cycles.  usage (through gpurun): tools/pmc_rows.sh <tag>
# -> gpurun_out/rows_<tag>/{kernel_stats.csv, counters.txt, rows.json}
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/rows_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
python -c "import sys, json; sys.path.insert(0, '$ROOT/parallel-ddp_amd'); import pyddp; json.dump(pyddp.build_id(), open('$OUT/build_id.json', 'w'))"      # what the counters belong to
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o run -- python $ROOT/bench.py --rows > $OUT/rows.log 2>&1
cp $OUT/trace/run_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/trace
grep "^{" $OUT/rows.log | tail -1 > $OUT/rows.json
run() { d=$1; shift; timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $OUT/$d -o run -- python $ROOT/bench.py --rows > $OUT/$d.log 2>&1; }
run c FETCH_SIZE
run d WRITE_SIZE
run b SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python - <<PY > $OUT/counters.txt
import csv, collections, statistics, glob
d = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/*/run_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("pddp::", "").split("(")[0]
        if k.startswith("k_"):
            d[(k, int(r["Grid_Size"]) if "Grid_Size" in r else 0, r["Counter_Name"])].append(float(r["Counter_Value"]))
keys = sorted({(k, g) for k, g, _ in d})
print("# per kernel and launch geometry (grid size in work-items), averages per launch; HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction of the MI355X guide)")
for k, g in keys:
    get = lambda c: statistics.mean(d[(k, g, c)]) if (k, g, c) in d else None
    f, w = get("FETCH_SIZE"), get("WRITE_SIZE")
    hbm = "" if f is None or w is None else " hbm_read_MB=%.2f hbm_write_MB=%.2f" % (2 * f * 1024 / 1e6, w * 1024 / 1e6)
    rest = " ".join("%s=%.4g" % (c, get(c)) for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if get(c) is not None)
    print("%s grid=%d launches=%d%s %s" % (k[:80], g, len(d[(k, g, "FETCH_SIZE")]) or len(d[(k, g, "SQ_WAVES")]), hbm, rest))
PY
find $OUT -name "run_counter_collection.csv" -delete; find $OUT -name "*.csv" -path "*/[bcd]/*" -delete
head -40 $OUT/counters.txt | cut -c1-250
