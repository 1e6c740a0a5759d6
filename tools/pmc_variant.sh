#!/bin/bash
# HBM traffic of k_bp_mfma for build variants (parallel-ddp_amd/lib/libpddp_<tag>.so): usage tools/pmc_variant.sh <out dir under gpurun_out> <tag> ...
# separate FETCH_SIZE / WRITE_SIZE passes (kernel trace only) of tools/bp_exp_times.py <tag>; prints 2 x FETCH_SIZE KiB + WRITE_SIZE KiB per launch
OUTN=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$OUTN; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for tag in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    PDDP_ROUNDS=1 timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc $c -d $OUT/pmc_${tag}_$c -o run -- python $ROOT/tools/bp_exp_times.py $tag > $OUT/pmc_${tag}_$c.log 2>&1
  done
done
python - "$OUT" "$@" <<'PY'
import csv, collections, statistics, glob, sys
out, tags = sys.argv[1], sys.argv[2:]
for tag in tags:
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = collections.defaultdict(list)
        for f in glob.glob(f"{out}/pmc_{tag}_{c}/**/run_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_bp_mfma" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    d[c].append(float(r["Counter_Value"]))
        v[c] = statistics.median(d[c]) if d[c] else float("nan")
    rd, wr = 2 * v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
    print(f"{tag}: k_bp_mfma per launch  read {rd / 1e6:.1f} MB  written {wr / 1e6:.1f} MB  total {(rd + wr) / 1e6:.1f} MB  (median over launches of both cost-to-go modes)")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "run_counter_collection.csv" -delete
