#!/bin/bash
# A/B of two builds of libpddp on ONE box (box-to-box spread of the pool is ~5 %): alternates tools/kernel_times.py between parallel-ddp_amd/lib/libpddp_A.so and _B.so
# usage (through gpurun): tools/ab_lib.sh [batch] [rounds]
B=${1:-16384}; R=${2:-3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for r in $(seq $R); do for v in A B; do echo -n "$v: "; PDDP_LIB=$ROOT/parallel-ddp_amd/lib/libpddp_$v.so python $ROOT/tools/kernel_times.py $B 2>&1 | tail -1; done; done
