#!/bin/bash
# like ab_lib.sh for few problems in flight: tools/kernel_times.py 1 64 with libpddp_A.so / libpddp_B.so alternating
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in A B; do echo "$v:"; PDDP_LIB=$ROOT/parallel-ddp_amd/lib/libpddp_$v.so python $ROOT/tools/kernel_times.py ${@:-1 64} 2>&1 | tail -2; done; done
