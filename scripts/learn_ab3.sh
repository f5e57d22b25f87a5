#!/bin/bash
# learning-call A/B at 6000 chains, N alternating repetitions: scripts/learn_ab3.sh <out> <N> <lib> [<lib> ...]
OUT=$1; N=$2; shift; shift
mkdir -p $(dirname $OUT)
for rep in $(seq $N); do
  for lib in "$@"; do
    MCPC_LIB=$lib python3 scripts/quick.py 1000 6000 >> $OUT 2>&1 || exit 1
  done
done
grep -v amdgpu $OUT | sed 's/mcpc::mcpc_steps_ws2_kernel<1, \(true\|false\)>\( (round schedule[^)]*)\)\?//'
