#!/bin/bash
# learning-call A/B of library builds at 6000 and 4096 chains: scripts/learn_ab2.sh <out> <lib> [<lib> ...]
OUT=$1; shift
mkdir -p $(dirname $OUT)
for rep in 1 2; do
  for lib in "$@"; do
    for n in 6000 4096; do
      MCPC_LIB=$lib python3 scripts/quick.py 600 $n >> $OUT 2>&1 || exit 1
    done
  done
done
grep -v amdgpu $OUT | sed 's/mcpc::mcpc_steps_ws2_kernel<1, \(true\|false\)>\( (round schedule[^)]*)\)\?//'
