#!/bin/bash
# A/B of library builds on the small-shard probe and the cfg-M probe inside one gpurun call: scripts/small_ab.sh <out> <lib> [<lib> ...]
OUT=$1; shift
mkdir -p $(dirname $OUT)
for rep in 1 2; do
  for lib in "$@"; do
    MCPC_LIB=$lib python3 scripts/small_shard.py 2000 256 >> $OUT 2>&1 || exit 1
    MCPC_LIB=$lib python3 scripts/quick.py 600 6000 >> $OUT 2>&1 || exit 1
  done
done
grep -v amdgpu $OUT
