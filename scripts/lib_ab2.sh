#!/bin/bash
# A/B of two builds of the library inside one gpurun call: scripts/tmp/sets_ab.sh <out> <libA> <libB>
OUT=$1; A=$2; B=$3
mkdir -p $(dirname $OUT)
for rep in 1 2; do
  for lib in $A $B; do
    for n in 4096 6000 256; do
      MCPC_LIB=$lib python3 scripts/quick.py 600 $n >> $OUT 2>&1 || exit 1
    done
  done
done
cat $OUT | grep -v amdgpu
