#!/bin/bash
# learning-call A/B of library builds with the flush serial or overlapped: scripts/learn_ab.sh <out> <lib> [<lib> ...]
OUT=$1; shift
mkdir -p $(dirname $OUT)
for lib in "$@"; do
  for t in "" "no_overlap=1"; do
    for n in 4096 6000; do
      echo "tuning=$t" >> $OUT
      MCPC_LIB=$lib QUICK_TUNING=$t python3 scripts/quick.py 600 $n >> $OUT 2>&1 || exit 1
    done
  done
done
grep -v amdgpu $OUT
