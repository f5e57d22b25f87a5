#!/bin/bash
# Instruction-cache counters of the step kernels (one group per pass; see pmc_round.sh for the rules).
#   bash scripts/pmc_icache.sh <tag>  ->  gpurun_out/<tag>/sum_<mode>_<group>.json
set -e -o pipefail
TAG=${1:-icache}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A PMCG
PMCG[ic1]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
PMCG[ic2]="SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for MODE in ${MODES:-inference}; do
  if [ $MODE = learning ]; then ARGS="--no-secondary"; else ARGS="--only-inference"; fi
  for G in ic1 ic2; do
    D=$OUT/raw_${MODE}_$G
    echo "== $MODE $G: ${PMCG[$G]}"
    rocprofv3 --pmc ${PMCG[$G]} --output-format csv -d $D -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-self-check $ARGS > $OUT/${MODE}_$G.json 2> $OUT/${MODE}_$G.err || { echo "pass failed"; tail -3 $OUT/${MODE}_$G.err; continue; }
    F=$(ls $D/*counter_collection.csv | head -1)
    python3 $ROOT/scripts/reduce_pmc.py "$F" $OUT/sum_${MODE}_$G.json
  done
done
