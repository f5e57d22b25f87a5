#!/bin/bash
# Instruction-cache / issue counters of the step kernel on the SMALL net (scripts/small_shard.py, 256 chains), one group per pass.
#   bash scripts/pmc_u.sh <tag> [tuning]  ->  gpurun_out/<tag>/sum_<group>.json
set -e -o pipefail
TAG=${1:-pmc_u}
export SMALL_TUNING=${2:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A PMCG
PMCG[ic1]="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
PMCG[ic2]="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
PMCG[ic3]="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F16"
PMCG[ic4]="SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
for G in ic1 ic2 ic3 ic4; do
  D=$OUT/raw_$G
  echo "== $G: ${PMCG[$G]}"
  rocprofv3 --pmc ${PMCG[$G]} --output-format csv -d $D -o pmc -- python3 $ROOT/scripts/small_shard.py 500 256 > $OUT/$G.txt 2> $OUT/$G.err || { echo "pass failed"; tail -3 $OUT/$G.err; continue; }
  F=$(ls $D/*counter_collection.csv | head -1)
  python3 $ROOT/scripts/reduce_pmc.py "$F" $OUT/sum_$G.json
done
