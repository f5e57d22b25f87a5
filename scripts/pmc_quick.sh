#!/bin/bash
# Quick PMC look at the step kernel of an inference call (developer tool):  [MCPC_TUNING=rr=0] bash scripts/pmc_quick.sh <tag>
set -e -o pipefail
TAG=${1:-pmcq}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A G
G[issue]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
G[mem]="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT"
G[valu]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
for g in issue mem valu; do
  rocprofv3 --pmc ${G[$g]} --output-format csv -d $OUT/raw_$g -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-self-check ${PMC_MODE:---only-inference} --T 600 > $OUT/$g.json 2> $OUT/$g.err || { echo "pass $g failed"; tail -3 $OUT/$g.err; continue; }
  python3 $ROOT/scripts/reduce_pmc.py "$(find $OUT/raw_$g -name '*counter_collection.csv' | head -1)" $OUT/sum_$g.json
  rm -rf $OUT/raw_$g
done
python3 - <<PY
import json
for g in ("issue","mem","valu"):
    d=json.load(open("$OUT/sum_%s.json" % g))
    for k,v in d.items():
        if "steps" in k: print(g, k, {n: "%.4g" % x for n,x in v["counters"].items()})
PY
