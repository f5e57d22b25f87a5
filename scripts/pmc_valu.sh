#!/bin/bash
# One more PMC pass for the step kernels: what KIND of VALU instructions share the SIMD with the MFMAs (run via gpurun).
set -e -o pipefail
TAG=${1:-pmc_valu}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in "valu_kinds:SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_VALU_MFMA_COEXEC_CYCLES" \
         "valu_tot:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FLOPS_FP32"; do
  N=${G%%:*}; C=${G#*:}
  echo "== $N: $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/raw_$N -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-self-check --only-inference --T 1200 > $OUT/$N.json 2> $OUT/$N.err || { echo failed; tail -3 $OUT/$N.err; continue; }
  python3 $ROOT/scripts/reduce_pmc.py "$(find $OUT/raw_$N -name '*counter_collection.csv' | head -1)" $OUT/sum_inference_$N.json
  rm -rf $OUT/raw_$N
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/sum_inference_*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        if "steps" in k:
            print(k, v["dispatches"], {c: "%.4g"%x for c,x in v["counters"].items()})
PY
