#!/bin/bash
# VALU / MFMA instruction totals of the step kernels for the current library, once per MCPC_TUNING value given (run via gpurun)
set -e -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-pmc_valu2}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in "" "no_lean=1"; do
  N=$( [ -z "$T" ] && echo lean || echo nolean )
  MCPC_TUNING="no_mix=1,$T" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/raw_$N -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-self-check --only-inference --T 600 > $OUT/$N.json 2> $OUT/$N.err
  python3 $ROOT/scripts/reduce_pmc.py "$(find $OUT/raw_$N -name '*counter_collection.csv' | head -1)" $OUT/sum_inference_$N.json
  rm -rf $OUT/raw_$N
  python3 - <<PY
import json
d=json.load(open("$OUT/sum_inference_$N.json"))
for k,v in d.items():
    if "steps" in k:
        c=v["counters"]; m=c["SQ_INSTS_MFMA"]
        print("$N", k, "disp", v["dispatches"], "VALU/MFMA %.3f"%((c["SQ_INSTS_VALU"]-m)/m), "SALU/MFMA %.3f"%(c["SQ_INSTS_SALU"]/m), "LDS/MFMA %.3f"%(c["SQ_INSTS_LDS"]/m), "VMEMRD/MFMA %.3f"%(c["SQ_INSTS_VMEM_RD"]/m), "activeVALU/wavecycles %.3f"%(c["SQ_ACTIVE_INST_VALU"]/c["SQ_WAVE_CYCLES"]))
PY
done
