#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box variance is several %):  bash scripts/lib_ab.sh <tag> base h2 h16 base ...
# (names: base = the in-tree libmcpc.so, X = scripts/bin/libmcpc_X.so); prints learning / inference us per step of a short bench run.
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  L=$PWD/scripts/bin/libmcpc_$v.so; [ $v = base ] && L=$PWD/montecarlopredictivecoding_amd/libmcpc.so
  MCPC_LIB=$L timeout -k 10 200 python3 bench.py --steps ${AB_STEPS:-4} --warmup 1 --no-cpu-baseline --no-self-check > gpurun_out/$TAG/$v.json 2> gpurun_out/$TAG/$v.err || { echo "$v failed"; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/$TAG/$v.json"))
print("%-8s learning %.2f us/step  plain-kernel %.2f  inference %.2f" % ("$v", d["config"]["us_per_langevin_step"], d["roofline"]["us_per_step"], d["config"]["inference_only"]["us_per_langevin_step"]))
PY
done
