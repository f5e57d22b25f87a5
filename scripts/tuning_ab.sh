#!/bin/bash
# A/B of tuning strings inside one gpurun call: learning AND inference-only us per step of a short bench run each ("-" = defaults).
mkdir -p gpurun_out/tab
for t in "$@"; do
  [ "$t" = "-" ] && t=""
  MCPC_TUNING="$t" timeout -k 10 300 python3 bench.py --steps ${AB_STEPS:-4} --warmup 1 --no-cpu-baseline --no-self-check > gpurun_out/tab/o.json 2> gpurun_out/tab/o.err || { echo "[$t] failed"; tail -3 gpurun_out/tab/o.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/tab/o.json"))
print("[%s] learning %.2f us/step  inference %.2f us/step  mixed-cycles %.2f" % ("$t", d["config"]["us_per_langevin_step"], d["config"]["inference_only"]["us_per_langevin_step"], d["roofline"]["mixed_schedule"]["us_per_step"]))
PY
done
