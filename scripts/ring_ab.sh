#!/bin/bash
# Spill-ring geometry A/B inside one gpurun call: learning-call us per step for tuning strings (ring slots / parts / tail segment).
mkdir -p gpurun_out/ring
for t in "$@"; do
  [ "$t" = "-" ] && t=""
  MCPC_TUNING="$t" timeout -k 10 200 python3 bench.py --steps ${AB_STEPS:-4} --warmup 1 --no-cpu-baseline --no-self-check --no-secondary > gpurun_out/ring/o.json 2> gpurun_out/ring/o.err || { echo "[$t] failed"; tail -3 gpurun_out/ring/o.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/ring/o.json"))
print("[%s] learning %.2f us/step  plain-kernel %.2f  slots %d  brackets %d" % ("$t", d["config"]["us_per_langevin_step"], d["roofline"]["us_per_step"], d["config"]["spill_slots"], d["roofline"]["brackets"]))
PY
done
