set -e
mkdir -p gpurun_out/fd3
run() { tag=$1; shift; "$@" > gpurun_out/fd3/$tag.out 2> gpurun_out/fd3/$tag.err || { echo "$tag failed"; tail -3 gpurun_out/fd3/$tag.err; }; python3 - <<PY
import json
for ln in open("gpurun_out/fd3/$tag.out"):
    if ln.startswith("{"):
        d=json.loads(ln); m=d["roofline"].get("mixed_schedule") or {}
        print("$tag", "learning us/step", round(d["config"]["us_per_langevin_step"],1), "inference us/step", round(d["config"]["inference_only"]["us_per_langevin_step"],1), "mixed us/step", round(m.get("us_per_step",0),1), "plain us/step", round(d["roofline"]["us_per_step"],1))
PY
}
A="--steps 3 --warmup 1 --T 1000 --no-cpu-baseline"
run base python3 bench.py $A
run dist python3 bench.py $A --force-dist
GPU_MAX_HW_QUEUES=8 run dist_q8 python3 bench.py $A --force-dist
GPU_MAX_HW_QUEUES=8 run base_q8 python3 bench.py $A
