#!/bin/bash
# The driver-format bench line (with roofline.traffic from profiles/r06_pmc_summary.json of the same csrc), the kernel stats of the same
# command, the other configurations and the GPU suite, at the tree's kernel sources:  bash scripts/final_lines.sh  ->  gpurun_out/r06/
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench_line.json')); r=d['roofline']; print(d['value'], d['config']['inference_only']['steps_per_s'], r['frac'], r['traffic'], d['build_info']['csrc'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-self-check > $OUT/stats_bench.json 2> $OUT/stats.err
cd $ROOT
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
grep steps_ws2 $OUT/kernel_stats.csv | cut -c1-120
(python3 scripts/other_configs.py; python3 scripts/train_recipe.py --batches 256 6000) 2>&1 | grep -v amdgpu > $OUT/other_configs.txt
python3 scripts/small_shard.py 2000 16 256 4096 8192 2>&1 | grep -v amdgpu > $OUT/small_shard.txt
python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
tail -1 $OUT/pytest.log
