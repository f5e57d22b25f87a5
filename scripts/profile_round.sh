#!/bin/bash
# Round-end evidence run on an MI355X box (via gpurun, from the repo root):   bash scripts/profile_round.sh <tag>  ->  gpurun_out/<tag>/
#   pytest.log          the GPU suite
#   bench_line.json     the driver's command (bench.py --steps 20 --warmup 5)
#   kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same program at --steps 3 --warmup 1 --no-secondary (learning calls only: the
#                       step kernel has ONE name in every stretch and mode, and roofline.avg_bracket_ms of the line is the average over
#                       the launches of the timed LEARNING calls -- inference-only calls use longer launches)
#   other_configs.txt   the other BASELINE.json configurations through the facade
# The PMC passes are scripts/pmc_round.sh (one counter group per pass; they serialise kernels, so they are a separate call).
set -e -o pipefail
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1 || { tail -20 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err
cat $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-self-check > $OUT/stats_bench.json 2> $OUT/stats.err
cd $ROOT
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 scripts/other_configs.py > $OUT/other_configs.txt 2>&1 || true
python3 scripts/train_recipe.py --batches 256 6000 >> $OUT/other_configs.txt 2>&1 || true
grep -v amdgpu $OUT/other_configs.txt
