#!/bin/bash
# Round-end evidence run on an MI355X box (via gpurun): default bench line, rocprofv3 kernel stats of the same command,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE) that scripts/collect_traffic.py reduces to HBM bytes per step.
#   usage (from the repo root on the GPU box):  bash scripts/profile_round.sh <tag>       -> gpurun_out/<tag>/
set -e -o pipefail
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err
cat $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp
# the timed call only (168 step-kernel launches), so that rocprofv3's average launch duration is directly comparable with
# roofline.avg_launch_ms of the bench line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 5000 --warmup 0 --no-secondary --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 5000 --warmup 0 --no-secondary --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 5000 --warmup 0 --no-secondary --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
cd $ROOT
find $OUT -name "*.csv" | sort
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 scripts/collect_traffic.py $F $W 5000 $OUT/hbm_traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# the raw PMC traces are large: keep only the reduced json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
python3 scripts/other_configs.py > $OUT/other_configs.txt 2>&1 || true
cat $OUT/other_configs.txt
