ROOT=$(pwd); OUT=$ROOT/gpurun_out/k3serial; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
QUICK_TUNING=no_overlap=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/scripts/quick.py 1000 6000 > $OUT/q.txt 2> $OUT/q.err
cd $ROOT
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/stats
cat $OUT/q.txt | grep -v amdgpu; head -8 $OUT/kernel_stats.csv
