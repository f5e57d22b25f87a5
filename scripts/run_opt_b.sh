set -e
mkdir -p gpurun_out
( python scripts/quick.py 1000 6000
QUICK_TUNING=heb_fp32=1 python scripts/quick.py 1000 6000
python scripts/quick.py 1000 4096
QUICK_TUNING=heb_fp32=1 python scripts/quick.py 1000 4096
python scripts/quick.py 1000 16384
QUICK_TUNING=heb_fp32=1 python scripts/quick.py 1000 16384
python scripts/small_shard.py 2000 256
MCPC_LIB=$PWD/scripts/bin/libmcpc_b6.so python scripts/quick.py 1000 6000 ) > gpurun_out/opt_b.txt 2>&1
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/opt_b_tests.txt 2>&1
tail -3 gpurun_out/opt_b_tests.txt
cat gpurun_out/opt_b.txt
