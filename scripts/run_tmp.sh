python3 scripts/rr_probe.py 1000 6000 "" "no_xl=1" 2>&1 | grep -v amdgpu
python3 scripts/rr_probe.py 1000 4000 "" "no_xl=1" 2>&1 | grep -v amdgpu
python3 scripts/small_shard.py 2000 256 2>&1 | grep -v amdgpu
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tail -15
