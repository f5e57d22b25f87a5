python3 scripts/quick.py 1000 6000 2>&1 | grep -v amdgpu | sed 's/mcpc::mcpc_steps_ws2_kernel<1, true> (round schedule: k=3 launches per cycle, every 16-chain unit in m=2 of them)//'
python3 scripts/quick.py 1000 4000 2>&1 | grep -v amdgpu
python3 scripts/small_shard.py 2000 256 2>&1 | grep -v amdgpu
python3 scripts/quick.py 1000 6000 2>&1 | grep -v amdgpu | sed 's/mcpc::mcpc_steps_ws2_kernel<1, true> (round schedule: k=3 launches per cycle, every 16-chain unit in m=2 of them)//'
timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
