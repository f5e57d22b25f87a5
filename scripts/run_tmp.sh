for v in base heb_nopad base heb_nopad; do
  L=$PWD/scripts/bin/libmcpc_$v.so; [ $v = base ] && L=$PWD/montecarlopredictivecoding_amd/libmcpc.so
  MCPC_LIB=$L python3 scripts/quick.py 1000 6000 2>&1 | grep -v amdgpu | sed 's/mcpc::mcpc_steps_ws2_kernel<1, true> (round schedule: k=3 launches per cycle, every 16-chain unit in m=2 of them)//'
  MCPC_LIB=$L python3 scripts/quick.py 1000 4000 2>&1 | grep -v amdgpu
done
timeout -k 10 600 python -m pytest tests -q -m gpu -k "headline or hebbian or rounds" 2>&1 | tail -3
