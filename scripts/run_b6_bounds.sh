# bounds of the bf16x6 step kernel's E waves (timing builds; wrong results for the NO* variants)
for v in b6 b6_nospill b6_noeload b6_nox b6_noepi base; do
  L=$PWD/scripts/bin/libmcpc_$v.so; [ $v = base ] && L=$PWD/montecarlopredictivecoding_amd/libmcpc.so
  echo "== $v"
  MCPC_LIB=$L timeout -k 10 200 python3 scripts/quick.py 1000 6000 2>&1 | grep -v amdgpu.ids
  MCPC_LIB=$L QUICK_TUNING=no_mix=1 timeout -k 10 200 python3 scripts/quick.py 1000 6000 2>&1 | grep -v amdgpu.ids
done
