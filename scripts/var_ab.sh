#!/bin/bash
# A/B of library variants (scripts/bin/libmcpc_X.so, `make -C montecarlopredictivecoding_amd/csrc variant VARNAME=X VARFLAGS=...`) inside ONE
# gpurun call: small shard (B = 256, the reference's training batch) and cfg-M (B = 6000) us per step.   bash scripts/var_ab.sh base X Y base
for v in "$@"; do
  L=$PWD/scripts/bin/libmcpc_$v.so; [ $v = base ] && L=$PWD/montecarlopredictivecoding_amd/libmcpc.so
  echo "== $v"
  MCPC_LIB=$L timeout -k 10 120 python3 scripts/small_shard.py ${SS_T:-2000} ${SS_B:-256} 2>&1 | grep -v amdgpu.ids || { echo "$v failed (small)"; exit 1; }
  [ -n "$NO_QUICK" ] || MCPC_LIB=$L timeout -k 10 200 python3 scripts/quick.py ${Q_T:-1000} 6000 2>&1 | grep -v amdgpu.ids || { echo "$v failed (quick)"; exit 1; }
done
