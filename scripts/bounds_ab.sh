#!/bin/bash
export MCPC_ALLOW_EXP=1
for v in base noepi nogemm noload nosplit nosplitnoload; do L=$PWD/scripts/bin/libmcpc_$v.so; [ $v = base ] && L=$PWD/montecarlopredictivecoding_amd/libmcpc.so
  echo -n "$v  "; MCPC_LIB=$L python scripts/quick.py 600 4096 2>&1 | tail -1 | sed 's/.*inference/inference/'
done
MCPC_LIB=$PWD/montecarlopredictivecoding_amd/libmcpc_stamps.so python3 scripts/stamps.py 200 4096 2>&1 | grep -v amdgpu | head -9
