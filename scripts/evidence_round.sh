#!/bin/bash
# The whole round-end evidence in ONE gpurun call (from the repo root):  bash scripts/evidence_round.sh <tag>  ->  gpurun_out/<tag>/
# = profile_round.sh (GPU suite, the driver's bench line, rocprofv3 kernel stats, other configurations) + pmc_round.sh (PMC passes)
#   + in-kernel stamps (needs `make -C montecarlopredictivecoding_amd/csrc stamps`) + short calls + the shard-size sweep.
set -e -o pipefail
TAG=${1:-evidence}
OUT=gpurun_out/$TAG
bash scripts/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || { tail -20 gpurun_out/${TAG}_profile.log; exit 1; }
echo "profile done"
bash scripts/pmc_round.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1 || { tail -20 gpurun_out/${TAG}_pmc.log; exit 1; }
echo "pmc done"
if [ -f montecarlopredictivecoding_amd/libmcpc_stamps.so ]; then
  ( MCPC_LIB=$PWD/montecarlopredictivecoding_amd/libmcpc_stamps.so python3 scripts/stamps.py 200 4000
    MCPC_LIB=$PWD/montecarlopredictivecoding_amd/libmcpc_stamps.so python3 scripts/stamps.py 64 4000 learn ) 2>&1 | grep -v amdgpu > $OUT/stamps_16chain.txt || true
fi
python3 scripts/short_calls.py 2>&1 | grep -v amdgpu > $OUT/short_calls.txt || true
echo "short calls done"
SWEEP_B=${SWEEP_B:-1024,2048,4096,4200,5000,6000,7000,8192,12000,16384,24000} python3 scripts/learn_sweep.py 2>&1 | grep -v amdgpu > $OUT/learn_sweep.txt || true
tail -3 $OUT/short_calls.txt; tail -12 $OUT/learn_sweep.txt
