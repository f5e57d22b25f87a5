#!/bin/bash
# A/B of the flush's MFMA-phase operand schedule (MCPC_HEB7_PF): serial flushes, the flush = "the rest" column
mkdir -p gpurun_out/r5b
for v in base pf1 pf2 base pf1 pf2; do
  L=$PWD/scripts/bin/libmcpc_$v.so; [ $v = base ] && L=$PWD/montecarlopredictivecoding_amd/libmcpc.so
  FLUSH_TUNINGS="no_overlap=1,slot_cap=128" MCPC_LIB=$L python3 scripts/flush_alone.py 384 6000 2>&1 | tail -1
done | tee gpurun_out/r5b/heb_pf_ab.txt
for v in pf1 pf2; do
  MCPC_LIB=$PWD/scripts/bin/libmcpc_$v.so python3 -m pytest tests/test_gpu_headline.py -q -x -m gpu 2>&1 | tail -2
done | tee -a gpurun_out/r5b/heb_pf_ab.txt
