#!/bin/bash
mkdir -p gpurun_out/r5b
for rep in 1 2; do
FLUSH_TUNINGS="no_overlap=1,slot_cap=128;no_overlap=1,slot_cap=128,heb_merge=1;;heb_merge=1" python3 scripts/flush_alone.py 384 6000 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r5b/heb_merge_ab.txt
MCPC_TUNING=heb_merge=1 python3 -m pytest tests/test_gpu_headline.py tests/test_gpu_fullsize.py -q -x -m gpu -k "headline or hebbian" 2>&1 | tail -2 | tee -a gpurun_out/r5b/heb_merge_ab.txt
