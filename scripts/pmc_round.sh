#!/bin/bash
# PMC evidence for the step kernels and the Hebbian GEMMs (run on an MI355X box via gpurun, from the repo root):
#   bash scripts/pmc_round.sh <tag>     ->  gpurun_out/<tag>/pmc_summary.json  (+ the per-pass raw sums)
# One counter group per rocprofv3 pass (SQ: 8 slots; FETCH_SIZE and WRITE_SIZE cannot share a pass), the program directly
# after `--`, nothing but --pmc on the command line (MI355X_MICROARCH.md, rocprofv3 PMC slots; gpurun rules).
# Counter collection serialises kernels: per-kernel COUNTS and RATIOS are what these passes are for, not wall time.
set -e -o pipefail
TAG=${1:-pmc}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A PMCG
PMCG[sq_issue]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
PMCG[sq_mem]="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT"
PMCG[sq_fifo]="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_CYCLES"
PMCG[sq_valu]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"
PMCG[sq_mfma]="SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F16"
PMCG[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
PMCG[tcp2]="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum"
# (a TA_* group aborted rocprofv3 on this pool in round 2 -- SIGABRT after minutes of silence -- and is left out)
PMCG[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
PMCG[fetch]="FETCH_SIZE"
PMCG[write]="WRITE_SIZE"
PMCG[grbm]="GRBM_GUI_ACTIVE GRBM_COUNT"
for MODE in ${PMC_MODES:-learning inference}; do
  if [ $MODE = learning ]; then ARGS="--no-secondary"; else ARGS="--only-inference"; fi
  for G in sq_issue sq_mem sq_fifo sq_valu sq_mfma tcp tcp2 tcc fetch write grbm; do
    D=$OUT/raw_${MODE}_$G
    echo "== $MODE $G: ${PMCG[$G]}"
    rocprofv3 --pmc ${PMCG[$G]} --output-format csv -d $D -o pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-self-check $ARGS > $OUT/${MODE}_$G.json 2> $OUT/${MODE}_$G.err || { echo "pass failed"; tail -3 $OUT/${MODE}_$G.err; continue; }
    F=$(find $D -name "*counter_collection.csv" | head -1)
    python3 $ROOT/scripts/reduce_pmc.py "$F" $OUT/sum_${MODE}_$G.json
    rm -rf $D
  done
done
cd $ROOT
[ -n "$PMC_NO_MERGE" ] || python3 scripts/reduce_pmc.py --merge $OUT $OUT/pmc_summary.json
