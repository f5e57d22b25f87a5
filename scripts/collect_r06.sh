#!/bin/bash
# Copies the round-6 evidence from gpurun_out/ (scratch) into profiles/ (tracked).  Run from the repo root after scripts/profile_round.sh r06,
# scripts/pmc_round.sh r06 and the A/B calls named in profiles/README.md.
set -e
G=gpurun_out
cp $G/r06/bench_line.json profiles/r06_bench_line.json
cp $G/r06/kernel_stats.csv profiles/r06_bench_kernel_stats.csv
[ -f $G/r06/pmc_summary.json ] && cp $G/r06/pmc_summary.json profiles/r06_pmc_summary.json
( echo "# python -m pytest tests -m gpu -q (inside scripts/profile_round.sh r06; builder-run, one MI355X)"; tail -1 $G/r06/pytest.log ) > profiles/r06_gpu_tests.txt
( echo "# scripts/other_configs.py, train_recipe.py --batches 256 6000 (inside scripts/profile_round.sh r06), then short_calls.py, learn_sweep.py, small_shard.py: builder-run, final csrc of round 6"
  grep -v amdgpu $G/r06/other_configs.txt
  echo; echo "## scripts/small_shard.py 2000 16 256 4096 8192 (the reference's net 20-128-128-784: MCPC | MAP (Adam) | learning, us per step; default tuning = the unified-wave kernel)"; cat $G/r06/small_shard.txt
  echo; echo "## scripts/short_calls.py (cfg-M's net, 6000 chains, energies at the last step; us per Langevin step all-in)"; cat $G/r06/short_calls.txt
  echo; echo "## scripts/learn_sweep.py (cfg-M's net, T = 1000 inference call / learning call with 800 accumulating steps)"; cat $G/r06/learn_sweep.txt ) > profiles/r06_other_configs.txt
cp $G/r06_k1_mixed.txt profiles/r06_k1_mixed.txt
cp $G/r06_k1_acc_split.txt profiles/r06_k1_acc_split.txt
cp $G/r06_flush_serial.txt profiles/r06_flush_serial.txt
[ -f $G/r06/accuracy.txt ] && ( echo "# python -m pytest tests/test_gpu_accuracy.py -q -s : the printed figures (max error / sum|terms| against fp64; the bound asserted; torch's fp32 GEMM on the same GPU)"; cat $G/r06/accuracy.txt ) > profiles/r06_accuracy.txt
python3 scripts/parity_report.py $G/parity_errors.jsonl > profiles/r06_parity_errors.txt 2>/dev/null || true
ls -la profiles/r06_*
