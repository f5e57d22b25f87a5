#!/usr/bin/env python3
"""Developer tool: call the wide-network fuzz test many times in one process, device memory polluted in between; prints the failures."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fails = 0
for i in range(n):
    junk = [torch.full((64 << 20,), float("nan"), device="cuda:0") for _ in range(4)] + [torch.full((64 << 20,), 3e38, device="cuda:0")]
    torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    try:
        F.test_wide_networks_against_oracle([40, 384, 200], 784)
    except Exception as e:          # noqa: BLE001
        fails += 1
        print(f"iteration {i}: FAILED\n{str(e)[:2500]}", flush=True)
        if fails >= 2: break
print("done", n, "iterations,", fails, "failures")
