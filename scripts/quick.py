#!/usr/bin/env python3
"""Developer probe: inference-only and learning steps/s of cfg-M for the library MCPC_LIB points at (default: libmcpc.so)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(B, 30, dev)
eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=dev, tuning=os.environ.get("QUICK_TUNING"))
eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
base = dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, lr=0.03, seed=1)
if os.environ.get("QUICK_NONOISE"): base.update(noise_mode=L.NOISE_NONE)            # what the Philox kick costs
if os.environ.get("QUICK_ELAST"): base.update(energy_mode=L.ENERGY_LAST)            # what the per-step energies / loss cost
out = []
if os.environ.get("QUICK_PROFILE"): eng.set_profiling(True)
if os.environ.get("QUICK_REC"): base.update(rec_begin=0, rec_stride=100, rec_count=(K + 99) // 100, rec_x=True)
for name, kw in (("inference", {}), ("learning", dict(acc_begin=K // 5, acc_end=K))):
    best = 1e9
    for rep in range(3):
        eng.load_state(xs)
        eng.run(50, **base)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run(K, **base, **kw)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K * 1e6)
    out.append(f"{name} {best:6.1f} us/step")
    if os.environ.get("QUICK_MIXING"):       # the bench's call shape: K/5 mixing + 4K/5 sampling, every repetition reported
        ts = []
        for rep in range(4):
            eng.load_state(xs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(K, **base, **(dict(acc_begin=K // 5, acc_end=K) if name == "learning" else {}))
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6)
        out.append("reps " + " ".join(f"{v:.1f}" for v in ts))
print(os.path.basename(os.environ.get("MCPC_LIB", "libmcpc.so")), f"B={B}", eng.query()["step_kernel"], " | ".join(out), flush=True)
