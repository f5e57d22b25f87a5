#!/usr/bin/env python3
"""Developer probe: what the Hebbian flush costs when NOTHING runs beside it (tuning no_overlap=1: serial flushes on the caller's
stream, so a kernel trace of this script gives the kernels' own durations) and what the step kernel's Hebbian stretches cost alone.
    [MCPC_LIB=...] rocprofv3 --kernel-trace --stats ... -- python3 scripts/flush_alone.py [T] [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402
T = int(sys.argv[1]) if len(sys.argv) > 1 else 640
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(B, 30, dev)
for tuning in [t or None for t in os.environ.get("FLUSH_TUNINGS", "no_overlap=1,slot_cap=128;").split(";")]:
    eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=dev, tuning=tuning)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
    base = dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, lr=0.03, seed=1)
    out = []
    for name, kw in (("inference", {}), ("learning (all steps accumulate)", dict(acc_begin=0, acc_end=T))):
        best, k1 = 1e9, 0.0
        for rep in range(3):
            eng.load_state(xs)
            eng.set_profiling(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(T, **base, **kw)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / T * 1e6
            ms, n, steps = eng.last_step_kernel_ms()
            eng.set_profiling(False)
            if dt < best:
                best, k1 = dt, ms * 1e3 / T
        out.append(f"{name} {best:6.1f} us/step (step kernel's launches {k1:6.1f}, the rest {best - k1:5.1f})")
    print(os.path.basename(os.environ.get("MCPC_LIB", "libmcpc.so")), f"B={B} T={T} tuning={tuning}:", " | ".join(out), flush=True)
    eng.close()
