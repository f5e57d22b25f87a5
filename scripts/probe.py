#!/usr/bin/env python3
"""Developer probe: steps/s of the fused kernel at cfg-M with parts of the epilogue switched off
(ablation by configuration, not by code removal).  Usage: python scripts/probe.py [steps] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(B, 30, dev)
eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=dev)
eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
variants = {
    "full (philox, bernoulli, energies)": dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL),
    "no energies": dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_NONE),
    "no noise": dict(noise_mode=L.NOISE_NONE, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL),
    "gaussian loss": dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_GAUSSIAN, energy_mode=L.ENERGY_ALL),
    "no loss": dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_NONE, energy_mode=L.ENERGY_ALL),
    "bare (no noise, no loss, no energies)": dict(noise_mode=L.NOISE_NONE, loss_kind=L.LOSS_NONE, energy_mode=L.ENERGY_NONE),
    "adam, no noise": dict(noise_mode=L.NOISE_NONE, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, xopt=L.XOPT_ADAM),
    "full again": dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL),
    "learning (acc all steps)": dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, acc_begin=0, acc_end=K),
}
for name, kw in variants.items():
    eng.load_state(xs)
    eng.run(50, lr=0.03, seed=1, **{k: (min(v, 50) if k == "acc_end" else v) for k, v in kw.items()})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(K, lr=0.03, seed=1, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:42s} {K / dt:9.1f} steps/s   {dt / K * 1e6:8.1f} us/step", flush=True)
