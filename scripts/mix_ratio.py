#!/usr/bin/env python3
"""Sensitivity of the mixed schedule to the assumed rate ratio of its two workgroup forms (tuning key mix_ratio, tenths):
us per step of a 2400-step inference stretch, cfg-M's net and the mcpc_ml net, 6000 chains.  Developer measurement."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

DEV = "cuda:0"
B, T = int(os.environ.get("MIX_B", "6000")), 2400
for sizes in ([30, 256, 256], [20, 128, 128]):
    g = torch.Generator().manual_seed(30)
    dims = [sizes[0]] + sizes + [784]
    W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) / dims[j] ** 0.5).to(DEV) for j in range(4)]
    b = [((torch.rand(dims[j + 1], generator=g) * 2 - 1) / dims[j] ** 0.5).to(DEV) for j in range(4)]
    y = (torch.rand(B, 784, generator=g) < 0.13).float().to(DEV)
    xs = [((torch.rand(B, n, generator=g) * 2 - 1)).to(DEV) for n in sizes]
    for tuning in ("no_mix=1", "mix_ratio=14", "mix_ratio=15", "mix_ratio=16", "mix_ratio=17", "mix_ratio=18", "mix_ratio=19", "mix_ratio=20"):
        eng = Engine(sizes, [L.ACT_RELU] * 3, sizes[0], 784, B, device=DEV, tuning=tuning)
        eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)

        def call():
            eng.load_state(xs)
            eng.run(T, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_LAST, noise_mode=L.NOISE_PHILOX, lr=0.03, seed=3, step_base=0)
            eng.sync_check()
        call()
        t0 = time.perf_counter()
        for _ in range(3):
            call()
        dt = (time.perf_counter() - t0) / 3
        print(f"{sizes} {tuning:14s}: {dt / T * 1e6:6.1f} us per step", flush=True)
        eng.close()
