#!/usr/bin/env python3
"""Short calls at cfg-M's size: ms per call and us per step for T in a few lengths, Adam (MAP) / SGD + noise, on the default plan
(16-chain workgroups, round schedule) and on the round-2 plan (`rr=0`: 32-chain workgroups + mixed schedule).
Developer measurement: what a call costs outside its steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

DEV = "cuda:0"
B = int(os.environ.get("SHORT_B", "6000"))
SIZES = [int(v) for v in os.environ.get("SHORT_SIZES", "30,256,256").split(",")]      # e.g. 20,128,128: the mcpc_ml net
if SIZES == [30, 256, 256]:
    W, b, y, xs = make_problem(B, 30, DEV)
else:
    g = torch.Generator().manual_seed(30)
    dims = [SIZES[0]] + SIZES + [784]
    W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) / dims[j] ** 0.5).to(DEV) for j in range(len(dims) - 1)]
    b = [((torch.rand(dims[j + 1], generator=g) * 2 - 1) / dims[j] ** 0.5).to(DEV) for j in range(len(dims) - 1)]
    y = (torch.rand(B, 784, generator=g) < 0.13).float().to(DEV)
    xs = [((torch.rand(B, n, generator=g) * 2 - 1) * 10).to(DEV) for n in SIZES]
xs = [x * 0.1 for x in xs]
for tuning in (None, "rr=0"):
    eng = Engine(SIZES, [L.ACT_RELU] * len(SIZES), SIZES[0], 784, B, device=DEV, tuning=tuning)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
    for name, kw in (("adam", dict(noise_mode=L.NOISE_NONE, xopt=L.XOPT_ADAM, lr=0.1)),
                     ("sgd+noise", dict(noise_mode=L.NOISE_PHILOX, lr=0.03, seed=3, step_base=0))):
        for T in [int(v) for v in os.environ.get("SHORT_T", "50,150,250,400,1000").split(",")]:
            def call():
                eng.load_state(xs)
                eng.run(T, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_LAST, **kw)
                eng.sync_check()
            call(); call()
            t0 = time.perf_counter()
            for _ in range(5):
                call()
            dt = (time.perf_counter() - t0) / 5
            print(f"{'rounds' if tuning is None else 'rr=0':6s} {name:9s} T={T:5d}: {dt*1e3:7.2f} ms per call, {dt/T*1e6:6.1f} us per step", flush=True)
    eng.close()
