#!/usr/bin/env python3
"""Developer probe: what the noise, the loss value and the energies cost the step kernel (plain schedule unless MCPC_TUNING says otherwise)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402
K, B = 1500, 6000
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(B, 30, dev)
eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=dev)
eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
cases = [("full (noise, BCE, energies every step)", dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL)),
         ("no noise", dict(noise_mode=L.NOISE_NONE, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL)),
         ("energies only at the last step", dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_LAST)),
         ("no noise, energies last", dict(noise_mode=L.NOISE_NONE, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_LAST)),
         ("no loss (free-running), noise", dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_NONE, energy_mode=L.ENERGY_ALL)),
         ("gaussian loss, noise", dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_GAUSSIAN, loss_var=0.3, energy_mode=L.ENERGY_ALL))]
for name, kw in cases:
    best = 1e9
    for rep in range(3):
        eng.load_state(xs)
        eng.run(50, lr=0.03, seed=1, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run(K, lr=0.03, seed=1, **kw)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K * 1e6)
    print(f"{name:45s} {best:6.1f} us/step", flush=True)
