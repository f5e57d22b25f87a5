#!/usr/bin/env python3
"""Developer tool: run cfg-M on the diagnostic library (make -C montecarlopredictivecoding_amd/csrc stamps)
and print the per-phase cycle shares the kernel stamps report.  MCPC_LIB must point at libmcpc_stamps.so."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(B, 30, dev)
if os.environ.get("STAMPS_NET") == "ml":          # mcpc_ml's net 20-128-128-784 (the reference's training recipe) instead of cfg-M's
    SIZES = [20, 128, 128]
    g = torch.Generator().manual_seed(1)
    dims = [20] + SIZES + [N_OUT]
    W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(4)]
    b = [((torch.rand(dims[j + 1], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(4)]
    xs = [(torch.rand(B, n, generator=g) * 2 - 1).to(dev) for n in SIZES]
eng = Engine(SIZES, [L.ACT_RELU] * 3, SIZES[0], N_OUT, B, device=dev)
eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y); eng.load_state(xs)
learn = len(sys.argv) > 3 and sys.argv[3] == "learn"       # learning mode: Hebbian sums over all K steps (flushes overlap)
eng.run(K, loss_kind=(L.LOSS_NONE if os.environ.get("STAMPS_LOSS") == "none" else L.LOSS_BERNOULLI), lr=0.03, noise_mode=L.NOISE_PHILOX, seed=1, energy_mode=(L.ENERGY_LAST if os.environ.get("STAMPS_ENERGY") == "last" else L.ENERGY_ALL),
        **(dict(acc_begin=0, acc_end=K) if learn else {}))
torch.cuda.synchronize()
