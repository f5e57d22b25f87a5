#!/usr/bin/env python3
"""Developer probe: us per step of a SMALL shard (the reference's training batch, table_1.py:38-43) on the library MCPC_LIB
points at: mcpc_ml's net 20-128-128-784 ReLU, Bernoulli read-out; MCPC (SGD + Philox kick) and MAP (Adam on x) calls of T steps,
energies at the last step only -- the two calls of one training iteration of the recipe (scripts/train_recipe.py).
    python3 scripts/small_shard.py [T] [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
batches = [int(v) for v in sys.argv[2:]] or [256]
dev = torch.device("cuda", 0)
sizes, n_out = [20, 128, 128], 784
g = torch.Generator().manual_seed(1)
dims = [20] + sizes + [n_out]
W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(4)]
b = [((torch.rand(dims[j + 1], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(4)]
for B in batches:
    y = (torch.rand(B, n_out, generator=g) < 0.13).float().to(dev)
    xs = [((torch.rand(B, n, generator=g) * 2 - 1)).to(dev) for n in sizes]
    eng = Engine(sizes, [L.ACT_RELU] * 3, 20, n_out, B, device=dev, tuning=os.environ.get("SMALL_TUNING"))
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
    out = []
    for name, kw in (("mcpc", dict(noise_mode=L.NOISE_PHILOX, lr=0.03)), ("map-adam", dict(noise_mode=L.NOISE_NONE, xopt=L.XOPT_ADAM, lr=0.1)),
                     ("mcpc-learn", dict(noise_mode=L.NOISE_PHILOX, lr=0.03, acc_begin=T // 3, acc_end=T))):
        best = 1e9
        for rep in range(4):
            eng.load_state(xs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(T, loss_kind=L.LOSS_BERNOULLI, seed=1, energy_mode=L.ENERGY_LAST, **kw)
            eng.sync_check(); best = min(best, (time.perf_counter() - t0) / T * 1e6)
        out.append(f"{name} {best:6.2f}")
    q = eng.query()
    print(os.path.basename(os.environ.get("MCPC_LIB", "libmcpc.so")), f"B={B} T={T} ct={q['chains_per_wg']} wgs={q['n_workgroups']}  us/step:", " | ".join(out), flush=True)
    eng.close()
