#!/usr/bin/env python3
"""Developer tool (round 6): calibrate the cost model the unified-wave kernel's rows are dealt by (csrc/mcpc_api.hip: build_phases_u; tuning
knobs u_row, u_gemm0, u_kb, u_kbt, u_eh, u_eb, u_ef) by measurement: us per step of MCPC / MAP / learning calls on the reference's net at
batch 256 for a grid of model parameters.    python3 scripts/u_cost_search.py [T]"""
import itertools
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda", 0)
sizes = [int(v) for v in os.environ.get("U_SIZES", "20,128,128").split(",")]
n_out, B = int(os.environ.get("U_NOUT", "784")), int(os.environ.get("U_B", "256"))
g = torch.Generator().manual_seed(1)
dims = [sizes[0]] + sizes + [n_out]
W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(4)]
b = [((torch.rand(dims[j + 1], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(4)]
y = (torch.rand(B, n_out, generator=g) < 0.13).float().to(dev)
xs = [((torch.rand(B, n, generator=g) * 2 - 1)).to(dev) for n in sizes]


def measure(tuning):
    eng = Engine(sizes, [L.ACT_RELU] * len(sizes), sizes[0], n_out, B, device=dev, tuning=tuning)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
    out = []
    for kw in (dict(noise_mode=L.NOISE_PHILOX, lr=0.03), dict(noise_mode=L.NOISE_NONE, xopt=L.XOPT_ADAM, lr=0.1),
               dict(noise_mode=L.NOISE_PHILOX, lr=0.03, acc_begin=T // 3, acc_end=T)):
        best = 1e9
        for rep in range(3):
            eng.load_state(xs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(T, loss_kind=L.LOSS_BERNOULLI, seed=1, energy_mode=L.ENERGY_LAST, **kw)
            eng.sync_check(); best = min(best, (time.perf_counter() - t0) / T * 1e6)
        out.append(best)
    eng.close()
    return out


base = dict(u_row=2000, u_gemm0=1200, u_kb=330, u_kbt=68, u_eh=1000, u_eb=1300, u_ef=500)
print("default", base, " ".join("%.2f" % v for v in measure("ws=3")), flush=True)
results = []
grid = {k: [int(x) for x in v.split("|")] for k, v in (kv.split("=") for kv in os.environ.get("U_GRID", "u_row=1000|2000|3500,u_kb=200|330|500,u_eh=600|1000|1500,u_eb=900|1300|2000").split(","))}
for vals in itertools.product(*grid.values()):
    p = dict(base, **dict(zip(grid.keys(), vals)))
    tuning = "ws=3," + ",".join(f"{k}={v}" for k, v in p.items())
    m = measure(tuning)
    results.append((sum(m), m, p))
    print(" ".join(f"{k}={v}" for k, v in p.items() if k in grid), " ".join("%.2f" % v for v in m), flush=True)
results.sort(key=lambda r: r[0])
print("best five:")
for s_, m, p in results[:5]:
    print("  ", " ".join("%.2f" % v for v in m), p)
