#!/usr/bin/env python3
"""Developer probe (round 6): the unified-wave kernel (tuning ws=3, csrc/mcpc_steps_u.h) against the in-place kernel (ws=2) on one net:
bitwise comparison of the final states, max energy difference, and us per step of MCPC / MAP / learning / zero-loss calls.
    python3 scripts/u_probe.py [T] [B ...]      (env U_SIZES=20,128,128 U_NOUT=784 U_ACT=relu)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
batches = [int(v) for v in sys.argv[2:]] or [256]
dev = torch.device("cuda", 0)
sizes = [int(v) for v in os.environ.get("U_SIZES", "20,128,128").split(",")]
n_out = int(os.environ.get("U_NOUT", "784"))
act = {"relu": L.ACT_RELU, "tanh": L.ACT_TANH}[os.environ.get("U_ACT", "relu")]
g = torch.Generator().manual_seed(1)
dims = [sizes[0]] + sizes + [n_out]
W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(len(dims) - 1)]
b = [((torch.rand(dims[j + 1], generator=g) * 2 - 1) / dims[j] ** 0.5).to(dev) for j in range(len(dims) - 1)]
MODES = (("mcpc", dict(loss_kind=L.LOSS_BERNOULLI, noise_mode=L.NOISE_PHILOX, lr=0.03)),
         ("map-adam", dict(loss_kind=L.LOSS_BERNOULLI, noise_mode=L.NOISE_NONE, xopt=L.XOPT_ADAM, lr=0.1)),
         ("learn", dict(loss_kind=L.LOSS_BERNOULLI, noise_mode=L.NOISE_PHILOX, lr=0.03, acc_begin=T // 3, acc_end=T)),
         ("gen-zero", dict(loss_kind=L.LOSS_NONE, noise_mode=L.NOISE_PHILOX, lr=0.1)))
for B in batches:
    y = (torch.rand(B, n_out, generator=g) < 0.13).float().to(dev)
    xs = [((torch.rand(B, n, generator=g) * 2 - 1)).to(dev) for n in sizes]
    res = {}
    for tuning in ("ws=2", "ws=3"):
        eng = Engine(sizes, [act] * len(sizes), sizes[0], n_out, B, device=dev, tuning=tuning)
        eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
        for name, kw in MODES:
            best = 1e9
            for rep in range(3):
                eng.load_state(xs)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = eng.run(T, seed=1, energy_mode=L.ENERGY_LAST, **kw)
                eng.sync_check(); best = min(best, (time.perf_counter() - t0) / T * 1e6)
            out = [torch.empty_like(x) for x in xs]
            eng.store_state(out)
            grads = eng.read_param_grads_flat().cpu() if name == "learn" else None
            res[(tuning, name)] = (best, [o.cpu() for o in out], r.energies.cpu(), grads)
        q = eng.query()
        eng.close()
    line = [f"B={B} T={T} {sizes}-{n_out} lds(ws2)={q['lds_bytes']}"]
    for name, _ in MODES:
        a, c = res[("ws=2", name)], res[("ws=3", name)]
        same = all(torch.equal(x, z) for x, z in zip(a[1], c[1]))
        dmax = max(float((x - z).abs().max()) for x, z in zip(a[1], c[1]))
        en = float(((a[2] - c[2]).abs() / a[2].abs().clamp_min(1e-30)).max())
        extra = ""
        if a[3] is not None:
            extra = " dG rel %.1e" % float((a[3] - c[3]).abs().max() / a[3].abs().max())
        line.append(f"{name}: ws2 {a[0]:6.2f} us  U {c[0]:6.2f} us  states {'BITWISE' if same else 'max|d| %.2e' % dmax}  energies rel {en:.1e}{extra}")
    print("\n   ".join(line), flush=True)
