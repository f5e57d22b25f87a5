#!/usr/bin/env python3
"""Developer probe: can 16-chain workgroups on the CUs the 32-chain kernel leaves idle run beside it at full speed?
Two engines on two streams: 120 pairs (3840 chains, 32-chain workgroups) and 136 single tiles (2176 chains, 16-chain
workgroups) = 256 workgroups, one per CU.  Times each alone and both together for the same number of steps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

dev = torch.device("cuda", 0)
base = dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, lr=0.03, seed=1)

def make(B, ct):
    os.environ["MCPC_TUNING"] = f"ws=2,ct={ct}"
    W, b, y, xs = make_problem(B, 30, dev)
    eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=dev)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y); eng.load_state(xs)
    return eng

e2, e1 = make(3840, 32), make(2176, 16)
print(e2.query(), e1.query())
s2, s1 = torch.cuda.Stream(), torch.cuda.Stream()
K2, K1 = 1000, 1700

def run(do2, do1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if do2:
        with torch.cuda.stream(s2): e2.run(K2, **base)
    if do1:
        with torch.cuda.stream(s1): e1.run(K1, **base)
    torch.cuda.synchronize()
    return time.perf_counter() - t0

run(True, True)
for name, a, b in (("pairs alone", True, False), ("singles alone", False, True), ("both", True, True), ("both", True, True)):
    dt = run(a, b)
    print(f"{name:14s} {dt*1e3:8.2f} ms   ({dt/K2*1e6:.1f} us per pair-step, {dt/K1*1e6:.1f} us per single-step)")
