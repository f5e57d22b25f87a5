#!/usr/bin/env python3
"""Developer measurement: what the ACCUMULATING launches of the step kernel pay beyond an inference launch, split by
same-data A/Bs in one process (cfg-M, 6000 chains, all steps of a call accumulate; HIP events around the step kernel's launches:
mcpc_last_step_kernel_ms, and the shader clock the kernel's own clock wave saw: mcpc_last_shader_clock_ghz):
  inference            the same kernel, nothing spilled
  spill, flush serial  K1 spills; every flush runs on the caller's stream BETWEEN the segments (tuning no_overlap=1): K1 alone on the chip
  spill, flush beside  the shipped schedule: the flush of segment s runs on two low-priority streams beside the launches of segment s + 1
Per row: us per Langevin step of the step kernel's launches alone (events), us per step of the whole call (wall), shader clock.
    python3 scripts/k1_acc_split.py [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(6000, 30, dev)
rows = []
for name, tuning, acc in (("inference", None, False), ("spill, flush serial (K1 alone on the chip)", "no_overlap=1,slot_cap=128", True),
                          ("spill, flush beside the next segment (shipped)", None, True)):
    eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, 6000, device=dev, tuning=tuning)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
    kw = dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, lr=0.03, seed=1)
    if acc:
        kw.update(acc_begin=0, acc_end=T)
    best = None
    for rep in range(3):
        eng.load_state(xs)
        eng.run(128, **({**kw, "acc_end": 128} if acc else kw)); torch.cuda.synchronize()          # warm-up (allocates the ring)
        eng.load_state(xs)
        eng.set_profiling(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.run(T, **kw)
        eng.sync_check(); wall = (time.perf_counter() - t0) / T * 1e6
        ms, n_launch, n_steps = eng.last_step_kernel_ms()
        ghz = eng.last_shader_clock_ghz()
        eng.set_profiling(False)
        row = (ms * 1e3 / max(n_steps, 1), wall, ghz, n_launch)
        best = row if best is None or row[0] < best[0] else best
    rows.append((name, best))
    eng.close()
print(f"cfg-M, 6000 chains, T = {T}: step kernel's launches (HIP events) | whole call (wall) | shader clock | launches")
for name, (k1, wall, ghz, n) in rows:
    print(f"  {name:52s} {k1:7.2f} us/step | {wall:7.2f} us/step | {ghz:5.3f} GHz | {n}")
inf, serial, beside = rows[0][1], rows[1][1], rows[2][1]
print(f"  -> the spill itself (stores, maxima): {serial[0] - inf[0]:+.2f} us per step of K1;  sharing the chip with the flush: {beside[0] - serial[0]:+.2f};"
      f"  clock {inf[2]:.3f} -> {serial[2]:.3f} -> {beside[2]:.3f} GHz ({(inf[2] / beside[2] - 1) * 100:+.1f} % of K1's time from the clock alone)")
print(f"  -> whole call: learning - inference = {beside[1] - inf[1]:.2f} us per step; the serial flush costs {serial[1] - serial[0]:.2f} us per step on its own")
