#!/usr/bin/env python3
"""Learning call (T = 1000: 200 mixing + 800 accumulating steps, grads read out) across shard sizes: us per step and
chain-steps per second.  Developer measurement: does every shard size get a sensible schedule?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

DEV = "cuda:0"
T = 1000
for B in [int(v) for v in os.environ.get("SWEEP_B", "1024,2048,4096,4200,5000,6000,7000,8192,12000").split(",")]:
    W, b, y, xs = make_problem(B, 30, DEV)
    for tuning in [None] + ([t for t in os.environ.get("SWEEP_TUNING", "").split(";") if t]):
        eng = Engine([30, 256, 256], [L.ACT_RELU] * 3, 30, 784, B, device=DEV, tuning=tuning)
        eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)

        def call(acc):
            eng.load_state(xs)
            eng.run(T, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_LAST, noise_mode=L.NOISE_PHILOX, lr=0.03, seed=3, step_base=0,
                    acc_begin=T // 5 if acc else 0, acc_end=T if acc else 0)
            if acc:
                eng.read_param_grads_flat(scale=1.0 / (0.8 * T * B))
            eng.sync_check()
        res = []
        for acc in (False, True):
            call(acc)
            t0 = time.perf_counter()
            for _ in range(3):
                call(acc)
            res.append((time.perf_counter() - t0) / 3)
        q = eng.query()
        print(f"B={B:6d} {str(tuning):12s} {q['chains_per_wg']:2d} chains x {q['n_workgroups']:3d} wg: inference {res[0] / T * 1e6:6.1f} us/step ({B * T / res[0] / 1e6:6.1f} M chain-steps/s)"
              f"   learning {res[1] / T * 1e6:6.1f} us/step ({B * T / res[1] / 1e6:6.1f} M chain-steps/s)", flush=True)
        eng.close()
