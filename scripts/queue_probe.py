#!/usr/bin/env python3
"""Does the mixed schedule keep its two launches concurrent in a process that also holds an RCCL communicator?
    queue_probe.py [rccl_first|engine_first|no_rccl] [collective]     ->  us per step inside the mixed cycles"""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "rccl_first"
coll = len(sys.argv) > 2
DEV = "cuda:0"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)


def group():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV))
    if coll:
        t = torch.ones(4, device=DEV); dist.all_reduce(t); torch.cuda.synchronize()


def engine():
    W, b, y, xs = make_problem(6000, 30, torch.device(DEV))
    eng = Engine([30, 256, 256], [L.ACT_RELU] * 3, 30, 784, 6000, device=DEV)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
    return eng, xs


if mode == "rccl_first":
    group(); eng, xs = engine()
elif mode == "engine_first":
    eng, xs = engine(); group()
else:
    eng, xs = engine()
for _ in range(2):
    eng.load_state(xs)
    eng.set_profiling(True)
    eng.run(1200, loss_kind=L.LOSS_BERNOULLI, lr=0.03, noise_mode=L.NOISE_PHILOX, seed=1, step_base=0, energy_mode=L.ENERGY_LAST)
    eng.sync_check()
    ms, n_cycles, n_steps = eng.last_mixed_cycles_ms()
    eng.set_profiling(False)
print(f"{mode:13s} collective={int(coll)} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '-')}: {ms / n_steps * 1e3:6.1f} us per step inside the cycles", flush=True)
eng.close()
if mode != "no_rccl":
    dist.destroy_process_group()
