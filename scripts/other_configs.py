#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations through the facade (developer measurement for DESIGN.md)."""
import os
import sys
import time
import warnings

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlopredictivecoding_amd.predictive_coding as pc  # noqa: E402
from montecarlopredictivecoding_amd.utils import model as um  # noqa: E402
from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer, get_pc_trainer  # noqa: E402

warnings.simplefilter("ignore")
dev = "cuda:0"


def timed(fn, T):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return T / dt, dt


# cfg-1: linear-Gaussian toy (figure_2.py:40-64), 256 identical chains, 1000 steps, everything recorded
m = nn.Sequential(nn.Linear(1, 1), pc.PCLayer(sample_x_fn=um.sample_x_fn_cte), nn.Linear(1, 1, bias=False)).train()
nn.init.constant_(m[0].bias, 0.2); nn.init.constant_(m[2].weight, 2.0); m.to(dev)
cfg = {"mixing": 0, "sampling": 1000, "optimizer_x_kwargs_mcpc": {"lr": 0.02}}
tr = get_mcpc_trainer(m, cfg, training=False)
z, y = torch.zeros(256, 1, device=dev), torch.ones(256, 1, device=dev)
r, dt = timed(lambda: tr.train_on_batch(inputs=z, loss_fn=um.fe_fn, loss_fn_kwargs={"_target": y, "_var": 1.0},
                                        callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": tr},
                                        is_log_progress=False, is_return_results_every_t=True, is_return_representations=True), 1000)
print(f"cfg-1  linear-Gaussian B=256 T=1000 (all steps recorded to host): {r:10.0f} steps/s  ({dt*1e3:.1f} ms per call)")

# figure_3b: single chain, 20-128-128-784, zero loss, outputs every step
c = dict(input_size=20, hidden_size=128, hidden2_size=128, output_size=784, activation_fn="relu", mixing=1000, sampling=9000,
         optimizer_x_kwargs_mcpc={"lr": 0.1})
m = um.get_model(c, True)
tr = get_mcpc_trainer(m, c, training=False)
for B, rec in ((1, True), (8192, False)):
    z = torch.zeros(B, 20, device=dev)
    r, dt = timed(lambda: tr.train_on_batch(inputs=z, loss_fn=um.zero_fn, loss_fn_kwargs={}, callback_after_t=um.random_step,
                                            callback_after_t_kwargs={"_pc_trainer": tr}, is_log_progress=False,
                                            is_return_results_every_t=rec, is_return_outputs=rec,
                                            is_checking_after_callback_after_t=False), 10000)
    print(f"cfg-gen 20-128-128-784 zero loss, {B:5d} chains, T=10000, outputs {'every step' if rec else 'last only'}: {r:10.0f} steps/s  ({dt:.3f} s per call)")

# cfg-PC: MAP inference, Adam-x lr 0.1, T=250, 30-256-256-784, B=6000
c = dict(input_size=30, hidden_size=256, hidden2_size=256, output_size=784, activation_fn="relu", T_pc=250,
         optimizer_x_fn_pc=torch.optim.Adam, optimizer_x_kwargs_pc={"lr": 0.1})
m = um.get_model(c, True)
tr = get_pc_trainer(m, c, is_mcpc=True, training=False)
z = torch.zeros(6000, 30, device=dev); y = (torch.rand(6000, 784, device=dev) < 0.13).float()
r, dt = timed(lambda: tr.train_on_batch(inputs=z, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None},
                                        is_log_progress=False, is_return_results_every_t=True), 250)
print(f"cfg-PC MAP Adam-x T=250 B=6000 (energies every step): {r:10.0f} steps/s  ({dt*1e3:.1f} ms per call)")
