#!/usr/bin/env python3
"""What staging a CPU-built model costs per call: the same cfg-M calls through the facade with the model on the GPU
and with the model on the CPU (W, b, x, inputs, target copied to the device per call; x, results, param.grad copied back)."""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import montecarlopredictivecoding_amd.utils.model as um  # noqa: E402
from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer, get_pc_trainer  # noqa: E402

warnings.simplefilter("ignore")
cfg = dict(input_size=30, hidden_size=256, hidden2_size=256, output_size=784, activation_fn="relu", T_pc=250, optimizer_x_fn_pc=torch.optim.Adam,
           optimizer_x_kwargs_pc={"lr": 0.1}, mixing=50, sampling=100, optimizer_x_kwargs_mcpc={"lr": 0.03}, optimizer_p_fn_mcpc=torch.optim.Adam,
           optimizer_p_kwargs_mcpc={"lr": 0.001}, loss_fn=um.bernoulli_fn, input_var=None)
for B in (256, 6000):
    for dev in ("cuda:0", "cpu"):
        torch.manual_seed(0)
        m = um.get_model(cfg, dev != "cpu")
        y = (torch.rand(B, 784) < 0.13).float().to(dev)
        inp = torch.zeros(B, 30, device=dev)
        pc_tr, mc_tr = get_pc_trainer(m, cfg, is_mcpc=True, training=False), get_mcpc_trainer(m, cfg, training=True)

        def it():
            pc_tr.train_on_batch(inputs=inp, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None}, is_log_progress=False,
                                 is_return_results_every_t=False, is_checking_after_callback_after_t=False)
            mc_tr.train_on_batch(inputs=inp, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None}, callback_after_t=um.random_step,
                                 callback_after_t_kwargs={"_pc_trainer": mc_tr}, is_sample_x_at_batch_start=False, is_log_progress=False,
                                 is_return_results_every_t=False, is_checking_after_callback_after_t=False)
        for _ in range(3):
            it()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            it()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"recipe iteration (MAP 250 + MCPC learning 150 steps, cfg-M net), batch {B:5d}, model on {dev:6s}: {dt * 1e3:7.2f} ms per iteration "
              f"({pc_tr.last_call_mode} / {mc_tr.last_call_mode})", flush=True)
