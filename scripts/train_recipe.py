#!/usr/bin/env python3
"""Cost of ONE TRAINING ITERATION of the reference's MNIST recipe through the facade (developer measurement for DESIGN.md).

The recipe is the `mcpc_ml_*` configuration (/root/reference/table_1.py:195-212) driven the way the learning loops of the
reference drive a batch (figure_4.py:387-389): a MAP warm-up call (Adam on x, T_pc = 250, no parameter update), then the
MCPC learning call (mixing 50 + sampling 100 Langevin steps, parameter gradients accumulated over the sampling steps, one
optimizer_p step at the end).  Net 20-128-128-784 ReLU, Bernoulli read-out, batch 256 (and larger batches for comparison).
Short calls on a small batch: what is measured here is the fixed cost per call (binding, packing, read-out, the Python
facade), not the step kernel.

    python3 scripts/train_recipe.py [--batches 256 2048] [--iters 20] [--profile]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlopredictivecoding_amd.utils import model as um  # noqa: E402
from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer, get_pc_trainer  # noqa: E402

warnings.simplefilter("ignore")
dev = "cuda:0"


def make(batch):
    cfg = dict(input_size=20, hidden_size=128, hidden2_size=128, output_size=784, activation_fn="relu", loss_fn=um.bernoulli_fn,
               input_var=None, T_pc=250, optimizer_x_fn_pc=torch.optim.Adam, optimizer_x_kwargs_pc={"lr": 0.1},
               mixing=50, sampling=100, optimizer_x_kwargs_mcpc={"lr": 0.03},
               optimizer_p_fn_mcpc=torch.optim.Adam, optimizer_p_kwargs_mcpc={"lr": 0.001})
    model = um.get_model(cfg, True)
    model.train()
    tr_map = get_pc_trainer(model, cfg, is_mcpc=True)
    tr_mc = get_mcpc_trainer(model, cfg, training=True)
    z = torch.zeros(batch, cfg["input_size"], device=dev)
    data = [(torch.rand(batch, 784, device=dev) < 0.13).float() for _ in range(4)]
    return cfg, model, tr_map, tr_mc, z, data


def iteration(cfg, tr_map, tr_mc, z, y):
    tr_map.train_on_batch(inputs=z, loss_fn=cfg["loss_fn"], loss_fn_kwargs={"_target": y, "_var": None}, is_log_progress=False,
                          is_return_results_every_t=False, is_checking_after_callback_after_t=False)
    tr_mc.train_on_batch(inputs=z, loss_fn=cfg["loss_fn"], loss_fn_kwargs={"_target": y, "_var": None},
                         callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": tr_mc},
                         is_sample_x_at_batch_start=False, is_log_progress=False, is_return_results_every_t=False,
                         is_checking_after_callback_after_t=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[256, 2048, 6000])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    for B in args.batches:
        cfg, model, tr_map, tr_mc, z, data = make(B)
        for i in range(3):
            iteration(cfg, tr_map, tr_mc, z, data[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.iters):
            iteration(cfg, tr_map, tr_mc, z, data[i % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        steps = cfg["T_pc"] + cfg["mixing"] + cfg["sampling"]
        # the two calls separately
        parts = []
        for which in (0, 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(args.iters):
                y = data[i % 4]
                if which == 0:
                    tr_map.train_on_batch(inputs=z, loss_fn=cfg["loss_fn"], loss_fn_kwargs={"_target": y, "_var": None}, is_log_progress=False,
                                          is_return_results_every_t=False, is_checking_after_callback_after_t=False)
                else:
                    tr_mc.train_on_batch(inputs=z, loss_fn=cfg["loss_fn"], loss_fn_kwargs={"_target": y, "_var": None},
                                         callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": tr_mc},
                                         is_sample_x_at_batch_start=False, is_log_progress=False, is_return_results_every_t=False,
                                         is_checking_after_callback_after_t=False)
            torch.cuda.synchronize(); parts.append((time.perf_counter() - t0) / args.iters)
        print(f"train recipe B={B:5d}: {dt*1e3:7.2f} ms per iteration ({steps} steps: {dt/steps*1e6:6.1f} us per step all-in; "
              f"{B/dt:9.0f} images/s); MAP call {parts[0]*1e3:6.2f} ms, MCPC learning call {parts[1]*1e3:6.2f} ms", flush=True)
        if args.profile:
            pr = cProfile.Profile(); pr.enable()
            for i in range(args.iters):
                iteration(cfg, tr_map, tr_mc, z, data[i % 4])
            torch.cuda.synchronize(); pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
