"""Port vs imported reference: Langevin steps/s on the same inputs, same host.  TEST INFRASTRUCTURE ONLY.

    python -m oracle.compare_speed [--batch 6000] [--steps 30] [--threads 8 1]

BASELINE.md section 3.1: `bench.py`'s `cpu_baseline` times `oracle/torch_port.py` (kind = "port") on the GPU box, because the
reference's Python cannot travel there.  This script, which runs only in the build container (it imports /root/reference),
shows that the port is a time-faithful stand-in: the imported reference (`PCTrainer.train_on_batch` + `random_step`, driven
exactly as the figure scripts drive it) and the port run the same cfg-M call -- 30-256-256-784 ReLU, Bernoulli read-out,
B chains, SGD-x lr 0.03 + Langevin noise var 2, parameter grads by autograd every step, accumulated from T/5 on -- back to
back, at each thread count.  It prints one JSON line; the numbers of the committed run are in BASELINE.md section 3.
"""
import argparse
import json
import os
import time
import warnings

import numpy as np
import torch

from oracle import torch_port
from oracle.gen_golden import import_reference

SIZES, N_IN, N_OUT = [30, 256, 256], 30, 784


def make_problem(batch, seed=30):
    g = torch.Generator().manual_seed(seed)
    dims = [N_IN] + SIZES + [N_OUT]
    W, b = [], []
    for j in range(len(dims) - 1):
        k = 1.0 / dims[j] ** 0.5
        W.append(((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) * k).numpy())
        b.append(((torch.rand(dims[j + 1], generator=g) * 2 - 1) * k).numpy())
    y = (torch.rand(batch, N_OUT, generator=g) < 0.13).float().numpy()
    xs = [((torch.rand(batch, n, generator=g) * 2 - 1) * 10.0).numpy() for n in SIZES]
    return W, b, y, xs


def time_reference(pc, um, W, b, y, xs, T, lr=0.03):
    import torch.nn as nn
    import torch.optim as optim
    x0 = [torch.from_numpy(x) for x in xs]
    mods = []
    dims = [N_IN] + SIZES + [N_OUT]
    for l in range(3):
        mods += [nn.Linear(dims[l], dims[l + 1]), pc.PCLayer(sample_x_fn=(lambda inp, v=x0[l]: v.clone())), nn.ReLU()]
    mods.append(nn.Linear(dims[3], dims[4]))
    model = nn.Sequential(*mods)
    lins = [m for m in model if isinstance(m, nn.Linear)]
    with torch.no_grad():
        for lin, w, v in zip(lins, W, b):
            lin.weight.copy_(torch.from_numpy(w)); lin.bias.copy_(torch.from_numpy(v))
    model.train()
    trainer = pc.PCTrainer(model, T=T, update_x_at="all", optimizer_x_fn=optim.SGD, optimizer_x_kwargs={"lr": lr},
                           update_p_at="never", accumulate_p_at=list(range(T // 5, T)), optimizer_p_fn=optim.SGD,
                           optimizer_p_kwargs={"lr": 0.0}, plot_progress_at=[])
    kw = dict(inputs=torch.zeros(y.shape[0], N_IN), loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": torch.from_numpy(y), "_var": None},
              callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": trainer}, is_log_progress=False,
              is_return_results_every_t=False, is_checking_after_callback_after_t=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        trainer.train_on_batch(**kw)
        return T / (time.perf_counter() - t0)


def time_port(W, b, y, xs, T, lr=0.03):
    model, nodes, lins = torch_port.build(SIZES, [1, 1, 1], N_IN, N_OUT, W, b)
    loss_fn = torch_port.make_loss("bernoulli", y)
    inputs = torch.zeros(y.shape[0], N_IN)
    t0 = time.perf_counter()
    torch_port.run(model, nodes, lins, inputs, xs, loss_fn, T, lr, noise_var=2.0, acc_begin=T // 5, record_energy=False)
    return T / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=6000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--threads", type=int, nargs="+", default=[8, 1])
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args()
    pc, um = import_reference()
    W, b, y, xs = make_problem(args.batch)
    rows = []
    for nt in args.threads:
        torch.set_num_threads(nt)
        time_reference(pc, um, W, b, y, xs, 4); time_port(W, b, y, xs, 4)          # warm-up
        ref, port = [], []
        for _ in range(args.repeats):                                              # interleaved: same thermal / load state
            ref.append(time_reference(pc, um, W, b, y, xs, args.steps))
            port.append(time_port(W, b, y, xs, args.steps))
        r, p = float(np.median(ref)), float(np.median(port))
        rows.append({"threads": nt, "reference_steps_per_s": r, "port_steps_per_s": p, "port_over_reference": p / r})
    print(json.dumps({"batch": args.batch, "steps_per_timing": args.steps, "repeats": args.repeats, "host_cpus": os.cpu_count(),
                      "torch": torch.__version__, "rows": rows}))


if __name__ == "__main__":
    main()
