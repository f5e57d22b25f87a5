"""A script written the way the reference's figure_2.py:29-75 and figure_6.py:22-93 are written: module-level
`import predictive_coding as pc`, `from utils.model import ...`, `from utils.training_evaluation import ...`, `from utils.plotting import
...`, and models built ON THE CPU -- no `.cuda()`, no `use_cuda`: `pseudo_input = torch.zeros(batch_size, hidden_size)`.  It names
NOTHING of this repository.  Started next to the reference's packages it drives the reference's CPU loop; started through
`python -m montecarlopredictivecoding_amd.run cpu_built_models.py` it drives the MI355X engine (tensors staged per call).
Prints one JSON line."""
import json
import random

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

import predictive_coding as pc
from utils.model import sample_x_fn_cte, sample_x_fn_normal, fe_fn, random_step
from utils.training_evaluation import get_pc_trainer, get_mcpc_trainer, get_mcpc_trainer_one_sample
from utils.plotting import setup_fig                                          # the script's own utils/ (plots: not the engine's business)

random.seed(1)
np.random.seed(2)
torch.manual_seed(30)


def posterior_linear_model(n_chains=256):
    """figure_2.py:29-75: prior x ~ N(0.2, 1), y = 2 x + N(0, 1), y = 1 -> posterior N(0.44, 0.2)."""
    hidden_size, output_size = 1, 1
    data = torch.ones(n_chains, output_size)
    pseudo_input = torch.zeros(n_chains, hidden_size)
    gen_pc = nn.Sequential(
        nn.Linear(hidden_size, hidden_size),
        pc.PCLayer(sample_x_fn=sample_x_fn_cte),
        nn.Linear(hidden_size, output_size, bias=False),
    )
    gen_pc.train()
    nn.init.constant_(gen_pc[0].bias, 0.2)
    nn.init.constant_(gen_pc[2].weight, 2.)
    config = {
        "input_var": 1.,
        "T_pc": 500, "optimizer_x_fn_pc": optim.Adam, "optimizer_x_kwargs_pc": {"lr": 0.02},
        "mixing": 200, "sampling": 1800, "optimizer_x_kwargs_mcpc": {"lr": 0.02}, "optimizer_p_fn_mcpc": optim.Adam,
        "loss_fn": fe_fn,
    }
    pc_trainer = get_pc_trainer(gen_pc, config, is_mcpc=True, training=False)
    mcpc_trainer = get_mcpc_trainer(gen_pc, config, training=False)
    pc_trainer.train_on_batch(inputs=pseudo_input, loss_fn=config["loss_fn"], loss_fn_kwargs={'_target': data, '_var': config["input_var"]},
                              is_return_results_every_t=True, is_return_representations=True)
    x_map = gen_pc[1].get_x()[0, 0].item()
    mc_results = mcpc_trainer.train_on_batch(inputs=pseudo_input, loss_fn=config["loss_fn"],
                                             loss_fn_kwargs={'_target': data, '_var': config["input_var"]},
                                             callback_after_t=random_step, callback_after_t_kwargs={'_pc_trainer': mcpc_trainer},
                                             is_sample_x_at_batch_start=True, is_return_results_every_t=True, is_return_representations=True)
    samples = torch.stack(mc_results["representations"][config["mixing"]:]).numpy().reshape(-1)
    return {"map": x_map, "mean": float(samples.mean()), "var": float(samples.var()), "n_samples": int(samples.size),
            "x_device": str(gen_pc[1].get_x().device), "mode": getattr(mcpc_trainer, "last_call_mode", None),
            "map_mode": getattr(pc_trainer, "last_call_mode", None)}


def varying_langevin_noise(noise_var=2.0, epochs=10, n=25, batch_size=2048):
    """figure_6.py:22-93: learned |W0| = sqrt(2 var / noise_var - 1) for data N(1, 5)."""
    hidden_size, output_size = 1, 1
    start = [-7, -5]
    mu, var = 1., 5.
    datas = [mu + np.sqrt(var) * torch.randn(batch_size, output_size) for i in range(n)]
    pseudo_input = torch.zeros(batch_size, hidden_size)
    gen_mcpc = nn.Sequential(
        nn.Linear(hidden_size, hidden_size),
        pc.PCLayer(sample_x_fn=sample_x_fn_normal),
        nn.Linear(hidden_size, output_size, bias=False),
    )
    gen_mcpc.train()
    nn.init.constant_(gen_mcpc[0].bias, start[0])
    nn.init.constant_(gen_mcpc[2].weight, start[1])
    config_mcpc = {
        "input_var": 1.,
        "K": 150,
        "optimizer_x_kwargs_mcpc": {"lr": np.clip(0.01 * noise_var / 2, 0.001, 0.05)},
        "optimizer_p_fn_mcpc": optim.Adam,
        "optimizer_p_kwargs_mcpc": {"lr": np.clip(0.3 / noise_var, 1 / 2, 3)},
        "loss_fn": fe_fn,
    }
    mcpc_trainer = get_mcpc_trainer_one_sample(gen_mcpc, config_mcpc, training=True)
    traj = []
    for e in range(epochs):
        for data in datas:
            mcpc_trainer.train_on_batch(inputs=pseudo_input, loss_fn=config_mcpc["loss_fn"],
                                        loss_fn_kwargs={'_target': data, '_var': config_mcpc["input_var"]},
                                        callback_after_t=random_step, callback_after_t_kwargs={'_pc_trainer': mcpc_trainer, 'var': noise_var},
                                        is_sample_x_at_batch_start=True, is_log_progress=False, is_return_results_every_t=False,
                                        is_checking_after_callback_after_t=False)
            traj.append([gen_mcpc[0].bias.item(), gen_mcpc[2].weight.item()])
    traj = np.array(traj)
    grad = gen_mcpc[2].weight.grad
    return {"abs_w0": float(np.abs(traj[-50:, 1]).mean()), "mu_w0": float(np.mean(traj[-50:, 0] * traj[-50:, 1])),
            "grad_device": None if grad is None else str(grad.device), "param_device": str(gen_mcpc[2].weight.device),
            "x_device": str(gen_mcpc[1].get_x().device), "mode": getattr(mcpc_trainer, "last_call_mode", None)}


if __name__ == "__main__":
    out = {"pc_module": pc.__name__, "pc_file": pc.__file__, "trainer_module": pc.PCTrainer.__module__,
           "random_step_module": random_step.__module__, "setup_fig": setup_fig()}
    out["fig2"] = posterior_linear_model()
    out["fig6"] = varying_langevin_noise()
    print(json.dumps(out))
