"""A script written the way the reference's figure scripts are written -- module-level `import predictive_coding as pc`,
`from utils.model import *`, factories from `utils.training_evaluation`, a config dict, a MAP call then an MCPC call with
`random_step` as `callback_after_t` -- for the linear-Gaussian toy of figure 2 (prior x ~ N(0.2, 1), y = 2 x + N(0, 1), y = 1:
posterior N(0.44, 0.2)).  It names NOTHING of this repository: run with `PYTHONPATH=<repo>/compat` it drives
the MI355X engine, run next to the reference's own packages it drives those.  Prints one JSON line."""
import json
import random

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

import predictive_coding as pc
from utils.model import *                                                    # noqa: F401,F403  (sample_x_fn_cte, fe_fn, random_step ...)
from utils.training_evaluation import get_pc_trainer, get_mcpc_trainer

random.seed(1)
np.random.seed(2)
torch.manual_seed(30)

use_cuda = torch.cuda.is_available()
device = "cuda" if use_cuda else "cpu"
n_chains = 256

gen_pc = nn.Sequential(
    nn.Linear(1, 1),
    pc.PCLayer(sample_x_fn=sample_x_fn_cte),                                 # noqa: F405
    nn.Linear(1, 1, bias=False),
)
gen_pc.train()
nn.init.constant_(gen_pc[0].bias, 0.2)
nn.init.constant_(gen_pc[2].weight, 2.)
gen_pc.to(device)

config = {
    "input_var": 1.,
    "T_pc": 500, "optimizer_x_fn_pc": optim.Adam, "optimizer_x_kwargs_pc": {"lr": 0.02},
    "mixing": 200, "sampling": 1800, "optimizer_x_kwargs_mcpc": {"lr": 0.02}, "optimizer_p_fn_mcpc": optim.Adam,
    "loss_fn": fe_fn,                                                        # noqa: F405
}
data = torch.ones(n_chains, 1, device=device)
pseudo_input = torch.zeros(n_chains, 1, device=device)

pc_trainer = get_pc_trainer(gen_pc, config, is_mcpc=True, training=False)
mcpc_trainer = get_mcpc_trainer(gen_pc, config, training=False)

pc_results = pc_trainer.train_on_batch(inputs=pseudo_input, loss_fn=config["loss_fn"], loss_fn_kwargs={'_target': data, '_var': config["input_var"]},
                                       is_log_progress=False, is_return_results_every_t=False)
x_map = gen_pc[1].get_x()[0, 0].item()
mc_results = mcpc_trainer.train_on_batch(inputs=pseudo_input, loss_fn=config["loss_fn"], loss_fn_kwargs={'_target': data, '_var': config["input_var"]},
                                         callback_after_t=random_step, callback_after_t_kwargs={'_pc_trainer': mcpc_trainer},      # noqa: F405
                                         is_sample_x_at_batch_start=False, is_log_progress=False, is_return_results_every_t=True,
                                         is_return_representations=True)
samples = torch.stack(mc_results["representations"][config["mixing"]:]).numpy().reshape(-1)
print(json.dumps({"pc_module": pc.__file__, "map": x_map, "mean": float(samples.mean()), "var": float(samples.var()),
                  "n_samples": int(samples.size), "mode": getattr(mcpc_trainer, "last_call_mode", None)}))
