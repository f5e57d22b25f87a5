"""Loss functions, x initialisers, the Langevin callback and the MLP factory.

Counterpart of /root/reference/utils/model.py:8-69 (same names, arguments and numerical meaning).
Every callable that the HIP engine can fuse carries a ``_mcpc`` tag; the torch bodies below are what
the tags *mean* (they are executed only by the opt-in generic path and by the loss recogniser's
self-check), the fused path maps the tag to a kernel epilogue:

    fe_fn / fe_fn_mask            -> MCPC_LOSS_GAUSSIAN  (+ mask_start)
    bernoulli_fn / *_mask         -> MCPC_LOSS_BERNOULLI (+ mask_start)
    zero_fn                       -> MCPC_LOSS_NONE
    random_step                   -> MCPC_NOISE_PHILOX inside the x update
"""
import numpy as np
import torch
import torch.nn as nn

from ..predictive_coding.pc_layer import PCLayer


def __getattr__(name):
    """A name of the script's own utils/model.py that this module does not define (the reference's has none: every name of its
    utils/model.py:8-163 is here) comes from the script's own module when it runs under the launcher (run.py: script_own_attr)."""
    if name.startswith("__"):
        raise AttributeError(name)
    from ..run import script_own_attr
    return script_own_attr("utils.model", name)


# ---- x initialisers (reference utils/model.py:8-15) --------------------------------------------------
def sample_x_fn(inputs):
    return inputs["mu"].detach().clone().uniform_(-10.0, 10.0)


def sample_x_fn_normal(inputs):
    return torch.randn_like(inputs["mu"])


def sample_x_fn_cte(inputs):
    return 3 * torch.ones_like(inputs["mu"])


# ---- losses (reference utils/model.py:17-33) --------------------------------------------------------
def _last_columns(n_out, perc):
    return round(n_out * perc)


def fe_fn(output, _target, _var):
    return (1 / _var) * 0.5 * (output - _target).pow(2).sum()


def bernoulli_fn(output, _target, _var=None, _reduction="sum"):
    return nn.functional.binary_cross_entropy_with_logits(output, _target, reduction=_reduction)


def fe_fn_mask(output, _target, _var, perc=0.5):
    k = _last_columns(output.shape[1], perc)
    return (1 / _var) * 0.5 * (output[:, -k:] - _target[:, -k:]).pow(2).sum()


def bernoulli_fn_mask(output, _target, _var=None, perc=0.5):
    k = _last_columns(output.shape[1], perc)
    return nn.functional.binary_cross_entropy_with_logits(output[:, -k:], _target[:, -k:], reduction="sum")


def zero_fn(output):
    return torch.tensor(0.0)


fe_fn._mcpc = dict(loss="gaussian", masked=False)
fe_fn_mask._mcpc = dict(loss="gaussian", masked=True)
bernoulli_fn._mcpc = dict(loss="bernoulli", masked=False)
bernoulli_fn_mask._mcpc = dict(loss="bernoulli", masked=True)
zero_fn._mcpc = dict(loss="none", masked=False)


# ---- the Langevin kick (reference utils/model.py:35-44) -------------------------------------------------
def random_step(t, _pc_trainer, var=2.0):
    """x <- x + sqrt(var*lr)*xi.  ``var`` must be 2 for a correct posterior sampler.

    Passed as ``callback_after_t`` it is recognised by its tag and fused into the HIP x-update
    (counter-based Philox noise, no callback is actually invoked per step).  When it IS invoked
    (step-wise path, i.e. next to other user callbacks) it does what the reference does: overwrite
    every x.grad with N(0, sqrt(var/lr)) and step the x optimizer once more.
    """
    xs = _pc_trainer.get_model_xs()
    optimizer = _pc_trainer.get_optimizer_x()
    std = np.sqrt(var / optimizer.defaults["lr"])
    for x in xs:
        if x.grad is None:
            x.grad = torch.zeros_like(x)
        x.grad.normal_(0.0, std)
    optimizer.step()


random_step._mcpc = dict(langevin=True)


# ---- model factory (reference utils/model.py:47-69) -----------------------------------------------------
def get_model(config, use_cuda, sample_x_fn=sample_x_fn):
    act = {"relu": nn.ReLU, "tanh": nn.Tanh}[config["activation_fn"]]
    n1, n2, n3, n0 = config["input_size"], config["hidden_size"], config["hidden2_size"], config["output_size"]
    gen_pc = nn.Sequential(
        nn.Linear(n1, n1), PCLayer(sample_x_fn=sample_x_fn), act(),
        nn.Linear(n1, n2), PCLayer(sample_x_fn=sample_x_fn), act(),
        nn.Linear(n2, n3), PCLayer(sample_x_fn=sample_x_fn), act(),
        nn.Linear(n3, n0),
    )
    gen_pc.train()
    if use_cuda:
        gen_pc.cuda()
    return gen_pc


# ---- representations for down-stream probes (reference utils/model.py:71-163) ----------------------------
def get_representations(gen_pc, config, trainers, loader, rep_type="MAP", use_cuda=False, n=None):
    """Top-latent-layer representations of every batch of ``loader`` as a ``TensorDataset``.

    rep_type "MAP"          x_1 after PC (MAP) inference with ``trainers[0]``;
             "expectation"  mean of x_1 over ALL recorded Langevin steps of ``trainers[1]`` (started from the MAP state);
             "full"         every ``sampling/n``-th sample after the mixing phase, labels repeated ``n`` times.
    The Langevin trajectories come back from the engine's device-side record buffer in one copy.
    """
    from torch.utils.data import TensorDataset
    device = next(gen_pc.parameters()).device
    input_size = len(gen_pc[0].bias)
    reps, labels = [], []
    if rep_type != "MAP":
        if len(trainers) != 2:
            raise NotImplementedError
        assert rep_type in ("full", "expectation")
    pc_trainer = trainers[0]
    stride = 1
    if rep_type == "full":
        if n is not None:
            stride = int(config["sampling"] / n)
        else:
            n = config["sampling"]
    for data, label in loader:
        pseudo_input = torch.zeros(data.shape[0], input_size, device=device)
        data, label = data.to(device), label.to(device)
        kw = dict(inputs=pseudo_input, loss_fn=config["loss_fn"],
                  loss_fn_kwargs={"_target": data, "_var": config["input_var"]},
                  is_return_results_every_t=False, is_checking_after_callback_after_t=False)
        pc_trainer.train_on_batch(is_log_progress=(rep_type == "MAP"), **kw)
        if rep_type == "MAP":
            reps.append(gen_pc[1].get_x().detach().clone())
            labels.append(label)
            continue
        mcpc_trainer = trainers[1]
        kw["is_return_results_every_t"] = True
        results = mcpc_trainer.train_on_batch(
            callback_after_t=random_step, callback_after_t_kwargs={"_pc_trainer": mcpc_trainer},
            is_log_progress=False, is_sample_x_at_batch_start=False, is_return_representations=True, **kw)
        traj = torch.stack(results["representations"]).to(device)          # [T, B, n_1]
        if rep_type == "expectation":
            reps.append(traj.mean(0))
            labels.append(label)
        else:
            kept = traj[config["mixing"]::stride]
            reps.append(kept.reshape(-1, traj.shape[2]))
            labels.append(label.repeat(n))
    return TensorDataset(torch.cat(reps, dim=0), torch.cat(labels, dim=0))
