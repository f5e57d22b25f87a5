"""Trainer factories: the three trainer shapes the scripts use.

Counterpart of /root/reference/utils/training_evaluation.py:16-70 (same config keys).
"""
import torch.optim as optim

from .. import predictive_coding as pc
# (the reference's module carries these names too -- figure_4.py:13 imports fe_fn from utils.training_evaluation)
from .model import bernoulli_fn, bernoulli_fn_mask, fe_fn, fe_fn_mask        # noqa: F401


def __getattr__(name):
    """Names of the reference's utils/training_evaluation.py that are not on the hot path (train, test, MNIST_LinearClassifier, KLdivergence,
    kl_divergence_discrete, get_paired_stat, get_fid): when a script runs under the launcher they come from the script's own module of that
    name (montecarlopredictivecoding_amd/run.py: script_own_attr); otherwise they do not exist here."""
    if name.startswith("__"):
        raise AttributeError(name)
    from ..run import script_own_attr
    return script_own_attr("utils.training_evaluation", name)


def get_pc_trainer(gen_pc, config, is_mcpc=False, training=True):
    """MAP / PC inference: T_pc steps, user optimizer on x, no noise (reference :16-39)."""
    common = dict(T=config["T_pc"], update_x_at="all", optimizer_x_fn=config["optimizer_x_fn_pc"],
                  optimizer_x_kwargs=config["optimizer_x_kwargs_pc"], early_stop_condition="False",
                  plot_progress_at=[])
    if is_mcpc:
        return pc.PCTrainer(gen_pc, update_p_at="never", **common)
    return pc.PCTrainer(gen_pc, update_p_at="last" if training else "never",
                        optimizer_p_fn=config["optimizer_p_fn"], optimizer_p_kwargs=config["optimizer_p_kwargs"],
                        **common)


def get_mcpc_trainer(gen_pc, config, training=True):
    """MCPC: mixing + sampling Langevin steps, gradients accumulated over the sampling steps (reference :43-56)."""
    mixing, sampling = config["mixing"], config["sampling"]
    return pc.PCTrainer(
        gen_pc, T=mixing + sampling, update_x_at="all", optimizer_x_fn=optim.SGD,
        optimizer_x_kwargs=config["optimizer_x_kwargs_mcpc"],
        update_p_at="last" if training else "never",
        accumulate_p_at=[mixing + i for i in range(sampling)],
        optimizer_p_fn=config["optimizer_p_fn_mcpc"] if training else optim.SGD,
        optimizer_p_kwargs=config["optimizer_p_kwargs_mcpc"] if training else {"lr": 0.0},
        plot_progress_at=[])


def get_mcpc_trainer_one_sample(gen_pc, config, training=True):
    """MCPC with a single Monte Carlo sample: K steps, update from the last one only (reference :58-70)."""
    return pc.PCTrainer(
        gen_pc, T=config["K"], update_x_at="all", optimizer_x_fn=optim.SGD,
        optimizer_x_kwargs=config["optimizer_x_kwargs_mcpc"],
        update_p_at="last" if training else "never",
        optimizer_p_fn=config["optimizer_p_fn_mcpc"] if training else optim.SGD,
        optimizer_p_kwargs=config["optimizer_p_kwargs_mcpc"] if training else {"lr": 0.0},
        plot_progress_at=[])


# ---- ancestral sampler and evaluators (reference utils/training_evaluation.py:72-100,140-206) -------------
def _cpu_normal_like(temp):
    """Standard normals of temp's shape drawn from torch's CPU generator and moved to temp's device -- the stream the
    reference consumes (training_evaluation.py:73-79 draws `torch.randn((N, n, 1))` on the CPU and `.cuda()`s it), so that a
    `torch.manual_seed` pins the ancestral sample on the GPU exactly as it does in the reference."""
    import torch
    return torch.randn(tuple(temp.shape)).to(temp.device)


def sample_pc(num_samples, model, config, use_cuda=False, is_return_hidden=False):
    """Ancestral sample of the generative model: unit-variance Gaussian noise is added at every PCLayer
    (the prior of a PC layer with the default energy), the read-out is sampled from the sensory
    distribution (Gaussian with ``input_var`` / Bernoulli of the logits) unless ``is_return_hidden``."""
    import torch
    from ..utils.model import bernoulli_fn, fe_fn
    device = next(model.parameters()).device
    temp = torch.zeros(num_samples, config["input_size"], device=device)
    with torch.no_grad():
        for module in model:
            if isinstance(module, pc.PCLayer):
                temp = temp + _cpu_normal_like(temp)
            else:
                temp = module(temp)
        if is_return_hidden:
            return temp.detach()
        if config["loss_fn"] is fe_fn:
            temp = temp + (config["input_var"] ** 0.5) * _cpu_normal_like(temp)
        elif config["loss_fn"] is bernoulli_fn:
            temp = (torch.rand_like(temp) <= temp.sigmoid()).double()
    return temp.detach()


def get_mse_rec(gen_pc, config, dataloader, use_cuda):
    """Masked-reconstruction error: infer (MAP) from the bottom half of each image, score the top half."""
    import torch
    from ..utils.model import bernoulli_fn, bernoulli_fn_mask, fe_fn, fe_fn_mask
    loss_fn = {fe_fn: fe_fn_mask, bernoulli_fn: bernoulli_fn_mask}[config["loss_fn"]]
    gen_pc.train()
    device = next(gen_pc.parameters()).device
    pc_trainer = get_pc_trainer(gen_pc, config, training=False, is_mcpc=True)
    mse, n_data = 0.0, 0
    for data, _ in dataloader:
        data = data.to(device)
        pseudo_input = torch.zeros(data.shape[0], config["input_size"], device=device)
        pc_trainer.train_on_batch(inputs=pseudo_input, loss_fn=loss_fn,
                                  loss_fn_kwargs={"_target": data, "_var": config["input_var"]},
                                  is_log_progress=False, is_return_results_every_t=False,
                                  is_checking_after_callback_after_t=False)
        with torch.no_grad():
            img = gen_pc[-1](gen_pc[-2](gen_pc[-3].get_x().detach()))
            if config["loss_fn"] is bernoulli_fn:
                img = (img > 0).type_as(img)          # logits: threshold at 0
            half = round(data.shape[1] / 2)
            mse += float(((img[:, :-half] - data[:, :-half]) ** 2).mean(1).sum())
        n_data += data.shape[0]
    return mse / n_data


def marginal_likelihood_from_logits(logits, dataloader):
    """log (1/S) sum_s p(y | o_s), averaged over the data, for read-out logits o_s [S, n0] (clamped to +-20 as the reference
    does, training_evaluation.py:177) -- the likelihood core of get_marginal_likelihood as a function of the prior samples.
    BCE(o, y) summed over pixels = sum softplus(o) - y.o, so the [B, S, n0] tensor the reference materialises per batch
    (:190-193) becomes one [B, n0] x [n0, S] product."""
    import torch
    logits = logits.clamp(-20, 20)
    sp = torch.nn.functional.softplus(logits).sum(1)                                                       # sum_j log(1+e^o)
    total, count = 0.0, 0
    with torch.no_grad():
        for data, _ in dataloader:
            data = data.to(logits.device).to(logits.dtype)
            nll = sp.unsqueeze(0) - data @ logits.t()                                                       # [B, S]
            m = nll.min(1).values
            p = torch.exp(-(nll - m.unsqueeze(1))).mean(1)
            total += float((torch.log(p) - m).sum())
            count += data.shape[0]
    return total / count


def get_marginal_likelihood(gen_pc, config, dataloader, use_cuda, n_samples=5000):
    """Importance-sampled log marginal likelihood per datum (Bernoulli read-out), samples from the prior."""
    from ..utils.model import bernoulli_fn
    if config["loss_fn"] is not bernoulli_fn:
        raise NotImplementedError("the reference implements the Bernoulli read-out only (training_evaluation.py:196)")
    logits = sample_pc(n_samples, gen_pc, config, use_cuda=use_cuda, is_return_hidden=True)                 # [S, n0]
    return marginal_likelihood_from_logits(logits, dataloader)
