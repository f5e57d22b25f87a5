"""Trainer factories: the three trainer shapes the scripts use.

Counterpart of /root/reference/utils/training_evaluation.py:16-70 (same config keys).
"""
import torch.optim as optim

from .. import predictive_coding as pc


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
