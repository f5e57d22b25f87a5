"""Reference-pinned fixtures for the rows SURVEY.md section 8f ranks after the hot path: the evaluators
(`sample_pc`, `get_mse_rec`, `get_marginal_likelihood`), a shipped checkpoint, and the toy learning runs of
figure_4 / figure_6.  TEST INFRASTRUCTURE ONLY; runs only in the build container (imports /root/reference).

    python -m oracle.gen_golden_eval            # writes tests/golden/g10_* g11_* g12_* g14_*
    python -m oracle.gen_golden_eval representations      # g14_representations only

What is stored is data: seeded inputs (philox), the reference's outputs, and -- for g11 -- one of the reference's
shipped weight files copied byte for byte (`models/mcpc_fid_3`, a `torch.save`d state_dict: a data file, loaded
by the reference's own scripts with `load_state_dict(..., strict=False)`, table_1.py:76).

Import stubs: `seaborn` (plots only, as in gen_golden.py) and `torchvision` (`utils/training_evaluation.py:6` imports
`save_image` for the FID dump, `utils/data.py:5` the MNIST loaders) -- neither is on any path exercised here.
"""
import json
import os
import shutil
import sys
import types
import warnings

import numpy as np

from oracle import philox
from oracle.cases import make_params
from oracle.gen_golden import GOLDEN, REF, import_reference


def import_reference_eval():
    pc, um = import_reference()
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.utils = types.ModuleType("torchvision.utils")
        tv.utils.save_image = lambda *a, **k: None
        tv.transforms = types.ModuleType("torchvision.transforms")
        tv.datasets = types.ModuleType("torchvision.datasets")
        sys.modules.update({"torchvision": tv, "torchvision.utils": tv.utils, "torchvision.transforms": tv.transforms,
                            "torchvision.datasets": tv.datasets})
    import utils.training_evaluation as te
    assert te.__file__.startswith(REF), te.__file__
    return pc, um, te


EVAL_SHAPE = dict(input_size=8, hidden_size=32, hidden2_size=32, output_size=64)


def eval_config(um, loss):
    import torch.optim as optim
    cfg = dict(EVAL_SHAPE, activation_fn="relu", loss_fn=um.bernoulli_fn if loss == "bernoulli" else um.fe_fn,
               input_var=None if loss == "bernoulli" else 0.3, T_pc=60, optimizer_x_fn_pc=optim.Adam,
               optimizer_x_kwargs_pc={"lr": 0.1})
    return cfg


def eval_inputs(seed, loss, n_data=48):
    W, b = make_params(seed, EVAL_SHAPE["input_size"], [EVAL_SHAPE["input_size"], EVAL_SHAPE["hidden_size"], EVAL_SHAPE["hidden2_size"]],
                       EVAL_SHAPE["output_size"])
    u = philox.uniform_pm(seed, 300, (n_data, EVAL_SHAPE["output_size"]), 0.0, 1.0)
    data = (u < 0.3).astype(np.float32) if loss == "bernoulli" else (2.0 * u - 0.5).astype(np.float32)
    labels = (np.arange(n_data) % 10).astype(np.int64)
    return W, b, data, labels


def load_params(model, W, b):
    import torch
    import torch.nn as nn
    lins = [m for m in model if isinstance(m, nn.Linear)]
    with torch.no_grad():
        for lin, w, v in zip(lins, W, b):
            lin.weight.copy_(torch.from_numpy(w))
            lin.bias.copy_(torch.from_numpy(v))


def gen_evaluators(pc, um, te, loss, seed):
    """sample_pc / get_mse_rec / get_marginal_likelihood of the reference on a small net with seeded weights.
    x0 = sample_x_fn_cte (3 everywhere, utils/model.py:14-15) so that the MAP inference inside get_mse_rec is deterministic."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    cfg = eval_config(um, loss)
    W, b, data, labels = eval_inputs(seed, loss)
    model = um.get_model(cfg, False, sample_x_fn=um.sample_x_fn_cte)
    load_params(model, W, b)
    loader = DataLoader(TensorDataset(torch.from_numpy(data), torch.from_numpy(labels)), batch_size=16)
    blob = {"meta_json": np.array(json.dumps(dict(loss=loss, seed=seed, shape=EVAL_SHAPE, T_pc=cfg["T_pc"], lr=0.1,
                                                  input_var=cfg["input_var"], batch_size=16, torch_seed=7,
                                                  n_hidden_samples=40, n_ml_samples=600)))}
    blob["data"], blob["labels"] = data, labels
    for j, (w, v) in enumerate(zip(W, b)):
        blob[f"W{j}"], blob[f"b{j}"] = w, v
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        blob["mse_rec"] = np.float64(float(te.get_mse_rec(model, cfg, loader, False)))
        # the read-out of the MAP state per batch, before thresholding: the same loop as training_evaluation.py:152-168
        loss_fn = um.fe_fn_mask if loss == "gaussian" else um.bernoulli_fn_mask
        tr = te.get_pc_trainer(model, cfg, training=False, is_mcpc=True)
        imgs = []
        for d, _ in loader:
            tr.train_on_batch(inputs=torch.zeros(d.shape[0], cfg["input_size"]), loss_fn=loss_fn,
                              loss_fn_kwargs={"_target": d, "_var": cfg["input_var"]}, is_log_progress=False,
                              is_return_results_every_t=False, is_checking_after_callback_after_t=False)
            imgs.append(model[-1](model[-2](model[-3].get_x().detach())).detach().numpy().copy())
        blob["map_readout"] = np.concatenate(imgs, 0)
    torch.manual_seed(7)
    blob["hidden_samples"] = te.sample_pc(40, model, cfg, use_cuda=False, is_return_hidden=True).numpy().copy()
    if loss == "bernoulli":
        torch.manual_seed(7)
        blob["ml_logits"] = te.sample_pc(600, model, cfg, use_cuda=False, is_return_hidden=True).numpy().copy()
        torch.manual_seed(7)
        blob["ml"] = np.float64(float(te.get_marginal_likelihood(model, cfg, loader, False, n_samples=600)))
        torch.manual_seed(7)
        s = te.sample_pc(40, model, cfg, use_cuda=False)
        blob["bernoulli_sample_mean"] = np.float64(float(s.mean()))
    path = os.path.join(GOLDEN, f"g10_eval_{loss}.npz")
    np.savez_compressed(path, **blob)
    return path


CKPT = "mcpc_fid_3"      # 20-128-128-784 ReLU, Bernoulli read-out (table_1.py:50-71); 0.6 MB incl. the stale `_x` of a batch of 96


def gen_checkpoint(pc, um, te):
    """Load a shipped checkpoint into the reference's get_model (strict=False, table_1.py:76) and record T = 20 Adam MAP
    steps (the warm-up every figure runs first) on 32 seeded binary images, x0 = 3."""
    import torch
    import torch.optim as optim
    src = os.path.join(REF, "models", CKPT)
    dst = os.path.join(GOLDEN, f"g11_ref_ckpt_{CKPT}.pt")
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)
    cfg = dict(input_size=20, hidden_size=128, hidden2_size=128, output_size=784, activation_fn="relu", loss_fn=um.bernoulli_fn,
               input_var=None, T_pc=20, optimizer_x_fn_pc=optim.Adam, optimizer_x_kwargs_pc={"lr": 0.1})
    model = um.get_model(cfg, False, sample_x_fn=um.sample_x_fn_cte)
    sd = torch.load(src, map_location="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    data = (philox.uniform_pm(11, 300, (32, 784), 0.0, 1.0) < 0.13).astype(np.float32)
    tr = te.get_pc_trainer(model, cfg, training=False, is_mcpc=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = tr.train_on_batch(inputs=torch.zeros(32, 20), loss_fn=um.bernoulli_fn,
                                loss_fn_kwargs={"_target": torch.from_numpy(data), "_var": None}, is_log_progress=False,
                                is_return_results_every_t=True, is_checking_after_callback_after_t=False)
    blob = dict(meta_json=np.array(json.dumps(dict(ckpt=os.path.basename(dst), config={k: v for k, v in cfg.items() if isinstance(v, (int, str))},
                                                   missing=list(missing), unexpected=list(unexpected), keys=list(sd.keys()),
                                                   T=20, lr=0.1, data_seed=11, target_p=0.13, batch=32))),
                data=data, loss=np.array(res["loss"]), energy=np.array(res["energy"]), overall=np.array(res["overall"]))
    for l, x in enumerate(tr.get_model_xs()):
        blob[f"x_final_l{l}"] = x.detach().numpy().copy()
    path = os.path.join(GOLDEN, f"g11_checkpoint_{CKPT}.npz")
    np.savez_compressed(path, **blob)
    return path


def gen_learning(pc, um, te):
    """The toy learning runs whose fixed points SURVEY.md section 4 lists as known answers, run on the REFERENCE with its own
    RNG (torch.manual_seed): per-call parameter trajectories.  figure_4.py:111-150 (start (mu, W0) = (1, 7), 3 epochs x 125
    batches of 256, mixing 150 + sampling 1, SGD-p 0.07 momentum 0.2) and figure_6.py:24-72 (start (-7, -5), 10 epochs x 25
    batches of 2048, K = 150 one-sample, Adam-p) for noise_var in {1, 2, 4}.  The GPU run cannot share the RNG; the tests
    compare both with the analytic fixed points, using the reference's own spread as the yardstick."""
    import torch
    import torch.nn as nn
    import torch.optim as optim
    blob = {}

    def toy():
        m = nn.Sequential(nn.Linear(1, 1), pc.PCLayer(sample_x_fn=um.sample_x_fn_normal), nn.Linear(1, 1, bias=False))
        m.train()
        return m

    # figure_4
    torch.manual_seed(30)
    mu, var, B = 1.0, 5.0, 256
    datas = [mu + np.sqrt(var) * torch.randn(B, 1) for _ in range(125)]
    cfg = {"mixing": 150, "sampling": 1, "optimizer_x_kwargs_mcpc": {"lr": 0.01}, "optimizer_p_fn_mcpc": optim.SGD,
           "optimizer_p_kwargs_mcpc": {"lr": 0.07, "momentum": 0.2}, "input_var": 1.0}
    model = toy()
    tr = te.get_mcpc_trainer(model, cfg, training=True)
    nn.init.constant_(model[0].bias, 1.0)
    nn.init.constant_(model[2].weight, 7.0)
    traj = [(1.0, 7.0)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3):
            for d in datas:
                tr.train_on_batch(inputs=torch.zeros(B, 1), loss_fn=um.fe_fn, loss_fn_kwargs={"_target": d, "_var": 1.0},
                                  callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": tr},
                                  is_sample_x_at_batch_start=False, is_log_progress=False, is_checking_after_callback_after_t=False)
                traj.append((model[0].bias[0].item(), model[2].weight[0, 0].item()))
    blob["fig4_traj_mu_w"] = np.array(traj)
    # figure_6
    for nv in (1.0, 2.0, 4.0):
        torch.manual_seed(30)
        B = 2048
        datas = [mu + np.sqrt(var) * torch.randn(B, 1) for _ in range(25)]
        cfg = {"K": 150, "optimizer_x_kwargs_mcpc": {"lr": float(np.clip(0.01 * nv / 2, 0.001, 0.05))},
               "optimizer_p_fn_mcpc": optim.Adam, "optimizer_p_kwargs_mcpc": {"lr": float(np.clip(0.3 / nv, 0.5, 3))}, "input_var": 1.0}
        model = toy()
        nn.init.constant_(model[0].bias, -7.0)
        nn.init.constant_(model[2].weight, -5.0)
        tr = te.get_mcpc_trainer_one_sample(model, cfg, training=True)
        traj = [(-7.0, -5.0)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(10):
                for d in datas:
                    tr.train_on_batch(inputs=torch.zeros(B, 1), loss_fn=um.fe_fn, loss_fn_kwargs={"_target": d, "_var": 1.0},
                                      callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": tr, "var": nv},
                                      is_sample_x_at_batch_start=True, is_log_progress=False, is_return_results_every_t=False,
                                      is_checking_after_callback_after_t=False)
                    traj.append((model[0].bias[0].item(), model[2].weight[0, 0].item()))
        blob[f"fig6_nv{nv:g}_traj_mu_w"] = np.array(traj)
    blob["meta_json"] = np.array(json.dumps(dict(data_mean=mu, data_var=var, torch_seed=30)))
    path = os.path.join(GOLDEN, "g12_learning_reference_trajectories.npz")
    np.savez_compressed(path, **blob)
    return path


REP_SEEDS = tuple(range(21, 33))      # 12 torch seeds: the spread of the reference's own Langevin means


def rep_config(um):
    import torch.optim as optim
    return dict(EVAL_SHAPE, activation_fn="relu", loss_fn=um.bernoulli_fn, input_var=None, T_pc=60, optimizer_x_fn_pc=optim.Adam,
                optimizer_x_kwargs_pc={"lr": 0.1}, mixing=100, sampling=400, optimizer_x_kwargs_mcpc={"lr": 0.03})


def gen_representations(pc, um, te):
    """`get_representations` of the reference (utils/model.py:71-163) on the g10 net (seeded weights, 48 binary images in batches
    of 16, x0 = 3 everywhere so that the MAP inference is deterministic):
      MAP          x_1 after 60 Adam steps per datapoint -- deterministic, pinned exactly;
      expectation  mean of x_1 over all 500 Langevin steps started from the MAP state, for 12 torch seeds (the reference's own
                   normal_): per-datapoint mean and spread over the seeds;
      full         n = 10: every 40th sample after the mixing phase, one seed (shape / label layout)."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    cfg = rep_config(um)
    W, b, data, labels = eval_inputs(10001, "bernoulli")
    model = um.get_model(cfg, False, sample_x_fn=um.sample_x_fn_cte)
    load_params(model, W, b)
    loader = DataLoader(TensorDataset(torch.from_numpy(data), torch.from_numpy(labels)), batch_size=16)
    pc_tr = te.get_pc_trainer(model, cfg, training=False, is_mcpc=True)
    mc_tr = te.get_mcpc_trainer(model, cfg, training=False)
    blob = {"meta_json": np.array(json.dumps(dict(seed=10001, shape=EVAL_SHAPE, T_pc=60, lr_pc=0.1, mixing=100, sampling=400, lr_mcpc=0.03,
                                                  batch_size=16, torch_seeds=list(REP_SEEDS), n_full=10))),
            "data": data, "labels": labels}
    for j, (w, v) in enumerate(zip(W, b)):
        blob[f"W{j}"], blob[f"b{j}"] = w, v
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = um.get_representations(model, cfg, [pc_tr], loader, rep_type="MAP", use_cuda=False)
        blob["map_reps"], blob["map_labels"] = ds.tensors[0].detach().numpy().copy(), ds.tensors[1].numpy().copy()
        exps = []
        for seed in REP_SEEDS:
            torch.manual_seed(seed)
            ds = um.get_representations(model, cfg, [pc_tr, mc_tr], loader, rep_type="expectation", use_cuda=False)
            exps.append(ds.tensors[0].detach().numpy().copy())
        blob["expectation_reps"] = np.array(exps)                       # [12, 48, 8]
        blob["expectation_labels"] = ds.tensors[1].numpy().copy()
        torch.manual_seed(REP_SEEDS[0])
        ds = um.get_representations(model, cfg, [pc_tr, mc_tr], loader, rep_type="full", use_cuda=False, n=10)
        blob["full_reps"], blob["full_labels"] = ds.tensors[0].detach().numpy().copy(), ds.tensors[1].numpy().copy()
    path = os.path.join(GOLDEN, "g14_representations.npz")
    np.savez_compressed(path, **blob)
    return path


def main():
    pc, um, te = import_reference_eval()
    if sys.argv[1:] == ["representations"]:
        made = [gen_representations(pc, um, te)]
    else:
        made = [gen_evaluators(pc, um, te, "bernoulli", 10001), gen_evaluators(pc, um, te, "gaussian", 10002),
                gen_checkpoint(pc, um, te), gen_learning(pc, um, te), gen_representations(pc, um, te)]
    for p in made:
        print("wrote", p, os.path.getsize(p) // 1024, "KiB")


if __name__ == "__main__":
    main()
