"""Generate tests/golden/*.npz by running the IMPORTED reference.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs /root/reference); the fixtures it writes are
plain data (inputs + expected outputs) and are what travels to the GPU box.

    python -m oracle.gen_golden            # regenerate everything under tests/golden/

How the reference is driven (every reference instruction on the path stays untouched):
  * ``seaborn`` is not installed; it is only used by the plotting branch
    (/root/reference/predictive_coding/pc_trainer.py:1020-1053), so an empty module is
    registered under that name before the import.
  * the initial latent state is pinned through the API-legal ``sample_x_fn`` hook
    (pc_layer.py:223-230): ``sample_x_fn = lambda inp: X0[l]``.
  * the Langevin noise is pinned through a ``callback_after_t`` that performs exactly
    ``random_step``'s arithmetic (/root/reference/utils/model.py:39-44) with the
    ``normal_`` draw replaced by a copy of pre-generated normals:
        x.grad.copy_(-sqrt(var/lr) * XI[t][l]);  optimizer.step()
    (the minus sign turns the reference's ``x -= lr*grad`` into the build's convention
    ``x += sqrt(var*lr)*xi``).
  * weights, x0, targets and the normals come from ``oracle/philox.py`` (seeded).
"""
import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"


def import_reference():
    import torch  # noqa: F401
    if "seaborn" not in sys.modules:
        sys.modules["seaborn"] = types.ModuleType("seaborn")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import predictive_coding as pc          # the reference library
    assert pc.__file__.startswith(REF), pc.__file__
    import utils.model as um                # the reference's loss fns / random_step
    assert um.__file__.startswith(REF), um.__file__
    return pc, um


from oracle import philox  # noqa: E402
from oracle.cases import make_case_inputs  # noqa: E402


def build_reference_model(pc, case, W, b, X0, device="cpu"):
    import torch
    import torch.nn as nn
    acts = {"identity": None, "relu": nn.ReLU, "tanh": nn.Tanh}
    sizes = case["sizes"]
    L = len(sizes)
    n_in = case["n_in"]
    mods = []
    dims = [n_in] + list(sizes) + ([case["n_out"]] if case["n_out"] else [])
    for l in range(L):
        lin = nn.Linear(dims[l], dims[l + 1], bias=b[l] is not None)
        mods.append(lin)
        c = case["ecoef"][l]
        x0 = torch.from_numpy(X0[l].copy()).to(device)
        kw = dict(sample_x_fn=(lambda inp, x0=x0: x0.clone()))
        if c != 1.0:
            kw["energy_fn"] = (lambda inputs, c=c: c * 0.5 * (inputs["mu"] - inputs["x"]) ** 2)
        mods.append(pc.PCLayer(**kw))
        a = acts[case["acts"][l]]
        if a is not None:
            mods.append(a())
    if case["n_out"]:
        mods.append(nn.Linear(dims[L], dims[L + 1], bias=b[L] is not None))
    model = nn.Sequential(*mods)
    lins = [m for m in model if isinstance(m, nn.Linear)]
    with torch.no_grad():
        for j, lin in enumerate(lins):
            lin.weight.copy_(torch.from_numpy(W[j]))
            if b[j] is not None:
                lin.bias.copy_(torch.from_numpy(b[j]))
    model.train()
    model.to(device)
    return model, lins


def reference_loss(um, case, target, device="cpu"):
    import torch
    kind = case["loss"]
    if kind == "none":
        return None, {}
    if kind == "zero":
        return um.zero_fn, {}
    t = torch.from_numpy(target).to(device)
    if kind == "gaussian":
        return um.fe_fn, {"_target": t, "_var": case["var"]}
    if kind == "bernoulli":
        return um.bernoulli_fn, {"_target": t, "_var": None}
    if kind == "gaussian_mask":
        return (lambda o, _target, _var: um.fe_fn_mask(o, _target, _var, perc=case["perc"])), \
            {"_target": t, "_var": case["var"]}
    if kind == "bernoulli_mask":
        return (lambda o, _target, _var=None: um.bernoulli_fn_mask(o, _target, _var, perc=case["perc"])), \
            {"_target": t, "_var": None}
    raise ValueError(kind)


def run_reference_call(pc, um, model, call, inputs, target, XI, case, device="cpu"):
    """One ``train_on_batch`` on the reference.  Returns dict of numpy results."""
    import torch
    import torch.optim as optim
    T = call["T"]
    xfn = optim.SGD if call["xopt"] == "sgd" else optim.Adam
    p_fn = {"sgd": optim.SGD, "adam": optim.Adam}[call.get("popt", "sgd")]
    xkw = {"lr": call["lr"]}
    xkw.update(call.get("xopt_extra", {}))            # e.g. SGD momentum: not fusable -> step-wise path
    trainer = pc.PCTrainer(
        model, T=T, update_x_at=call.get("update_x_at", "all"), optimizer_x_fn=xfn, optimizer_x_kwargs=xkw,
        x_lr_discount=call.get("x_lr_discount", 1.0), x_lr_amplifier=call.get("x_lr_amplifier", 1.0),
        update_p_at=call.get("update_p_at", "never"),
        accumulate_p_at=call.get("accumulate_p_at", "never"),
        optimizer_p_fn=p_fn, optimizer_p_kwargs=call.get("popt_kwargs", {"lr": 0.0}),
        plot_progress_at=[],
    )
    loss_fn, loss_kwargs = reference_loss(um, case, target, device)
    kw = {}
    if call.get("noise", False):
        var = call.get("noise_var", 2.0)

        def injected_random_step(t, _pc_trainer):
            xs = _pc_trainer.get_model_xs()
            optimizer = _pc_trainer.get_optimizer_x()
            std = np.sqrt(var / optimizer.defaults["lr"])
            for l, x in enumerate(xs):
                x.grad.copy_(torch.from_numpy(XI[t][l]).to(device) * (-std))
            optimizer.step()
        kw = dict(callback_after_t=injected_random_step, callback_after_t_kwargs={"_pc_trainer": trainer})
    if "clip_x_grad" in call:       # a user callback between backward and the x step (pc_trainer.py:865-866)
        clip = call["clip_x_grad"]

        def clip_grads(t, _pc_trainer):
            for x in _pc_trainer.get_model_xs():
                x.grad.clamp_(-clip, clip)
        kw.update(callback_after_backward=clip_grads, callback_after_backward_kwargs={"_pc_trainer": trainer})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = trainer.train_on_batch(
            inputs=torch.from_numpy(inputs).to(device), loss_fn=loss_fn, loss_fn_kwargs=loss_kwargs,
            is_sample_x_at_batch_start=call.get("sample_x", True),
            is_log_progress=False, is_return_results_every_t=True,
            is_checking_after_callback_after_t=False,
            is_return_outputs=True, is_return_xs=True, **kw)
    out = {
        "energy": np.array(res["energy"], dtype=np.float64),
        "overall": np.array(res["overall"], dtype=np.float64),
        "loss": np.array(res["loss"], dtype=np.float64) if len(res["loss"]) else np.zeros(T),
    }
    rec = call.get("record_at", [])
    nc = case.get("rec_chains", None)      # big nets: keep the first few chains only
    for t in rec:
        for l, x in enumerate(res["xs"][t]):
            out[f"x_t{t}_l{l}"] = x.cpu().numpy()[:nc].copy()
        out[f"out_t{t}"] = res["outputs"][t].detach().cpu().numpy()[:nc].copy()
    for l, x in enumerate(trainer.get_model_xs()):
        out[f"x_final_l{l}"] = x.detach().cpu().numpy()[:nc].copy()
    return out, trainer


def param_grads(lins):
    g = {}
    for j, lin in enumerate(lins):
        if lin.weight.grad is not None:
            g[f"gW{j}"] = lin.weight.grad.cpu().numpy().copy()
        if lin.bias is not None and lin.bias.grad is not None:
            g[f"gb{j}"] = lin.bias.grad.cpu().numpy().copy()
    return g


def gen_case(pc, um, name, case, store_inputs=True, grad_samples=None):
    """case: dict(sizes, acts, ecoef, n_in, n_out, loss, var, perc, B, seed, inputs_zero, calls=[...])."""
    import torch
    torch.manual_seed(0)
    sizes, B, seed = case["sizes"], case["B"], case["seed"]
    W, b, X0, inputs, target = make_case_inputs(case)
    model, lins = build_reference_model(pc, case, W, b, X0)
    blob = {}
    t_base = 0
    for ci, call in enumerate(case["calls"]):
        T = call["T"]
        XI = None
        if call.get("noise", False):
            XI = [[philox.layer_normals(seed, t_base + t, l, 0, B, n) for l, n in enumerate(sizes)]
                  for t in range(T)]
        out, trainer = run_reference_call(pc, um, model, call, inputs, target, XI, case)
        for k, v in out.items():
            blob[f"c{ci}_{k}"] = v
        g = param_grads(lins)
        if grad_samples is None:
            for k, v in g.items():
                blob[f"c{ci}_{k}"] = v
        else:   # big nets: keep a checksum + sampled entries only
            rs = np.random.RandomState(grad_samples)
            for k, v in g.items():
                flat = v.reshape(-1)
                idx = rs.randint(0, flat.size, size=min(64, flat.size))
                blob[f"c{ci}_{k}_idx"] = idx
                blob[f"c{ci}_{k}_val"] = flat[idx].copy()
                blob[f"c{ci}_{k}_sum"] = np.float64(flat.astype(np.float64).sum())
                blob[f"c{ci}_{k}_abs"] = np.float64(np.abs(flat.astype(np.float64)).sum())
        # parameters after the call (a p-step may have changed them)
        for j, lin in enumerate(lins):
            if call.get("update_p_at", "never") != "never":
                blob[f"c{ci}_W{j}_after"] = lin.weight.detach().numpy().copy()
                if lin.bias is not None:
                    blob[f"c{ci}_b{j}_after"] = lin.bias.detach().numpy().copy()
        t_base += T
    blob["case_json"] = np.array(json.dumps(case))
    if store_inputs:
        for j in range(len(W)):
            blob[f"W{j}"] = W[j]
            if b[j] is not None:
                blob[f"b{j}"] = b[j]
        for l in range(len(sizes)):
            blob[f"X0_{l}"] = X0[l]
        blob["inputs"] = inputs
        if target is not None:
            blob["target"] = target
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **blob)
    return path


def all_cases():
    cases = {}
    # G1: tiny 6-16-16-24 net, every loss x activation x optimiser mode
    rec = [0, 1, 2, 5, 10, 25, 49]
    for act in ("relu", "tanh"):
        for loss in ("bernoulli", "gaussian", "gaussian_mask", "bernoulli_mask", "zero"):
            for mode in ("sgdnoise", "sgd", "adam"):
                call = dict(T=50, xopt="adam" if mode == "adam" else "sgd",
                            lr=0.05 if mode == "adam" else 0.03, noise=(mode == "sgdnoise"),
                            record_at=rec)
                cases[f"g1_{act}_{loss}_{mode}"] = dict(
                    sizes=[6, 16, 16], acts=[act] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24,
                    loss=loss, var=0.3, perc=0.5, B=8, seed=1000 + len(cases), x0_range=2.0,
                    calls=[call])
    # G3: accumulate semantics (mixing 3, sampling 4), p-step with a real optimizer
    cases["g3_accumulate_last"] = dict(
        sizes=[6, 16, 16], acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24,
        loss="bernoulli", var=1.0, perc=0.5, B=8, seed=3001, x0_range=2.0,
        calls=[dict(T=7, xopt="sgd", lr=0.03, noise=True, update_p_at="last",
                    accumulate_p_at=[3, 4, 5, 6], popt="adam", popt_kwargs={"lr": 0.01},
                    record_at=[0, 6])])
    cases["g3_accumulate_never"] = dict(
        sizes=[6, 16, 16], acts=["tanh"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24,
        loss="gaussian", var=0.3, perc=0.5, B=8, seed=3002, x0_range=2.0,
        calls=[dict(T=7, xopt="sgd", lr=0.03, noise=True, update_p_at="never",
                    accumulate_p_at=[3, 4, 5, 6], record_at=[0, 6])])
    cases["g3_pc_last_noacc"] = dict(
        sizes=[6, 16, 16], acts=["tanh"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24,
        loss="gaussian", var=0.3, perc=0.5, B=8, seed=3003, x0_range=2.0,
        calls=[dict(T=9, xopt="adam", lr=0.05, noise=False, update_p_at="last",
                    popt="sgd", popt_kwargs={"lr": 0.07, "momentum": 0.2}, record_at=[0, 8])])
    # G4: linear-Gaussian toy net of figure_2.py:40-64 (b0=0.2, W=2, y=1, var=1, x0=3)
    cases["g4_linear_gaussian"] = dict(
        sizes=[1], acts=["identity"], ecoef=[1.0], n_in=1, n_out=1, no_bias=[1],
        loss="gaussian", var=1.0, perc=0.5, B=4, seed=4001,
        const_params=[(None, 0.2), (2.0, None)],
        calls=[dict(T=30, xopt="adam", lr=0.02, noise=False, record_at=[0, 29]),
               dict(T=40, xopt="sgd", lr=0.02, noise=True, sample_x=False, record_at=[0, 39])])
    # G5: sensory PCLayer with a scaled energy (figure_3.py:47-55), free-running generation
    cases["g5_sensory_linear"] = dict(
        sizes=[1, 1], acts=["identity", "identity"], ecoef=[1.0, 1.0 / 0.5], n_in=1, n_out=0,
        loss="none", var=1.0, perc=0.5, B=4, seed=5001, const_params=[(None, 0.5), (2.0, None)],
        no_bias=[1],
        calls=[dict(T=40, xopt="sgd", lr=0.3 * 0.3, noise=True, record_at=[0, 39])])
    cases["g5_sensory_wide"] = dict(
        sizes=[4, 8, 12], acts=["tanh", "relu", "identity"], ecoef=[1.0, 1.0, 2.5], n_in=4, n_out=0,
        loss="none", var=1.0, perc=0.5, B=6, seed=5002,
        calls=[dict(T=40, xopt="sgd", lr=0.05, noise=True, record_at=[0, 39])])
    # G6: non-zero pseudo-inputs (W0 receives a gradient), learning step
    cases["g6_nonzero_inputs"] = dict(
        sizes=[5, 12], acts=["tanh", "tanh"], ecoef=[1.0, 1.0], n_in=7, n_out=9, inputs_zero=False,
        loss="gaussian", var=0.5, perc=0.5, B=10, seed=6001,
        calls=[dict(T=12, xopt="sgd", lr=0.05, noise=True, update_p_at="last",
                    accumulate_p_at=list(range(4, 12)), popt="sgd", popt_kwargs={"lr": 0.1},
                    record_at=[0, 11])])
    # G7: MAP warm-up (Adam) then MCPC continuing from it (figure_2.py:227-228 pattern)
    cases["g7_map_then_mcpc"] = dict(
        sizes=[6, 16, 16], acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24,
        loss="bernoulli", var=1.0, perc=0.5, B=8, seed=7001, x0_range=10.0,
        calls=[dict(T=20, xopt="adam", lr=0.1, noise=False, record_at=[0, 19]),
               dict(T=30, xopt="sgd", lr=0.03, noise=True, sample_x=False, record_at=[0, 29])])
    # ragged sizes (not multiples of 16 / 4), B not a multiple of the chain tile
    cases["g8_ragged"] = dict(
        sizes=[3, 17, 33], acts=["relu", "tanh", "relu"], ecoef=[1.0] * 3, n_in=3, n_out=21,
        loss="bernoulli", var=1.0, perc=0.5, B=37, seed=8001, x0_range=2.0,
        calls=[dict(T=25, xopt="sgd", lr=0.03, noise=True, update_p_at="last",
                    accumulate_p_at=list(range(10, 25)), popt="sgd", popt_kwargs={"lr": 0.0},
                    record_at=[0, 24])])
    # G9: generic API features that run on the step-wise HIP path
    base9 = dict(sizes=[6, 16, 16], acts=["tanh"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24, loss="gaussian", var=0.5,
                 perc=0.5, B=8, x0_range=2.0)
    cases["g9_update_p_all"] = dict(base9, seed=9001, calls=[dict(
        T=6, xopt="sgd", lr=0.03, noise=True, update_p_at="all", popt="sgd", popt_kwargs={"lr": 0.01}, record_at=[0, 5])])
    cases["g9_x_lr_discount"] = dict(base9, seed=9002, calls=[dict(
        T=15, xopt="sgd", lr=0.4, noise=False, x_lr_discount=0.7, record_at=[0, 14])])
    cases["g9_clip_after_backward"] = dict(base9, seed=9003, x0_range=6.0, calls=[dict(
        T=10, xopt="sgd", lr=0.05, noise=True, clip_x_grad=0.8, record_at=[0, 9])])
    cases["g9_sgd_momentum_x"] = dict(base9, seed=9004, calls=[dict(
        T=10, xopt="sgd", lr=0.03, noise=False, xopt_extra={"momentum": 0.5}, record_at=[0, 9])])
    cases["g9_update_x_last_half"] = dict(base9, seed=9005, calls=[dict(
        T=10, xopt="sgd", lr=0.05, noise=False, update_x_at="last_half", record_at=[0, 4, 5, 9])])
    return cases


def cfg_m_case():
    # G2: the BASELINE metric's network at full width, reduced batch / steps
    return dict(
        sizes=[30, 256, 256], acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=30, n_out=784,
        loss="bernoulli", var=1.0, perc=0.5, B=64, seed=30, x0_range=10.0, rec_chains=4,
        target_p=0.13,
        calls=[dict(T=100, xopt="sgd", lr=0.03, noise=True, update_p_at="never",
                    accumulate_p_at=list(range(20, 100)), record_at=[0, 9, 99])])


def main():
    pc, um = import_reference()
    made = []
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, case in all_cases().items():
        if only and not name.startswith(only):
            continue
        made.append(gen_case(pc, um, name, case))
    if only:
        print("wrote", made)
        return
    made.append(gen_case(pc, um, "g2_cfgM_b64", cfg_m_case(), store_inputs=False, grad_samples=7))
    total = sum(os.path.getsize(p) for p in made)
    print(f"wrote {len(made)} fixtures, {total/1024:.0f} KiB total -> {GOLDEN}")


if __name__ == "__main__":
    main()
