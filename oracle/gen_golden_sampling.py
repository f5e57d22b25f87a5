"""g13_sampling_moments_*: what the reference's OWN sampler -- its `random_step` with torch's `normal_`
(/root/reference/utils/model.py:35-44), driven by `PCTrainer.train_on_batch` (/root/reference/predictive_coding/pc_trainer.py:712-981)
-- produces in distribution on a non-linear, multi-unit net.  TEST INFRASTRUCTURE ONLY; runs only in the build container.

    python -m oracle.gen_golden_sampling [tanh_gaussian|relu_bernoulli|relu_zero]     # writes tests/golden/g13_sampling_moments_<net>.npz

Trajectory fixtures (g1..g9) pin the arithmetic through INJECTED normals; nothing in them would notice a fused generator
whose normals are correlated between units, layers, consecutive steps or neighbouring chains, or a mis-scaled kick.  Here the
reference runs with its own generator, for several torch seeds, and what is stored are stationary statistics of the latent
state over chains and time: the mean vector and the full covariance matrix of the concatenated latents (38 x 38: variances,
intra-layer and inter-layer covariances), and the time-averaged loss / energy / overall.  The spread over the seeds is the
yardstick: the GPU run (fused Philox4x32-10 + Box-Muller) has to sit inside it (tests/test_gpu_sampling.py).

net 6-16-16-24 (`get_model`'s shape), B = 4096 chains with their own targets, SGD-x lr 0.03, noise var 2, T = 2500:
statistics over x_t, t in [500, 2500) (x_t = the state BEFORE the update of step t, what the reference records, :768-797).

`relu_zero` (round 6) is BASELINE config 5's call on that net: UNCLAMPED generation -- `loss_fn = zero_fn`
(/root/reference/utils/model.py:31-33), the free-running sampler of /root/reference/figure_3.py:125-161, whose product is the READ-OUT
(`is_return_outputs`, figure_3.py:153-161: the generated images).  Beside the latents' moments it stores the mean vector and the full
24 x 24 covariance of the read-out `out = f(x_3) W_3^T + b_3` of the same states (FID cannot be evaluated offline: BASELINE.md section 4
promised pixel-moment statistics in its place).
"""
import json
import os
import warnings

import numpy as np

from oracle.cases import make_case_inputs
from oracle.gen_golden import GOLDEN, build_reference_model, import_reference, reference_loss

BURN, T, B, LR, VAR = 500, 2500, 4096, 0.03, 2.0
SEEDS = tuple(101 * k for k in range(1, 13))       # 12 torch seeds: the spread is the yardstick

CASES = {
    "tanh_gaussian": dict(sizes=[6, 16, 16], acts=["tanh"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24, loss="gaussian", var=0.3,
                          perc=0.5, B=B, seed=13001, x0_range=2.0, calls=[dict(T=T)]),
    "relu_bernoulli": dict(sizes=[6, 16, 16], acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24, loss="bernoulli", var=1.0,
                           perc=0.5, B=B, seed=13002, x0_range=2.0, calls=[dict(T=T)]),
    "relu_zero": dict(sizes=[6, 16, 16], acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24, loss="zero", var=1.0,
                      perc=0.5, B=B, seed=13003, x0_range=2.0, calls=[dict(T=T)]),
}


def run_reference(pc, um, case, torch_seed):
    import torch
    import torch.optim as optim
    W, b, X0, inputs, target = make_case_inputs(case)
    model, lins = build_reference_model(pc, case, W, b, X0)
    trainer = pc.PCTrainer(model, T=T, update_x_at="all", optimizer_x_fn=optim.SGD, optimizer_x_kwargs={"lr": LR},
                           update_p_at="never", optimizer_p_fn=optim.SGD, optimizer_p_kwargs={"lr": 0.0}, plot_progress_at=[])
    loss_fn, loss_kwargs = reference_loss(um, case, target)
    n = sum(case["sizes"])
    s1 = torch.zeros(n, dtype=torch.float64)
    s2 = torch.zeros(n, n, dtype=torch.float64)
    count = [0]
    with_out = case["loss"] == "zero"                             # generation: the read-out of the same states is the product
    no = case["n_out"]
    o1 = torch.zeros(no, dtype=torch.float64)
    o2 = torch.zeros(no, no, dtype=torch.float64)
    act = {"relu": torch.relu, "tanh": torch.tanh, "identity": (lambda v: v)}[case["acts"][-1]]

    def langevin_then_tally(t, _pc_trainer):
        um.random_step(t, _pc_trainer, var=VAR)                   # the reference's own callback, its own normal_
        if BURN <= t + 1 < T:                                     # the state now is x_{t+1}
            xs = [p.detach() for p in _pc_trainer.get_model_xs()]
            x = torch.cat(xs, dim=1).double()
            s1.add_(x.sum(0))
            s2.add_(x.T @ x)
            count[0] += x.shape[0]
            if with_out:                                          # what the model's forward returns for this state (its last Linear)
                with torch.no_grad():
                    out = lins[-1](act(xs[-1])).double()
                o1.add_(out.sum(0))
                o2.add_(out.T @ out)

    torch.manual_seed(torch_seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = trainer.train_on_batch(inputs=torch.from_numpy(inputs), loss_fn=loss_fn, loss_fn_kwargs=loss_kwargs,
                                     callback_after_t=langevin_then_tally, callback_after_t_kwargs={"_pc_trainer": trainer},
                                     is_log_progress=False, is_return_results_every_t=True, is_checking_after_callback_after_t=False)
    assert count[0] == (T - BURN) * B
    mean = (s1 / count[0]).numpy()
    cov = (s2 / count[0]).numpy() - np.outer(mean, mean)
    en = np.array([np.mean(res[k][BURN:T]) for k in ("loss", "energy", "overall")])
    if with_out:
        om = (o1 / count[0]).numpy()
        return mean, cov, en, om, (o2 / count[0]).numpy() - np.outer(om, om)
    return mean, cov, en, None, None


def main():
    import sys
    pc, um = import_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    for name, case in CASES.items():
        if sys.argv[1:] and name not in sys.argv[1:]:
            continue
        means, covs, ens, omeans, ocovs = [], [], [], [], []
        for seed in SEEDS:
            m, c, e, om, oc = run_reference(pc, um, case, seed)
            means.append(m); covs.append(c); ens.append(e)
            if om is not None:
                omeans.append(om); ocovs.append(oc)
            print(name, seed, "mean[:3]", m[:3], "var[:3]", np.diag(c)[:3], "energies", e, flush=True)
        path = os.path.join(GOLDEN, f"g13_sampling_moments_{name}.npz")
        extra = dict(out_mean=np.array(omeans), out_cov=np.array(ocovs)) if omeans else {}
        np.savez_compressed(path, case_json=np.array(json.dumps(case)), seeds=np.array(SEEDS), mean=np.array(means),
                            cov=np.array(covs), energies=np.array(ens), burn=BURN, T=T, lr=LR, noise_var=VAR, **extra)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
