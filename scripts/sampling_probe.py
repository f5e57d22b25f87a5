"""Developer probe behind tests/test_gpu_sampling.py: the fused sampler's stationary statistics for several Philox seeds against the
g13 fixtures -- largest deviations (in units of the reference's seed spread) with their indices and absolute sizes."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlopredictivecoding_amd import _lib as L          # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine      # noqa: E402
from oracle.cases import make_case_inputs                     # noqa: E402

DEV = torch.device("cuda", 0)
for name in ("tanh_gaussian", "relu_bernoulli"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"g13_sampling_moments_{name}.npz"))
    case = json.loads(str(g["case_json"]))
    burn, T, lr, nvar = int(g["burn"]), int(g["T"]), float(g["lr"]), float(g["noise_var"])
    W, b, X0, inputs, target = make_case_inputs(case)
    sizes, B = case["sizes"], case["B"]
    act = {"tanh": L.ACT_TANH, "relu": L.ACT_RELU}[case["acts"][0]]
    eng = Engine(sizes, [act] * 3, case["n_in"], case["n_out"], B, device=DEV)
    eng.bind_params([torch.from_numpy(w).to(DEV) for w in W], [torch.from_numpy(v).to(DEV) for v in b])
    eng.bind_inputs(None)
    eng.bind_target(torch.from_numpy(target).to(DEV))
    kind = L.LOSS_GAUSSIAN if case["loss"] == "gaussian" else L.LOSS_BERNOULLI
    n = sum(sizes)
    iu = np.triu_indices(n, k=1)
    refc = np.array([c[iu] for c in g["cov"]])
    nseed = refc.shape[0]
    zs = []
    for seed in (20260104, 1, 2, 3, 4, 5):
        eng.load_state([torch.from_numpy(x).to(DEV) for x in X0])
        res = eng.run(T, loss_kind=kind, loss_var=case["var"], xopt=L.XOPT_SGD, lr=lr, noise_mode=L.NOISE_PHILOX, noise_var=nvar,
                      seed=seed, step_base=0, energy_mode=L.ENERGY_ALL, rec_begin=burn, rec_stride=1, rec_count=T - burn, rec_x=True)
        eng.sync_check()
        x = torch.cat(res.rec_x, dim=2).double().reshape(-1, n)
        mean = x.mean(0)
        cov = (x.T @ x / x.shape[0] - torch.outer(mean, mean)).cpu().numpy()
        z = (cov[iu] - refc.mean(0)) / (refc.std(0, ddof=1) * np.sqrt(1 + 1 / nseed))
        zs.append(z)
        top = np.argsort(-np.abs(z))[:4]
        print(name, "seed", seed, "mean z^2 %.3f" % (z * z).mean(), "top:",
              [(int(iu[0][k]), int(iu[1][k]), round(float(z[k]), 2), "%.4f vs %.4f +- %.4f" % (cov[iu][k], refc.mean(0)[k], refc.std(0, ddof=1)[k])) for k in top], flush=True)
    zs = np.array(zs)
    zm = zs.mean(0)
    top = np.argsort(-np.abs(zm))[:6]
    print(name, "mean z over GPU seeds, top:", [(int(iu[0][k]), int(iu[1][k]), round(float(zm[k]), 2), round(float(zs[:, k].std(ddof=1)), 2)) for k in top])
    # leave-one-out on the reference itself: each seed against the other 11
    loo = []
    for s in range(nseed):
        rest = np.delete(refc, s, axis=0)
        loo.append(np.abs((refc[s] - rest.mean(0)) / (rest.std(0, ddof=1) * np.sqrt(1 + 1 / (nseed - 1)))).max())
    print(name, "reference leave-one-out max |z| per seed:", np.round(loo, 2))
    eng.close()
