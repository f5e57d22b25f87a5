"""SURVEY.md section 8f rows 3 and 4 pinned to the reference: the ancestral sampler, the marginal-likelihood core and the
checkpoint layout, against fixtures written by oracle/gen_golden_eval.py from the IMPORTED reference
(/root/reference/utils/training_evaluation.py:72-100,143-206; figure_2.py:184 / table_1.py:76 for the checkpoint).

These parts of the package are plain torch (no Langevin loop), so they are checked on the CPU here, every round; the parts
that run the HIP engine (get_mse_rec's MAP inference, MAP energies from the checkpoint) are in tests/test_gpu_widening.py.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

import montecarlopredictivecoding_amd.utils.model as um
from montecarlopredictivecoding_amd.utils.training_evaluation import marginal_likelihood_from_logits, sample_pc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_eval_fixture(loss, device="cpu"):
    z = np.load(os.path.join(GOLDEN, f"g10_eval_{loss}.npz"))
    meta = json.loads(str(z["meta_json"]))
    cfg = dict(meta["shape"], activation_fn="relu", loss_fn=um.bernoulli_fn if loss == "bernoulli" else um.fe_fn,
               input_var=meta["input_var"], T_pc=meta["T_pc"], optimizer_x_fn_pc=torch.optim.Adam,
               optimizer_x_kwargs_pc={"lr": meta["lr"]})
    model = um.get_model(cfg, device != "cpu", sample_x_fn=um.sample_x_fn_cte)
    lins = [m for m in model if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for j, lin in enumerate(lins):
            lin.weight.copy_(torch.from_numpy(z[f"W{j}"]))
            lin.bias.copy_(torch.from_numpy(z[f"b{j}"]))
    loader = DataLoader(TensorDataset(torch.from_numpy(z["data"]), torch.from_numpy(z["labels"])), batch_size=meta["batch_size"])
    return z, meta, cfg, model, loader


@pytest.mark.parametrize("loss", ["bernoulli", "gaussian"])
def test_sample_pc_consumes_the_reference_rng_stream(loss):
    z, meta, cfg, model, loader = load_eval_fixture(loss)
    torch.manual_seed(meta["torch_seed"])
    h = sample_pc(meta["n_hidden_samples"], model, cfg, use_cuda=False, is_return_hidden=True)
    np.testing.assert_allclose(h.numpy(), z["hidden_samples"], rtol=0, atol=1e-6)


def test_marginal_likelihood_matches_reference():
    z, meta, cfg, model, loader = load_eval_fixture("bernoulli")
    # the likelihood core on the reference's own prior samples ...
    ml = marginal_likelihood_from_logits(torch.from_numpy(z["ml_logits"]), loader)
    assert ml == pytest.approx(float(z["ml"]), rel=2e-6)
    # ... and end to end: same seed -> same prior samples -> same estimate
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_marginal_likelihood
    torch.manual_seed(meta["torch_seed"])
    assert get_marginal_likelihood(model, cfg, loader, False, n_samples=meta["n_ml_samples"]) == pytest.approx(float(z["ml"]), rel=2e-6)
    torch.manual_seed(meta["torch_seed"])
    s = sample_pc(meta["n_hidden_samples"], model, cfg, use_cuda=False)
    assert s.dtype == torch.float64 and float(s.mean()) == pytest.approx(float(z["bernoulli_sample_mean"]), abs=1e-12)


def test_shipped_checkpoint_loads_like_the_reference():
    """The byte-for-byte copy of /root/reference/models/mcpc_fid_3 loads into this package's get_model with strict=False and
    reports the same missing / unexpected keys the reference's own model reported (stale `N._x` latents are ignored)."""
    z = np.load(os.path.join(GOLDEN, "g11_checkpoint_mcpc_fid_3.npz"))
    meta = json.loads(str(z["meta_json"]))
    cfg = dict(meta["config"], loss_fn=um.bernoulli_fn)
    model = um.get_model(cfg, False)
    sd = torch.load(os.path.join(GOLDEN, meta["ckpt"]), map_location="cpu")
    assert list(sd.keys()) == meta["keys"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert list(missing) == meta["missing"] and list(unexpected) == meta["unexpected"]
    assert torch.equal(model[9].weight, sd["9.weight"]) and torch.equal(model[0].bias, sd["0.bias"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="build container only: needs the reference's models/")
def test_every_shipped_pc_checkpoint_loads():
    """Every pc_* / mcpc_* weight file of the reference (table_1.py:132-212 architectures) loads into get_model with the
    sizes its tensors imply; the dlgm_* files belong to the out-of-scope DLGM baseline and only have to unpickle."""
    n = 0
    for path in sorted(glob.glob("/root/reference/models/*")):
        if os.path.isdir(path):
            continue
        sd = torch.load(path, map_location="cpu")
        if os.path.basename(path).startswith("dlgm"):
            assert len(sd) > 0
            continue
        cfg = dict(input_size=sd["0.weight"].shape[0], hidden_size=sd["3.weight"].shape[0], hidden2_size=sd["6.weight"].shape[0],
                   output_size=sd["9.weight"].shape[0], activation_fn="relu")
        model = um.get_model(cfg, False)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not missing and all(k.endswith("._x") for k in unexpected)
        n += 1
    assert n == 18
