"""representations (MAP / expectation / full), ancestral sampler, masked-reconstruction
error, importance-sampled marginal likelihood -- on the HIP engine, small synthetic data."""
import warnings

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(loss="bernoulli"):
    import montecarlopredictivecoding_amd.utils.model as um
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer, get_pc_trainer
    torch.manual_seed(5)
    cfg = dict(input_size=8, hidden_size=32, hidden2_size=32, output_size=64, activation_fn="relu",
               loss_fn=um.bernoulli_fn if loss == "bernoulli" else um.fe_fn, input_var=0.3,
               T_pc=60, optimizer_x_fn_pc=torch.optim.Adam, optimizer_x_kwargs_pc={"lr": 0.1},
               mixing=20, sampling=40, optimizer_x_kwargs_mcpc={"lr": 0.03},
               optimizer_p_fn_mcpc=torch.optim.Adam, optimizer_p_kwargs_mcpc={"lr": 0.01})
    model = um.get_model(cfg, True)
    data = (torch.rand(48, 64) < 0.3).float()
    labels = torch.arange(48) % 10
    loader = DataLoader(TensorDataset(data, labels), batch_size=16)
    return um, cfg, model, loader, get_pc_trainer(model, cfg, is_mcpc=True, training=False), get_mcpc_trainer(model, cfg, training=False)


def test_get_representations_modes():
    um, cfg, model, loader, pc_tr, mc_tr = _setup()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds_map = um.get_representations(model, cfg, [pc_tr], loader, rep_type="MAP", use_cuda=True)
        ds_exp = um.get_representations(model, cfg, [pc_tr, mc_tr], loader, rep_type="expectation", use_cuda=True)
        ds_full = um.get_representations(model, cfg, [pc_tr, mc_tr], loader, rep_type="full", use_cuda=True, n=10)
    assert tuple(ds_map.tensors[0].shape) == (48, 8) and tuple(ds_exp.tensors[0].shape) == (48, 8)
    assert tuple(ds_full.tensors[0].shape) == (48 * 10, 8) and tuple(ds_full.tensors[1].shape) == (480,)
    assert pc_tr.last_call_mode == "fused" and mc_tr.last_call_mode == "fused"
    # the Langevin mean stays near the MAP estimate it was started from (posterior mass around the mode)
    d = (ds_exp.tensors[0] - ds_map.tensors[0]).abs().mean().item()
    assert np.isfinite(d) and d < 10.0
    assert torch.equal(ds_map.tensors[1].cpu(), torch.arange(48) % 10)


def test_sample_pc_and_marginal_likelihood():
    um, cfg, model, loader, pc_tr, mc_tr = _setup()
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_marginal_likelihood, sample_pc
    s = sample_pc(256, model, cfg, use_cuda=True)
    assert tuple(s.shape) == (256, 64) and set(s.unique().tolist()) <= {0.0, 1.0}
    h = sample_pc(256, model, cfg, use_cuda=True, is_return_hidden=True)
    assert tuple(h.shape) == (256, 64) and h.dtype == torch.float32
    ml = get_marginal_likelihood(model, cfg, loader, True, n_samples=512)
    # between the trivial bounds: log p(x) <= 0 and >= -64*log(2)-ish for an untrained net on 30%-dense bits
    assert -200.0 < ml < 0.0


def test_get_mse_rec_masked_inference():
    um, cfg, model, loader, pc_tr, mc_tr = _setup()
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_mse_rec
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mse = get_mse_rec(model, cfg, loader, True)
    assert 0.0 <= mse <= 1.0


def test_state_dict_layout_matches_reference_checkpoints():
    """Reference checkpoints hold '0.weight','0.bias','1._x','3.weight',... (the stale latents leak in because _x is a
    Parameter) and are loaded with strict=False (figure_2.py:184)."""
    um, cfg, model, loader, pc_tr, mc_tr = _setup()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pc_tr.train_on_batch(inputs=torch.zeros(16, 8, device=DEV), loss_fn=um.bernoulli_fn,
                             loss_fn_kwargs={"_target": torch.zeros(16, 64, device=DEV), "_var": None}, is_log_progress=False)
    keys = list(model.state_dict().keys())
    assert keys == ["0.weight", "0.bias", "1._x", "3.weight", "3.bias", "4._x", "6.weight", "6.bias", "7._x", "9.weight", "9.bias"]
    fresh = um.get_model(cfg, True)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    missing, unexpected = fresh.load_state_dict(sd, strict=False)
    assert sorted(unexpected) == ["1._x", "4._x", "7._x"] and not missing       # stale latents are ignored, weights load
    assert torch.equal(fresh[9].weight, model[9].weight)
