"""representations (MAP / expectation / full), ancestral sampler, masked-reconstruction
error, importance-sampled marginal likelihood -- on the HIP engine, small synthetic data."""
import warnings

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

from tests import parity_log

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


def test_two_models_of_one_architecture_share_an_engine_safely():
    """figure_4.py:483-484 and table_1.py run trainers of several same-shape models side by side (training=False).  Engines
    are cached per architecture, so which parameters an engine is bound to is tracked on the ENGINE: interleaving two
    persistent trainers A, B, A must give A the same result as running A alone (the noise-free MAP call is deterministic)."""
    import montecarlopredictivecoding_amd.utils.model as um
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_pc_trainer
    um_, cfg, model_a, loader, _, _ = _setup()
    torch.manual_seed(11)
    model_b = um.get_model(cfg, True)
    assert not torch.equal(model_a[9].weight, model_b[9].weight)
    tr_a = get_pc_trainer(model_a, cfg, is_mcpc=True, training=False)
    tr_b = get_pc_trainer(model_b, cfg, is_mcpc=True, training=False)
    data = (torch.rand(16, 64, generator=torch.Generator().manual_seed(1)) < 0.3).float().to(DEV)
    inputs = torch.zeros(16, 8, device=DEV)
    x0 = [torch.rand(16, n, generator=torch.Generator().manual_seed(2 + i)).to(DEV) for i, n in enumerate((8, 32, 32))]

    def call(tr, model):
        for layer, x in zip(tr.get_model_pc_layers(), x0):
            layer._sample_x_fn = lambda inp, _x=x: _x.clone()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = tr.train_on_batch(inputs=inputs, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": data, "_var": None},
                                  is_log_progress=False, is_return_results_every_t=True)
        return np.array(r["overall"]), [x.detach().clone() for x in tr.get_model_xs()]

    ov_a1, xs_a1 = call(tr_a, model_a)
    ov_b, _ = call(tr_b, model_b)
    ov_a2, xs_a2 = call(tr_a, model_a)          # A's weights did not change: the engine must still be re-bound to them
    assert not np.allclose(ov_a1, ov_b)
    assert np.array_equal(ov_a1, ov_a2)
    for p, q in zip(xs_a1, xs_a2):
        assert torch.equal(p, q)


def test_adam_state_carried_across_calls_runs_stepwise():
    """With Adam on x and neither is_sample_x_at_batch_start nor is_reset_optimizer_x_at_batch_start, the reference keeps
    optimizer_x (moments and step count) across calls (pc_trainer.py:742-752).  The fused kernel restarts Adam, so such a call
    must take the step-wise path with the trainer's persistent torch optimizer: two calls of T steps then equal one call of
    2T steps."""
    import montecarlopredictivecoding_amd.predictive_coding as pc
    um, cfg, model, loader, _, _ = _setup()
    data = (torch.rand(16, 64, generator=torch.Generator().manual_seed(1)) < 0.3).float().to(DEV)
    inputs = torch.zeros(16, 8, device=DEV)
    kw = dict(inputs=inputs, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": data, "_var": None}, is_log_progress=False,
              is_return_results_every_t=True)

    def trainer(T):
        return pc.PCTrainer(model, T=T, optimizer_x_fn=torch.optim.Adam, optimizer_x_kwargs={"lr": 0.05}, update_p_at="never",
                            plot_progress_at=[])

    x0 = [torch.rand(16, n, generator=torch.Generator().manual_seed(20 + i)).to(DEV) for i, n in enumerate((8, 32, 32))]

    def reset_x():
        for layer, x in zip(model.modules() if False else [m for m in model if isinstance(m, pc.PCLayer)], x0):
            layer._sample_x_fn = lambda inp, _x=x: _x.clone()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        reset_x()
        long = trainer(12)
        r_long = long.train_on_batch(**kw)
        x_long = [x.detach().clone() for x in long.get_model_xs()]
        reset_x()
        short = trainer(6)
        r1 = short.train_on_batch(**kw)
        assert short.last_call_mode == "fused"
        r2 = short.train_on_batch(is_sample_x_at_batch_start=False, is_reset_optimizer_x_at_batch_start=False, **kw)
        assert short.last_call_mode == "stepwise"
        x_short = [x.detach().clone() for x in short.get_model_xs()]
    # the fused first call leaves optimizer_x in the state 6 Adam steps produce; the step-wise continuation (torch's Adam on
    # the kernel's gradients) then reproduces steps 6..11 of the long call
    assert short.get_optimizer_x().state_dict()["state"][0]["step"].item() == 12
    np.testing.assert_allclose(r1["overall"] + r2["overall"], r_long["overall"], rtol=2e-5)
    for p, q in zip(x_short, x_long):
        np.testing.assert_allclose(p.cpu().numpy(), q.cpu().numpy(), rtol=0, atol=2e-4)


@pytest.mark.parametrize("loss", ["bernoulli", "gaussian"])
def test_get_mse_rec_matches_reference(loss):
    """Masked-reconstruction error (reference utils/training_evaluation.py:143-170) on the fixture of
    oracle/gen_golden_eval.py: seeded weights and data, x0 = sample_x_fn_cte, 60 Adam MAP steps per batch on the HIP engine;
    the reference's scalar and its read-out of the MAP state are the expected values."""
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_mse_rec
    from tests.test_evaluators_golden import load_eval_fixture
    z, meta, cfg, model, loader = load_eval_fixture(loss, device=DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mse = get_mse_rec(model, cfg, loader, True)
    with torch.no_grad():
        last = model[-1](model[-2](model[-3].get_x().detach())).cpu().numpy()       # read-out of the last batch's MAP state
    np.testing.assert_allclose(last, z["map_readout"][-last.shape[0]:], rtol=0, atol=2e-3)
    if loss == "gaussian":
        assert mse == pytest.approx(float(z["mse_rec"]), rel=1e-4)
    else:
        # thresholded logits: a pixel whose logit is within the trajectory tolerance of 0 may flip -- at most two of 48 x 32
        assert abs(mse - float(z["mse_rec"])) <= 2.0 / (48 * 32) + 1e-9


def test_shipped_checkpoint_map_energies_match_reference():
    """figure_2.py:184 / table_1.py:76: a shipped state_dict (models/mcpc_fid_3, copied byte for byte as a data fixture) is
    loaded with strict=False and drives the engine unchanged: 20 Adam MAP steps on 32 seeded images give the reference's
    loss / energy per step and its final latent state."""
    import json
    import os
    import montecarlopredictivecoding_amd.utils.model as um
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_pc_trainer
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(gold, "g11_checkpoint_mcpc_fid_3.npz"))
    meta = json.loads(str(z["meta_json"]))
    cfg = dict(meta["config"], loss_fn=um.bernoulli_fn, input_var=None, optimizer_x_fn_pc=torch.optim.Adam,
               optimizer_x_kwargs_pc={"lr": meta["lr"]})
    model = um.get_model(cfg, True, sample_x_fn=um.sample_x_fn_cte)
    sd = torch.load(os.path.join(gold, meta["ckpt"]), map_location=DEV)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and sorted(unexpected) == sorted(meta["unexpected"])
    tr = get_pc_trainer(model, cfg, training=False, is_mcpc=True)
    data = torch.from_numpy(z["data"]).to(DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = tr.train_on_batch(inputs=torch.zeros(meta["batch"], cfg["input_size"], device=DEV), loss_fn=um.bernoulli_fn,
                                loss_fn_kwargs={"_target": data, "_var": None}, is_log_progress=False,
                                is_return_results_every_t=True, is_checking_after_callback_after_t=False)
    assert tr.last_call_mode == "fused"
    group = "shipped checkpoint mcpc_fid_3, 20 Adam MAP steps vs reference"
    parity_log.close(group, "loss[t]", res["loss"], z["loss"], rtol=1e-6)
    parity_log.close(group, "energy[t]", res["energy"], z["energy"], rtol=1e-6)
    parity_log.close(group, "overall[t]", res["overall"], z["overall"], rtol=1e-6)
    for l, x in enumerate(tr.get_model_xs()):
        parity_log.close(group, "final states", x.detach().cpu().numpy(), z[f"x_final_l{l}"], rtol=0, atol=1e-5)


def test_long_trajectories_are_recorded_in_slices_through_a_device_ring():
    """figure_2 / figure_5 pull every step of all latents to the host (pc_trainer.py:440-445,772-774).
    Above `mcpc_record_chunk_bytes` the fused call runs as slices whose records go through a two-buffer device ring to pinned
    host memory on a side stream; energies, every recorded state, outputs and the final state must be bitwise those of the
    one-launch call, the Hebbian sums equal up to the grouping of their fp32 partial sums."""
    um, cfg, model, loader, pc_tr, _ = _setup()
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer
    cfg = dict(cfg, mixing=20, sampling=37)
    data = (torch.rand(16, 64, generator=torch.Generator().manual_seed(1)) < 0.3).float().to(DEV)
    inputs = torch.zeros(16, 8, device=DEV)
    x0 = [torch.rand(16, n, generator=torch.Generator().manual_seed(70 + i)).to(DEV) for i, n in enumerate((8, 32, 32))]
    import montecarlopredictivecoding_amd.predictive_coding.pc_trainer as pt
    outs = []
    for chunk in (1 << 30, 16 * (8 + 32 + 32) * 4 * 10):          # one launch / slices of 5 steps (ring halves of 5)
        for layer, x in zip([m for m in model if hasattr(m, "get_x")], x0):
            layer._sample_x_fn = lambda inp, _x=x: _x.clone()
        tr = get_mcpc_trainer(model, cfg, training=False)
        tr.mcpc_materialize_unused_grads = True
        tr.mcpc_record_chunk_bytes = chunk
        base = pt._PHILOX_STEPS[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = tr.train_on_batch(inputs=inputs, loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": data, "_var": None},
                                  callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": tr},
                                  is_log_progress=False, is_return_results_every_t=True, is_return_xs=True, is_return_outputs=True)
        pt._PHILOX_STEPS[0] = base                                   # the second run replays the same noise
        outs.append((r, [x.detach().clone() for x in tr.get_model_xs()], tr.last_record_slices,
                     [m.weight.grad.clone() for m in model if isinstance(m, torch.nn.Linear)]))
        for m in model:
            if isinstance(m, torch.nn.Linear):
                m.weight.grad = None; m.bias.grad = None
    (ra, xa, sa, ga), (rb, xb, sb, gb) = outs
    assert sa == 0 and sb == 12                                      # 57 steps in slices of 5
    assert ra["overall"] == rb["overall"] and ra["loss"] == rb["loss"]
    assert len(rb["xs"]) == 57 and all(not t.is_cuda for t in rb["xs"][3])
    for t in range(57):
        for p, q in zip(ra["xs"][t], rb["xs"][t]):
            assert torch.equal(p, q)
        assert torch.equal(ra["outputs"][t], rb["outputs"][t])
    for p, q in zip(xa, xb):
        assert torch.equal(p, q)
    for p, q in zip(ga, gb):                                          # Hebbian sums: one flush per slice regroups the fp32 sums
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-5 * float(p.abs().max()))


def test_outputs_of_a_model_without_read_out_stay_on_the_device_when_sliced():
    """with no read-out Linear `outputs` is the last latent layer (pc_trainer.py:733: the Sequential's output); the
    sliced recorder drains latent records to pinned HOST memory, but `results["outputs"]` are live device tensors in the
    reference and on the unsliced path -- their device must not depend on mcpc_record_chunk_bytes."""
    import montecarlopredictivecoding_amd.predictive_coding as pc
    import montecarlopredictivecoding_amd.predictive_coding.pc_trainer as pt
    import montecarlopredictivecoding_amd.utils.model as um
    torch.manual_seed(2)
    model = torch.nn.Sequential(torch.nn.Linear(4, 4), pc.PCLayer(), torch.nn.Tanh(), torch.nn.Linear(4, 12),
                                pc.PCLayer(energy_fn=lambda inputs: 2.0 * 0.5 * (inputs["mu"] - inputs["x"]) ** 2)).to(DEV)
    model.train()
    outs = []
    for chunk in (1 << 30, 8 * 16 * 4 * 6):
        tr = pc.PCTrainer(model, T=30, optimizer_x_fn=torch.optim.SGD, optimizer_x_kwargs={"lr": 0.05}, update_p_at="never", plot_progress_at=[])
        tr.mcpc_record_chunk_bytes = chunk
        tr.mcpc_seed = 5                                             # (the Philox key follows torch.initial_seed() at construction)
        base = pt._PHILOX_STEPS[0]
        torch.manual_seed(9)                                         # same x0 draw in both runs
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = tr.train_on_batch(inputs=torch.zeros(8, 4, device=DEV), callback_after_t=um.random_step,
                                  callback_after_t_kwargs={"_pc_trainer": tr}, is_log_progress=False, is_return_results_every_t=True,
                                  is_return_outputs=True, is_return_xs=True)
        pt._PHILOX_STEPS[0] = base
        assert tr.last_call_mode == "fused"
        outs.append((r, tr.last_record_slices))
    (ra, sa), (rb, sb) = outs
    assert sa == 0 and sb > 1
    assert len(ra["outputs"]) == len(rb["outputs"]) == 30
    for p, q in zip(ra["outputs"], rb["outputs"]):
        assert p.is_cuda and q.is_cuda and torch.equal(p, q)
    assert not rb["xs"][0][0].is_cuda and torch.equal(rb["xs"][5][1].to(DEV), rb["outputs"][5])
