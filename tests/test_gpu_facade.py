"""GPU parity of the PCLayer / PCTrainer facade.

The SAME harness that drove the imported reference when the fixtures were generated
(oracle/gen_golden.py: build_reference_model / run_reference_call) is pointed at this package's
``predictive_coding`` and ``utils.model`` modules instead, on cuda:0, and its outputs are compared
with the stored reference outputs.  Calls with injected noise use an arbitrary callback and therefore
exercise the step-wise HIP path; noise-free calls exercise the fused path.
"""
import warnings

import numpy as np
import pytest
import torch

from oracle import gen_golden, philox
from oracle.cases import make_case_inputs
from tests.golden_util import Golden, fixture_names, first_kink_step, relu_kink_events
from tests import parity_log

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(params=["default", "barrier"], autouse=True)
def step_kernel(request, monkeypatch):
    """Run every facade test on the default kernel (in-place, wave-specialised) and on the barrier kernel kept as fallback (ws=0);
    MCPC_TUNING is read by engine.py when an engine is created, and a plan that does not fit falls back by itself."""
    if request.param == "barrier":
        monkeypatch.setenv("MCPC_TUNING", "ws=0")
    return request.param


def _mods():
    import montecarlopredictivecoding_amd.predictive_coding as pc
    import montecarlopredictivecoding_amd.utils.model as um
    return pc, um


def _replay(name, small_only=True):
    pc, um = _mods()
    g = Golden(name)
    case = g.case
    W, b, X0, inputs, target = g.W, g.b, g.X0, g.inputs, g.target
    model, lins = gen_golden.build_reference_model(pc, case, W, b, X0, device=DEV)
    t_base = 0
    modes = []
    for ci, call in enumerate(case["calls"]):
        T = call["T"]
        XI = None
        if call.get("noise", False):
            XI = [[philox.layer_normals(case["seed"], t_base + t, l, 0, case["B"], n) for l, n in enumerate(case["sizes"])]
                  for t in range(T)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out, trainer = gen_golden.run_reference_call(pc, um, model, call, inputs, target, XI, case, device=DEV)
        modes.append(trainer.last_call_mode)
        grads = gen_golden.param_grads(lins)
        yield g, ci, call, out, grads, lins, trainer
        t_base += T


@pytest.mark.parametrize("name", fixture_names())
def test_facade_replays_reference_harness(name):
    grads_carried = True     # .grad of earlier calls is only present if those calls materialised it
    for g, ci, call, out, grads, lins, trainer in _replay(name):
        generic = any(k in call for k in ("x_lr_discount", "clip_x_grad", "xopt_extra", "update_x_at")) or \
            call.get("update_p_at", "never") == "all"
        assert trainer.last_call_mode == ("stepwise" if (call.get("noise", False) or generic) else "fused")
        # the boundary the user calls, at the contract (energies rel 1e-6, states abs 1e-5), logged
        # (these were rtol 3e-5 / atol 3e-4 and 1e-3, and unlogged).  The results lists are fp32 `.item()`s in
        # the reference and fp64 sums here: 6e-8 of rounding on the reference's side is inside the 1e-6.
        group = "facade (PCTrainer.train_on_batch, %s) vs reference fixtures" % trainer.last_call_mode
        # ReLU-kink events of the reference's trajectory (tests/golden_util.py: a unit within 2e-6 of x = 0 takes its side of the kink
        # by the last bit of a GEMM; g2_cfgM_b64 has six): strict up to the first one, a stated and logged tolerance after it
        tk = first_kink_step(name, ci)
        after = group + ", steps AFTER a ReLU-kink event of the reference trajectory"
        kink_chains = {c for k, evs in relu_kink_events(name).items() if k <= ci for _, c in evs}
        for key in ("energy", "overall") + (("loss",) if g.case["loss"] != "none" else ()):
            got, want = np.asarray(out[key]), g.get(ci, key)
            cut = len(want) if tk is None else tk + 1
            parity_log.close(group, key + "[t]", got[:cut], want[:cut], rtol=1e-6, atol=1e-6)
            if cut < len(want):
                parity_log.close(after, key + "[t]", got[cut:], want[cut:], rtol=1e-5, atol=1e-6)
        for k, v in out.items():
            if not (k.startswith("x_") or k.startswith("out_")):
                continue
            want = g.get(ci, k)
            t_rec = call["T"] if k.startswith("x_final") else int(k.split("_")[1][1:])
            if tk is not None and t_rec > tk and kink_chains:
                keep = [c for c in range(want.shape[0]) if c not in kink_chains]       # those chains follow another valid path for a while
                v, want = v[keep], want[keep]
            parity_log.close(group, "states" if k.startswith("x_") else "outputs", v, want, rtol=0, atol=1e-5 if k.startswith("x_") else 3e-5, err_msg=k)
        updates = call.get("update_p_at", "never") != "never"
        zeroes = updates or call.get("accumulate_p_at", "never") != "never"
        if updates or (trainer.last_call_mode == "stepwise" and (zeroes or grads_carried)):
            # .grad as the reference leaves it (the fused path only materialises grads it needs)
            for k, v in grads.items():
                if g.has(ci, k):
                    ref = g.get(ci, k)
                    np.testing.assert_allclose(v, ref, rtol=3e-4, atol=1e-4 * max(1.0, np.abs(ref).max()), err_msg=k)
        if trainer.last_call_mode == "fused" and not updates:
            grads_carried = False
        if updates:
            for j, lin in enumerate(lins):
                np.testing.assert_allclose(lin.weight.detach().cpu().numpy(), g.get(ci, f"W{j}_after"), rtol=1e-4, atol=2e-5)
                if lin.bias is not None:
                    np.testing.assert_allclose(lin.bias.detach().cpu().numpy(), g.get(ci, f"b{j}_after"), rtol=1e-4, atol=2e-5)


def test_unused_grads_can_be_materialised():
    """Reference quirk: autograd fills .grad on inference-only calls; opt-in here."""
    pc, um = _mods()
    g = Golden("g1_tanh_gaussian_sgd")
    case, call = g.case, g.case["calls"][0]
    model, lins = gen_golden.build_reference_model(pc, case, g.W, g.b, g.X0, device=DEV)
    trainer = pc.PCTrainer(model, T=call["T"], optimizer_x_fn=torch.optim.SGD, optimizer_x_kwargs={"lr": call["lr"]},
                           update_p_at="never", plot_progress_at=[])
    trainer.mcpc_materialize_unused_grads = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        trainer.train_on_batch(inputs=torch.zeros(case["B"], case["n_in"], device=DEV), loss_fn=um.fe_fn,
                               loss_fn_kwargs={"_target": torch.from_numpy(g.target).to(DEV), "_var": case["var"]},
                               is_log_progress=False)
    assert trainer.last_call_mode == "fused"
    for j, lin in enumerate(lins):
        ref = g.get(0, f"gW{j}")
        np.testing.assert_allclose(lin.weight.grad.cpu().numpy(), ref, rtol=3e-4, atol=1e-4 * max(1.0, np.abs(ref).max()))


def test_fused_langevin_is_deterministic_and_matches_oracle_noise():
    """Tagged random_step -> fused Philox path.  Same seed => bit-identical trajectories; and the
    trajectory follows the oracle driven by the NumPy twin of the device generator."""
    from montecarlopredictivecoding_amd.predictive_coding import pc_trainer as pt
    from oracle import mcpc_oracle as mo
    pc, um = _mods()
    case = dict(sizes=[6, 16, 16], acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24, loss="bernoulli",
                var=1.0, perc=0.5, B=40, seed=77, x0_range=2.0, calls=[dict(T=25)])
    W, b, X0, inputs, target = make_case_inputs(case)
    finals = []
    for rep in range(2):
        model, lins = gen_golden.build_reference_model(pc, case, W, b, X0, device=DEV)
        trainer = pc.PCTrainer(model, T=25, optimizer_x_fn=torch.optim.SGD, optimizer_x_kwargs={"lr": 0.03},
                               update_p_at="never", plot_progress_at=[])
        trainer.mcpc_seed = 1234
        pt._PHILOX_STEPS[0] = 500
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = trainer.train_on_batch(inputs=torch.zeros(40, 6, device=DEV), loss_fn=um.bernoulli_fn,
                                         loss_fn_kwargs={"_target": torch.from_numpy(target).to(DEV), "_var": None},
                                         callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": trainer},
                                         is_log_progress=False, is_return_xs=True)
        assert trainer.last_call_mode == "fused"
        finals.append([x.detach().cpu().numpy().copy() for x in trainer.get_model_xs()])
    for a, c in zip(*finals):
        assert np.array_equal(a, c)
    net = mo.NetSpec(sizes=case["sizes"], acts=[mo.ACT_RELU] * 3, W=W, b=b)
    ref = mo.run(net, inputs, X0, mo.LossSpec(mo.LOSS_BERNOULLI, target), mo.XOpt(mo.OPT_SGD, 0.03), 25,
                 noise=lambda t, l: philox.layer_normals(1234, 500 + t, l, 0, 40, case["sizes"][l]))
    for l in range(3):
        np.testing.assert_allclose(finals[0][l], ref.xs[l], rtol=0, atol=5e-4)
    np.testing.assert_allclose(res["overall"], ref.overall, rtol=5e-5)


def _users_own_random_step(t, _pc_trainer, var=2.):
    """What a script that keeps its own utils/model.py passes as callback_after_t: the reference's Langevin callback
    (utils/model.py:35-44) restated verbatim in behaviour -- untagged, defined outside this package, NOT imported from
    /root/reference."""
    xs = _pc_trainer.get_model_xs()
    optimizer = _pc_trainer.get_optimizer_x()
    for x in xs:
        x.grad.normal_(0., np.sqrt(var / optimizer.defaults['lr']))
    optimizer.step()


def test_untagged_reference_random_step_is_fused_and_other_callbacks_warn():
    """north_star: "figure_*.py scripts run unchanged".  The reference's own random_step is recognised by what its code CAN do
    (recognise._static_langevin_check: straight-line code over the handful of names a Langevin kick needs) and by what it DOES
    (recognise._probe_langevin_callback), and fused as Philox noise: the call is bitwise the one made with this package's tagged
    random_step.  A callback that is not a plain Langevin kick still runs (step-wise HIP path) -- and says so."""
    from montecarlopredictivecoding_amd.predictive_coding import pc_trainer as pt
    pc, um = _mods()
    case = dict(sizes=[6, 16, 16], acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=6, n_out=24, loss="bernoulli",
                var=1.0, perc=0.5, B=40, seed=77, x0_range=2.0, calls=[dict(T=25)])
    W, b, X0, inputs, target = make_case_inputs(case)
    tgt = torch.from_numpy(target).to(DEV)
    finals, overalls = {}, {}

    def half_kick(t, _pc_trainer):                    # not a Langevin kick: rescales the noise after drawing it
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., 1.).mul_(0.5)
        _pc_trainer.get_optimizer_x().step()

    for key, cb, kw in (("tagged", um.random_step, {"var": 1.5}), ("own", _users_own_random_step, {"var": 1.5}), ("other", half_kick, {})):
        model, lins = gen_golden.build_reference_model(pc, case, W, b, X0, device=DEV)
        trainer = pc.PCTrainer(model, T=25, optimizer_x_fn=torch.optim.SGD, optimizer_x_kwargs={"lr": 0.03},
                               update_p_at="never", plot_progress_at=[])
        trainer.mcpc_seed = 99
        pt._PHILOX_STEPS[0] = 700
        torch.manual_seed(5)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            res = trainer.train_on_batch(inputs=torch.zeros(40, 6, device=DEV), loss_fn=um.bernoulli_fn,
                                         loss_fn_kwargs={"_target": tgt, "_var": None}, callback_after_t=cb,
                                         callback_after_t_kwargs=dict(kw, _pc_trainer=trainer), is_log_progress=False,
                                         is_return_results_every_t=False, is_checking_after_callback_after_t=False)
        slow = [w for w in caught if issubclass(w.category, RuntimeWarning) and "leaves the fused HIP loop" in str(w.message)]
        if key == "other":
            assert trainer.last_call_mode == "stepwise" and len(slow) == 1
            assert "not provably a plain Langevin kick" in str(slow[0].message) and ".mul_" in str(slow[0].message)
        else:
            assert trainer.last_call_mode == "fused" and not slow
            said = [w for w in caught if issubclass(w.category, RuntimeWarning) and "NOT invoked per step" in str(w.message)]
            assert len(said) <= (1 if key == "own" else 0)      # an untagged callback that is fused is named, once per process
        finals[key] = [x.detach().clone() for x in trainer.get_model_xs()]
        overalls[key] = res["overall"]
    for a, c in zip(finals["tagged"], finals["own"]):
        assert torch.equal(a, c)
    assert overalls["tagged"] == overalls["own"]
    assert all(bool(torch.isfinite(x).all()) for x in finals["other"])
    assert not torch.equal(finals["other"][1], finals["own"][1])


def test_kat_linear_gaussian_posterior():
    """figure_2.py:29-79: prior x1~N(0.2,1), y = 2*x1 + N(0,1), y = 1  =>  posterior N(0.44, 0.2)."""
    import torch.nn as nn
    pc, um = _mods()
    B = 8192
    model = nn.Sequential(nn.Linear(1, 1), pc.PCLayer(sample_x_fn=um.sample_x_fn_cte), nn.Linear(1, 1, bias=False))
    model.train()
    nn.init.constant_(model[0].bias, 0.2)
    nn.init.constant_(model[2].weight, 2.0)
    model.to(DEV)
    cfg = {"T_pc": 500, "optimizer_x_fn_pc": torch.optim.Adam, "optimizer_x_kwargs_pc": {"lr": 0.02},
           "mixing": 400, "sampling": 600, "optimizer_x_kwargs_mcpc": {"lr": 0.02}, "optimizer_p_fn_mcpc": torch.optim.Adam}
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_pc_trainer, get_mcpc_trainer
    pc_tr = get_pc_trainer(model, cfg, is_mcpc=True, training=False)
    mc_tr = get_mcpc_trainer(model, cfg, training=False)
    data = torch.ones(B, 1, device=DEV)
    zeros = torch.zeros(B, 1, device=DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pc_tr.train_on_batch(inputs=zeros, loss_fn=um.fe_fn, loss_fn_kwargs={"_target": data, "_var": 1.0},
                             is_log_progress=False, is_return_results_every_t=False)
        x_map = model[1].get_x().detach().cpu().numpy()
        assert np.abs(x_map - 0.44).max() < 5e-3          # MAP = posterior mean for a Gaussian
        res = mc_tr.train_on_batch(inputs=zeros, loss_fn=um.fe_fn, loss_fn_kwargs={"_target": data, "_var": 1.0},
                                   callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": mc_tr},
                                   is_sample_x_at_batch_start=False, is_log_progress=False,
                                   is_return_results_every_t=True, is_return_representations=True)
    assert pc_tr.last_call_mode == "fused" and mc_tr.last_call_mode == "fused"
    reps = torch.stack(res["representations"][400:]).numpy().reshape(-1)      # sampling phase, all chains
    # SGLD with step lr inflates the stationary variance by O(lr): exact discrete-time value for this
    # quadratic energy (curvature a = 1 + W^2 = 5): var = 2*lr / (1 - (1-a*lr)^2) = 0.2/(1 - a*lr/2) = 0.2105
    assert abs(reps.mean() - 0.44) < 5e-3
    assert abs(reps.var() - 0.2 / (1 - 5 * 0.02 / 2)) < 6e-3


def test_kat_linear_generation_marginal():
    """figure_3.py:50-76: x1~N(0.5,1), x0~N(2*x1, 1)  =>  marginal x0 ~ N(1.0, 5.0); sensory PCLayer diffuses too."""
    import torch.nn as nn
    pc, um = _mods()
    B = 8192
    var = 1.0
    model = nn.Sequential(nn.Linear(1, 1), pc.PCLayer(sample_x_fn=um.sample_x_fn_normal), nn.Linear(1, 1, bias=False),
                          pc.PCLayer(energy_fn=lambda inputs: (1 / var) * 0.5 * (inputs["mu"] - inputs["x"]) ** 2,
                                     sample_x_fn=um.sample_x_fn_normal))
    model.train()
    nn.init.constant_(model[0].bias, 0.5)
    nn.init.constant_(model[2].weight, 2.0)
    model.to(DEV)
    cfg = {"mixing": 0, "sampling": 1500, "optimizer_x_kwargs_mcpc": {"lr": 0.02}}
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer
    mc_tr = get_mcpc_trainer(model, cfg, training=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = mc_tr.train_on_batch(inputs=torch.zeros(B, 1, device=DEV), callback_after_t=um.random_step,
                                   callback_after_t_kwargs={"_pc_trainer": mc_tr}, is_log_progress=False,
                                   is_return_results_every_t=False, is_return_outputs=True)
    assert mc_tr.last_call_mode == "fused"
    x0 = res["outputs"][-1].detach().cpu().numpy().reshape(-1)
    assert abs(x0.mean() - 1.0) < 0.12
    assert abs(x0.var() - 5.0) < 0.45
