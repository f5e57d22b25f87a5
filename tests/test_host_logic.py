"""Host-side logic of the facade that needs no GPU: schedules, recognisers, PCLayer torch semantics,
loud failure when no device / unsupported configuration."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn

import montecarlopredictivecoding_amd.predictive_coding as pc
from montecarlopredictivecoding_amd import _lib as L
from montecarlopredictivecoding_amd.predictive_coding import recognise
from montecarlopredictivecoding_amd.predictive_coding.pc_layer import probe_energy_coefficient
from montecarlopredictivecoding_amd.utils import model as um
from montecarlopredictivecoding_amd.utils.training_evaluation import (get_mcpc_trainer, get_mcpc_trainer_one_sample,
                                                                      get_pc_trainer)

CFG = dict(input_size=5, hidden_size=12, hidden2_size=9, output_size=20, activation_fn="tanh",
           T_pc=7, optimizer_x_fn_pc=torch.optim.Adam, optimizer_x_kwargs_pc={"lr": 0.1},
           optimizer_p_fn=torch.optim.Adam, optimizer_p_kwargs={"lr": 0.01},
           mixing=3, sampling=4, K=6, optimizer_x_kwargs_mcpc={"lr": 0.03},
           optimizer_p_fn_mcpc=torch.optim.Adam, optimizer_p_kwargs_mcpc={"lr": 0.01})


def test_schedule_strings_and_lists():
    m = um.get_model(CFG, False)
    tr = pc.PCTrainer(m, T=8, update_x_at="last_half", update_p_at="last", accumulate_p_at=[2, 3], plot_progress_at=[])
    assert tr._update_x_at == [4, 5, 6, 7] and tr._update_p_at == [7] and tr._accumulate_p_at == [2, 3]
    assert pc.PCTrainer(m, T=3, update_p_at="never", plot_progress_at=[])._update_p_at == []
    assert pc.PCTrainer(m, T=3, plot_progress_at=[])._update_p_at == [0, 1, 2]          # default 'all'
    with pytest.raises(AssertionError):
        pc.PCTrainer(m, T=3, update_p_at=[3], plot_progress_at=[])
    with pytest.raises(NotImplementedError):
        pc.PCTrainer(m, T=3, update_p_at="sometimes", plot_progress_at=[])
    with pytest.warns(RuntimeWarning):
        pc.PCTrainer(m, T=2, plot_progress_at=[])       # T < num_pc_layers + 1 only warns (figure_4.py uses T_pc=1)


def test_trainer_factories_and_grad_windows():
    m = um.get_model(CFG, False)
    mc = get_mcpc_trainer(m, CFG, training=True)
    assert mc.get_T() == 7 and mc._accumulate_p_at == [3, 4, 5, 6] and mc._update_p_at == [6]
    assert mc._grad_window() == 3                       # zeroed at accumulate_p_at[0], summed to the end
    assert get_mcpc_trainer(m, CFG, training=False)._update_p_at == []
    one = get_mcpc_trainer_one_sample(m, CFG, training=True)
    assert one._grad_window() == 5                      # single Monte Carlo sample: last step only
    mp = get_pc_trainer(m, CFG, is_mcpc=True)
    assert mp._update_p_at == [] and mp._grad_window() is None
    assert get_pc_trainer(m, CFG, training=True)._grad_window() == 6
    odd = pc.PCTrainer(m, T=7, update_p_at="last", accumulate_p_at=[1, 2], plot_progress_at=[])
    assert odd._grad_window() == 6                      # update step outside the accumulate list zeroes again
    assert len(list(mc.get_model_parameters())) == 8 and mc.get_num_pc_layers() == 3 and mc.get_least_T() == 4


def test_describe_model_variants():
    net, why = recognise.describe_model(um.get_model(CFG, False))
    assert why == "" and net.sizes == [5, 12, 9] and net.acts == [L.ACT_TANH] * 3 and net.n_in == 5 and net.n_out == 20
    toy = nn.Sequential(nn.Linear(1, 1), pc.PCLayer(), nn.Linear(1, 1, bias=False),
                        pc.PCLayer(energy_fn=lambda inputs: (1 / 0.5) * 0.5 * (inputs["mu"] - inputs["x"]) ** 2))
    net, why = recognise.describe_model(toy)
    assert net.n_out == 0 and net.ecoef == [1.0, 2.0] and net.acts == [L.ACT_IDENTITY] * 2
    bad = nn.Sequential(nn.Linear(3, 3), pc.PCLayer(energy_fn=lambda i: (i["mu"] - i["x"]).abs()), nn.Linear(3, 2))
    assert recognise.describe_model(bad)[0] is None
    assert recognise.describe_model(nn.Sequential(nn.Linear(3, 3), nn.ReLU(), nn.Linear(3, 2)))[0] is None
    assert recognise.describe_model(nn.Sequential(nn.Linear(3, 3), pc.PCLayer(M=torch.ones(3)), nn.Linear(3, 2)))[0] is None
    assert recognise.describe_model(nn.Linear(3, 3))[0] is None


def test_energy_probe():
    assert probe_energy_coefficient(lambda i: 0.5 * (i["mu"] - i["x"]) ** 2) == pytest.approx(1.0)
    assert probe_energy_coefficient(lambda i: (1 / 0.3) * 0.5 * (i["mu"] - i["x"]) ** 2) == pytest.approx(1 / 0.3)
    assert probe_energy_coefficient(lambda i: 0.5 * (i["mu"] - i["x"].detach()) ** 2) == pytest.approx(1.0)
    assert probe_energy_coefficient(lambda i: (i["mu"] - i["x"]).norm(2)) is None
    assert probe_energy_coefficient(lambda i: 0.5 * (i["mu"] - i["x"]) ** 2 + i["x"] ** 2) is None


def test_describe_loss_tags_and_probes():
    B, n = 6, 20
    y = (torch.rand(B, n) < 0.3).float()
    dev = torch.device("cpu")
    d, _ = recognise.describe_loss(None, {}, n, B, dev)
    assert d.kind == L.LOSS_NONE and not d.returns_value
    d, _ = recognise.describe_loss(um.zero_fn, {}, n, B, dev)
    assert d.kind == L.LOSS_NONE and d.returns_value
    d, _ = recognise.describe_loss(um.fe_fn, {"_target": y, "_var": 0.3}, n, B, dev)
    assert (d.kind, d.var, d.mask_start) == (L.LOSS_GAUSSIAN, 0.3, 0)
    d, _ = recognise.describe_loss(um.bernoulli_fn_mask, {"_target": y, "_var": None, "perc": 0.25}, n, B, dev)
    assert (d.kind, d.mask_start) == (L.LOSS_BERNOULLI, 15)
    d, _ = recognise.describe_loss(um.fe_fn_mask, {"_target": y, "_var": 2.0}, n, B, dev)
    assert (d.kind, d.mask_start) == (L.LOSS_GAUSSIAN, 10)
    # untagged callables (e.g. the reference's own functions or lambdas) are recognised by behaviour
    d, _ = recognise.describe_loss(lambda o, _target: nn.BCEWithLogitsLoss(reduction="sum")(o[:, -5:], _target[:, -5:]),
                                   {"_target": y}, n, B, dev)
    assert (d.kind, d.mask_start) == (L.LOSS_BERNOULLI, 15)
    d, _ = recognise.describe_loss(lambda o, _target, _var: (1 / _var) * 0.5 * (o - _target).pow(2).sum(),
                                   {"_target": y, "_var": 0.7}, n, B, dev)
    assert d.kind == L.LOSS_GAUSSIAN and d.var == pytest.approx(0.7, rel=1e-5)
    d, why = recognise.describe_loss(lambda o, _target: (o - _target).abs().sum(), {"_target": y}, n, B, dev)
    assert d is None and "neither" in why
    d, why = recognise.describe_loss(um.fe_fn, {"_target": y, "_var": 1.0}, 0, B, dev)
    assert d is None


def test_describe_optimizer_and_callback():
    assert recognise.describe_x_optimizer(torch.optim.SGD, {"lr": 0.03})[0].kind == L.XOPT_SGD
    a = recognise.describe_x_optimizer(torch.optim.Adam, {"lr": 0.1, "betas": (0.8, 0.9), "eps": 1e-6})[0]
    assert (a.kind, a.betas, a.eps) == (L.XOPT_ADAM, (0.8, 0.9), 1e-6)
    assert recognise.describe_x_optimizer(torch.optim.SGD, {"lr": 0.1, "momentum": 0.9})[0] is None
    assert recognise.describe_x_optimizer(torch.optim.RMSprop, {"lr": 0.1})[0] is None
    tr = object()
    assert recognise.describe_callback(None, {}, tr) == (None, "")
    assert recognise.describe_callback(um.random_step, {"_pc_trainer": tr}, tr) == (2.0, "")
    assert recognise.describe_callback(um.random_step, {"_pc_trainer": tr, "var": 0.5}, tr) == (0.5, "")
    assert recognise.describe_callback(lambda t: None, {}, tr)[1] != ""


def test_pclayer_torch_semantics():
    layer = pc.PCLayer(sample_x_fn=lambda inp: inp["mu"].detach().clone() + 1.0)
    mu = torch.randn(4, 3)
    assert not layer.training and layer(mu) is mu                # a fresh layer is in eval mode: identity
    layer.train()
    with pytest.warns(RuntimeWarning):                           # no x yet: sampled with a warning
        out = layer(mu)
    assert out is layer.get_x() and torch.allclose(out, mu + 1.0)
    assert float(layer.energy()) == pytest.approx(0.5 * 12)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        layer(mu)                                                # same shape: no resample, no warning
    with pytest.warns(RuntimeWarning):
        layer(torch.randn(5, 3))                                 # batch size changed -> resample
    assert tuple(layer.get_x().shape) == (5, 3)
    layer.set_is_sample_x(True)
    layer(torch.zeros(5, 3))
    assert torch.allclose(layer.get_x(), torch.ones(5, 3)) and not layer.get_is_sample_x()
    keep = pc.PCLayer(is_keep_energy_per_datapoint=True)
    keep.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        keep(torch.randn(4, 3))
    assert tuple(keep.energy_per_datapoint().shape) == (4, 1)


@pytest.mark.skipif(torch.cuda.is_available(), reason="with a HIP device a CPU-built model is staged onto it (tests/test_gpu_staging.py)")
def test_cpu_model_fails_loudly_not_silently():
    # no HIP device in the CPU suite: nothing computes on the CPU instead -- every path raises
    m = um.get_model(CFG, False)
    tr = get_mcpc_trainer(m, CFG, training=False)
    y = torch.zeros(4, 20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(L.MCPCLibraryError, match="no CPU path"):
            tr.train_on_batch(inputs=torch.zeros(4, 5), loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None},
                              is_log_progress=False)
        m.eval()
        with pytest.raises(AssertionError):
            tr.train_on_batch(inputs=torch.zeros(4, 5), is_log_progress=False)


def test_unsupported_configurations_name_their_reason():
    # what the kernels do not express goes to the generic torch loop -- but never as a way around a missing GPU: without a HIP
    # device every call fails loudly (no CPU path)
    m = nn.Sequential(nn.Linear(3, 3), pc.PCLayer(M=torch.ones(3)), nn.Linear(3, 2))
    m.train()
    tr = pc.PCTrainer(m, T=4, plot_progress_at=[])
    if not torch.cuda.is_available():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with pytest.raises(L.MCPCLibraryError, match="no CPU path"):
                tr.train_on_batch(inputs=torch.zeros(2, 3), is_log_progress=False)
    plan, why = tr._plan(torch.zeros(2, 3), None, {}, False, False, None, None, {}, {}, False, False)
    assert plan is None and "S/M masks" in why
    tr2 = pc.PCTrainer(um.get_model(CFG, False), T=4)          # plot_progress_at defaults to 'all'
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(NotImplementedError, match="plot_progress"):
            tr2.train_on_batch(inputs=torch.zeros(2, 5), is_log_progress=False)


def test_random_step_torch_body_matches_reference_arithmetic():
    """Executed on the step-wise path: overwrite x.grad with N(0, sqrt(var/lr)), step once more."""
    x = nn.Parameter(torch.zeros(20000))

    class T_:
        def get_model_xs(self): return [x]
        def get_optimizer_x(self): return opt
    opt = torch.optim.SGD([x], lr=0.05)
    torch.manual_seed(0)
    um.random_step(0, T_(), var=2.0)
    assert abs(float(x.std()) - np.sqrt(2.0 * 0.05)) < 0.01 and abs(float(x.mean())) < 0.01


def _reference_random_step(t, _pc_trainer, var=2.):
    """A user's own copy of the reference's Langevin callback (utils/model.py:35-44), verbatim in behaviour: untagged, defined
    outside this package -- what a script that keeps its own utils/model.py passes as callback_after_t."""
    xs = _pc_trainer.get_model_xs()
    optimizer = _pc_trainer.get_optimizer_x()
    for x in xs:
        x.grad.normal_(0., np.sqrt(var / optimizer.defaults['lr']))
    optimizer.step()


def test_untagged_reference_random_step_is_recognised_by_behaviour():
    m = um.get_model(CFG, False)
    tr = get_mcpc_trainer(m, CFG, training=False)
    assert not hasattr(_reference_random_step, "_mcpc")
    torch.manual_seed(11); np.random.seed(12)
    want = (torch.rand(4), np.random.rand(3))
    torch.manual_seed(11); np.random.seed(12)
    recognise._FUSED_ANNOUNCED.discard(_reference_random_step)
    with pytest.warns(RuntimeWarning, match="_reference_random_step.*fused.*NOT invoked per step"):      # said once, by name
        assert recognise.describe_callback(_reference_random_step, {"_pc_trainer": tr}, tr) == (2.0, "")
    assert recognise.describe_callback(_reference_random_step, {"_pc_trainer": tr, "var": 0.7}, tr) == (0.7, "")
    # the probe consumed nothing of the script's seeded generators
    assert torch.equal(torch.rand(4), want[0]) and np.array_equal(np.random.rand(3), want[1])
    # and the trainer's own latents / optimizer were never touched (none exist yet)
    assert tr.get_optimizer_x() is None

    def twice(t, _pc_trainer):
        _reference_random_step(t, _pc_trainer); _pc_trainer.get_optimizer_x().step()

    def uniform(t, _pc_trainer):
        for x in _pc_trainer.get_model_xs():
            x.grad.uniform_(-1, 1)
        _pc_trainer.get_optimizer_x().step()

    def rescaled(t, _pc_trainer):
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., 1.).mul_(3.)
        _pc_trainer.get_optimizer_x().step()

    def annealed(t, _pc_trainer):
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., 1. + t)
        _pc_trainer.get_optimizer_x().step()

    def writes_x(t, _pc_trainer):
        for x in _pc_trainer.get_model_xs():
            x.data.add_(1.0)
        _reference_random_step(t, _pc_trainer)

    def wants_model(t, _pc_trainer):
        _pc_trainer.get_model().eval()

    # what a behavioural probe at a few t cannot see -- a kick that also logs, counts in a closure, skips a step, or
    # clamps beyond the probe's range -- must NOT be fused (fusing means the callable is never called again)
    seen = []

    def counting(t, _pc_trainer):
        seen.append(t)
        _reference_random_step(t, _pc_trainer)

    def logging_kick(t, _pc_trainer, var=2.):
        print("kick", var)
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., np.sqrt(var / _pc_trainer.get_optimizer_x().defaults['lr']))
        _pc_trainer.get_optimizer_x().step()

    def skips_a_step(t, _pc_trainer, var=2.):
        if t == 7:
            return
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., np.sqrt(var / _pc_trainer.get_optimizer_x().defaults['lr']))
        _pc_trainer.get_optimizer_x().step()

    def clamps(t, _pc_trainer, var=2.):
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., np.sqrt(var / _pc_trainer.get_optimizer_x().defaults['lr'])).clamp_(-100, 100)
        _pc_trainer.get_optimizer_x().step()

    import functools
    for fn, word in ((twice, "global '_reference_random_step'"), (uniform, ".uniform_"), (rescaled, ".mul_"), (annealed, "reads its step argument"),
                     (writes_x, ".data"), (wants_model, ".get_model"), (counting, "closes over"), (logging_kick, "global 'print'"),
                     (skips_a_step, "reads its step argument"), (clamps, ".clamp_"),
                     (functools.partial(_reference_random_step, var=1.0), "not a plain Python function")):
        var, why = recognise.describe_callback(fn, {"_pc_trainer": tr}, tr)
        assert var is None and word in why, (getattr(fn, "__name__", fn), why)
    assert seen == []                                       # ... and none of them was run to find that out
    other = get_mcpc_trainer(m, CFG, training=False)
    assert recognise.describe_callback(_reference_random_step, {"_pc_trainer": other}, tr)[0] is None     # bound to another trainer
    assert recognise.describe_callback(lambda t: None, {}, tr)[0] is None                                  # a closure: cannot be probed


def test_static_langevin_check_is_an_allow_list():
    """the bytecode check of an untagged callback must not depend on which opcodes a deny-list happens to know.  A plain kick
    (the shape of the reference's random_step, utils/model.py:35-44) passes; reading t, branching, calling anything else, or an opcode
    outside the allow-list fails -- such a callable keeps the step-wise path, where it is really called."""
    from montecarlopredictivecoding_amd.predictive_coding import recognise

    def kick(t, _pc_trainer, var=2.):
        xs = _pc_trainer.get_model_xs()
        optimizer = _pc_trainer.get_optimizer_x()
        for x in xs:
            x.grad.normal_(0., np.sqrt(var / optimizer.defaults['lr']))
        optimizer.step()

    def reads_t(t, _pc_trainer, var=2.):
        optimizer = _pc_trainer.get_optimizer_x()
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., np.sqrt(var / optimizer.defaults['lr']) * t)
        optimizer.step()

    def branches(t, _pc_trainer, var=2.):
        optimizer = _pc_trainer.get_optimizer_x()
        for x in _pc_trainer.get_model_xs():
            if var:
                x.grad.normal_(0., np.sqrt(var / optimizer.defaults['lr']))
        optimizer.step()

    def logs(t, _pc_trainer, var=2.):
        optimizer = _pc_trainer.get_optimizer_x()
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., np.sqrt(var / optimizer.defaults['lr']))
        optimizer.step()
        print("kicked")

    def formats(t, _pc_trainer, var=2.):
        optimizer = _pc_trainer.get_optimizer_x()
        for x in _pc_trainer.get_model_xs():
            x.grad.normal_(0., np.sqrt(var / optimizer.defaults['lr']))
        optimizer.step()
        return f"{var}"                                   # FORMAT_VALUE / BUILD_STRING: not on the allow-list

    assert recognise._static_langevin_check(kick) == ""
    assert "reads its step argument" in recognise._static_langevin_check(reads_t)
    assert "branch" in recognise._static_langevin_check(branches)
    assert "global 'print'" in recognise._static_langevin_check(logs)
    assert "no use for" in recognise._static_langevin_check(formats)
    assert "closes over" in recognise._static_langevin_check((lambda k: (lambda t, _pc_trainer: k))(1))
    # every opcode of the accepted kick is on the allow-list of THIS interpreter -- and a fused-name opcode carrying t is caught
    import dis
    assert {i.opname for i in dis.get_instructions(kick)} <= recognise._KICK_ALLOWED_OPS
    import weakref
    assert isinstance(recognise._FUSED_ANNOUNCED, weakref.WeakSet)


def test_staged_calls_thread_cap_is_reentrant_and_respects_the_users_own_setting():
    """the cap on torch's intra-op threads during a staged call (pc_trainer._few_cpu_threads) is process-global state: nested
    calls share one cap, the outermost restores the caller's value, and a value the user sets inside a callback is kept."""
    import torch
    from montecarlopredictivecoding_amd.predictive_coding.pc_trainer import _few_cpu_threads as cap
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(16)
        with cap(True):
            assert torch.get_num_threads() == cap.CAP
            with cap(True):
                assert torch.get_num_threads() == cap.CAP
            assert torch.get_num_threads() == cap.CAP          # the inner exit did not lift the outer cap
        assert torch.get_num_threads() == 16
        with cap(True):
            torch.set_num_threads(4)                            # e.g. a callback's own choice
        assert torch.get_num_threads() == 4
        with cap(False):
            assert torch.get_num_threads() == 4
    finally:
        torch.set_num_threads(before)
