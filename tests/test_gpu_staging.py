"""A model built on the CPU is staged onto the MI355X, not refused (VERDICT r4 missing #1; round 5).

Most of the reference's call sites build their models on the CPU (figure_2.py:29-75, figure_4.py:537, figure_5.py:25-27,
figure_6.py:55-93).  The facade copies W, b, x, inputs and the target to the GPU, runs the SAME HIP path (fused or step-wise) and
writes x, the results and param.grad back to the CPU tensors the script holds.  So a staged call must equal the device call BITWISE
(same kernels on the same numbers) -- and thereby the reference's fixtures at the tolerances of tests/test_gpu_facade.py -- and leave
every tensor the script can see on the CPU."""
import warnings

import numpy as np
import pytest
import torch

from oracle import gen_golden, philox
from tests.golden_util import Golden
from tests import parity_log

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

FIXTURES = ["g1_relu_bernoulli_sgdnoise", "g1_tanh_gaussian_mask_adam", "g1_relu_zero_sgd", "g3_accumulate_last", "g4_linear_gaussian",
            "g5_sensory_wide", "g6_nonzero_inputs", "g7_map_then_mcpc", "g8_ragged", "g9_update_p_all", "g9_clip_after_backward"]


def _mods():
    import montecarlopredictivecoding_amd.predictive_coding as pc
    import montecarlopredictivecoding_amd.utils.model as um
    return pc, um


def _replay(name, device):
    pc, um = _mods()
    g = Golden(name)
    case = g.case
    model, lins = gen_golden.build_reference_model(pc, case, g.W, g.b, g.X0, device=device)
    t_base, outs = 0, []
    for ci, call in enumerate(case["calls"]):
        T = call["T"]
        XI = None
        if call.get("noise", False):
            XI = [[philox.layer_normals(case["seed"], t_base + t, l, 0, case["B"], n) for l, n in enumerate(case["sizes"])]
                  for t in range(T)]
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            out, trainer = gen_golden.run_reference_call(pc, um, model, call, g.inputs, g.target, XI, case, device=device)
        staged_said = [1] if getattr(trainer, "_staging_announced", False) else []       # (the harness silences warnings itself)
        grads = {k: v.copy() for k, v in gen_golden.param_grads(lins).items()}
        devices = dict(x={str(x.device.type) for x in trainer.get_model_xs()}, p={str(p.device.type) for lin in lins for p in lin.parameters()},
                       g={str(p.grad.device.type) for lin in lins for p in lin.parameters() if p.grad is not None})
        outs.append(dict(out=out, grads=grads, mode=trainer.last_call_mode, devices=devices, said=len(staged_said),
                         W=[lin.weight.detach().cpu().numpy().copy() for lin in lins]))
        t_base += T
    return g, outs


@pytest.mark.parametrize("name", FIXTURES)
def test_cpu_built_model_is_staged_and_equals_the_device_run_bitwise(name):
    g, on_cpu = _replay(name, "cpu")
    _, on_gpu = _replay(name, DEV)
    stepped = False          # a parameter step has happened in an earlier call of this fixture
    for ci, (c, d) in enumerate(zip(on_cpu, on_gpu)):
        assert c["mode"] == d["mode"] and c["mode"] in ("fused", "stepwise")         # never the generic loop, never a CPU computation
        assert c["said"] == 1 and d["said"] == 0                                       # announced (once per trainer) iff staged
        assert c["devices"]["x"] == {"cpu"} and c["devices"]["p"] == {"cpu"} and c["devices"]["g"] <= {"cpu"}
        assert d["devices"]["x"] == {"cuda"}
        # Everything the ENGINE computes is bitwise the device run's.  What torch computes around it is torch's: the staged model's
        # `grad / div` and optimizer_p.step() run on the CPU -- the reference's own arithmetic -- while a device model's run on the GPU,
        # where torch divides by a scalar as a multiplication by its reciprocal: last-bit differences in param.grad and, after a
        # parameter step, in the weights every later call starts from.
        call = g.case["calls"][ci]
        torch_side = call.get("update_p_at", "never") != "never"
        assert sorted(c["out"]) == sorted(d["out"])
        for k in c["out"]:
            a, b = np.asarray(c["out"][k]), np.asarray(d["out"][k])
            if stepped or (torch_side and call.get("update_p_at") == "all"):
                np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(b).max())), err_msg=f"{ci} {k}")
            else:
                assert np.array_equal(a, b), (ci, k)
        assert sorted(c["grads"]) == sorted(d["grads"])
        for k in c["grads"]:
            if torch_side or stepped or c["mode"] == "stepwise":
                np.testing.assert_allclose(c["grads"][k], d["grads"][k], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(d["grads"][k]).max())), err_msg=f"{ci} {k}")
            else:
                assert np.array_equal(c["grads"][k], d["grads"][k]), (ci, k)
        for a, b in zip(c["W"], d["W"]):
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-6)
        stepped = stepped or torch_side
        # ... and against the reference's own outputs, at the contract (energies rel 1e-6, states abs 1e-5; Adam-x: see DESIGN section 2)
        out = c["out"]
        for key in ("energy", "overall") + (("loss",) if g.case["loss"] != "none" else ()):
            parity_log.close("staged cpu model vs reference fixtures", key, out[key], g.get(ci, key), rtol=1e-6, atol=1e-6)
        for k, v in out.items():
            if k.startswith("x_"):
                parity_log.close("staged cpu model vs reference fixtures", "states", v, g.get(ci, k), rtol=0, atol=1e-5)


def test_staged_results_live_where_the_model_lives():
    """`outputs` are live tensors on the model's device in the reference (pc_trainer.py:733,770); representations / xs are CPU copies."""
    pc, um = _mods()
    torch.manual_seed(3)
    cfg = dict(input_size=5, hidden_size=16, hidden2_size=16, output_size=20, activation_fn="relu")
    m = um.get_model(cfg, False)
    tr = pc.PCTrainer(m, T=7, optimizer_x_fn=torch.optim.Adam, optimizer_x_kwargs={"lr": 0.05}, update_p_at="last",
                      accumulate_p_at="last_half", optimizer_p_fn=torch.optim.SGD, optimizer_p_kwargs={"lr": 0.01}, plot_progress_at=[])
    y = (torch.rand(9, 20) > 0.5).float()
    w_before = m[-1].weight.detach().clone()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = tr.train_on_batch(inputs=torch.zeros(9, 5), loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None},
                                is_log_progress=False, is_return_outputs=True, is_return_xs=True, is_return_representations=True)
    assert tr.last_call_mode == "fused"
    assert len(res["outputs"]) == 7 and all(o.device.type == "cpu" and tuple(o.shape) == (9, 20) for o in res["outputs"])
    assert all(x.device.type == "cpu" for step in res["xs"] for x in step) and res["representations"][0].device.type == "cpu"
    assert all(p.device.type == "cpu" and p.grad is not None and p.grad.device.type == "cpu" for p in tr.get_model_parameters())
    assert not torch.equal(m[-1].weight.detach(), w_before)                 # optimizer_p stepped on the CPU parameters ...
    st = tr.get_optimizer_x().state[m[1].get_x()]                           # ... and optimizer_x is left as T fused Adam steps leave it
    assert st["exp_avg"].device.type == "cpu" and float(st["step"]) == 7.0
    # the next call sees the updated CPU weights (re-staged by version), and a call that keeps optimizer_x continues step-wise
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr.train_on_batch(inputs=torch.zeros(9, 5), loss_fn=um.bernoulli_fn, loss_fn_kwargs={"_target": y, "_var": None},
                          is_log_progress=False, is_sample_x_at_batch_start=False, is_reset_optimizer_x_at_batch_start=False)
    assert tr.last_call_mode == "stepwise"


def test_inputs_on_another_device_than_the_model_are_not_fused():
    pc, um = _mods()
    cfg = dict(input_size=5, hidden_size=16, hidden2_size=16, output_size=20, activation_fn="relu")
    m = um.get_model(cfg, False)
    tr = pc.PCTrainer(m, T=3, update_p_at="never", plot_progress_at=[])
    plan, why = tr._plan(torch.zeros(4, 5, device=DEV), None, {}, False, False, None, None, {}, {}, False, False)
    assert plan is None and "inputs on cuda" in why
