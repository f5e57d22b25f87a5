"""Seeded sweep of random network shapes and call modes THROUGH THE FACADE (the boundary a user of the reference calls:
`PCTrainer.train_on_batch`) against the NumPy oracle, at the contract of BASELINE.md section 3 -- energies rel 1e-6, states abs 1e-5.

tests/test_gpu_fuzz.py sweeps the same shapes through the C ABI; the facade adds what sits between a script and the ABI: model
recognition (ragged widths, bias-free Linears, scaled energies, identity / ReLU / tanh), loss recognition (tagged and behaviourally
probed callables: the masked losses arrive as untagged lambdas), the fused Langevin callback as Philox noise (compared with the oracle
driven by the NumPy twin of the device generator), Adam-x MAP calls, the gradient window and its normalisation, a real `optimizer_p`
step -- with the model on the GPU and, every other case, built on the CPU and staged (round 5)."""
import warnings

import numpy as np
import pytest
import torch

from oracle import gen_golden
from oracle import mcpc_oracle as mo
from oracle import philox
from oracle.cases import make_case_inputs
from tests import parity_log
from tests.test_gpu_fuzz import _random_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_CASES = 24


@pytest.mark.parametrize("i", range(N_CASES))
def test_random_shape_through_the_facade_matches_oracle(i):
    import montecarlopredictivecoding_amd.predictive_coding as pc
    import montecarlopredictivecoding_amd.utils.model as um
    from montecarlopredictivecoding_amd.predictive_coding import pc_trainer as pt
    case, mode = _random_case(i)
    if case["loss"] == "zero" and case["n_out"] == 0:
        pytest.skip("no read-out to apply zero_fn to")
    sizes, n_out, B, T = case["sizes"], case["n_out"], case["B"], case["calls"][0]["T"]
    W, b, X0, inputs, target = make_case_inputs(case)
    device = "cpu" if i % 2 else DEV                                     # every other case: a CPU-built model, staged per call
    act_o = {"relu": mo.ACT_RELU, "tanh": mo.ACT_TANH, "identity": mo.ACT_IDENTITY}[case["acts"][0]]
    loss = case["loss"]
    kind_o, mask = mo.LOSS_NONE, 0
    if loss.startswith("bernoulli"):
        kind_o = mo.LOSS_BERNOULLI
    elif loss.startswith("gaussian"):
        kind_o = mo.LOSS_GAUSSIAN
    if loss.endswith("_mask"):
        mask = mo.mask_start_from_perc(n_out, case["perc"])
    lspec = mo.LossSpec(kind_o, target, case["var"], mask) if kind_o != mo.LOSS_NONE else mo.LossSpec()
    net = mo.NetSpec(sizes=sizes, acts=[act_o] * len(sizes), W=W, b=b, ecoef=case["ecoef"], has_head=bool(n_out))
    xopt = mo.XOpt(mo.OPT_ADAM if mode["adam"] else mo.OPT_SGD, mode["lr"])
    learn = i % 3 == 0                                                   # a learning call: gradient window + optimizer_p.step()
    acc = list(range(mode["acc_begin"], T)) if learn else []
    seed, step_base, noise_var, lr_p = 40 + i, 1000 * (i + 1), 1.5, 0.05
    noise = (lambda t, l: philox.layer_normals(seed, step_base + t, l, 0, B, sizes[l])) if mode["noise"] else None
    ref = mo.run(net, inputs, X0, lspec, xopt, T, noise=noise, noise_var=noise_var, update_p_at=[T - 1] if learn else [], accumulate_p_at=acc)

    model, lins = gen_golden.build_reference_model(pc, case, W, b, X0, device=device)
    trainer = pc.PCTrainer(model, T=T, update_x_at="all", optimizer_x_fn=torch.optim.Adam if mode["adam"] else torch.optim.SGD,
                           optimizer_x_kwargs={"lr": mode["lr"]}, update_p_at="last" if learn else "never",
                           accumulate_p_at=acc if acc else "never", optimizer_p_fn=torch.optim.SGD, optimizer_p_kwargs={"lr": lr_p},
                           plot_progress_at=[])
    trainer.mcpc_seed = seed
    pt._PHILOX_STEPS[0] = step_base
    loss_fn, loss_kw = gen_golden.reference_loss(um, case, target, device)
    kw = {}
    if mode["noise"]:
        kw = dict(callback_after_t=um.random_step, callback_after_t_kwargs={"_pc_trainer": trainer, "var": noise_var})
    inp = torch.from_numpy(inputs).to(device) if not case["inputs_zero"] else torch.zeros(B, case["n_in"], device=device)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = trainer.train_on_batch(inputs=inp, loss_fn=loss_fn, loss_fn_kwargs=loss_kw, is_log_progress=False,
                                     is_return_results_every_t=True, is_checking_after_callback_after_t=False, **kw)
    assert trainer.last_call_mode == "fused", trainer.last_call_mode
    grp = "facade fuzz (24 random shapes / modes, GPU and staged CPU models), " + ("Adam-x" if mode["adam"] else "SGD-x")
    scale = max(1.0, float(np.abs(ref.overall).max()))
    parity_log.close(grp, "overall[t]", res["overall"], ref.overall, rtol=1e-6, atol=1e-6 * scale)
    parity_log.close(grp, "energy[t]", res["energy"], ref.energy, rtol=1e-6, atol=1e-6 * scale)
    if loss_fn is not None:
        parity_log.close(grp, "loss[t]", res["loss"], ref.loss, rtol=1e-6, atol=1e-6 * scale)
    for l, x in enumerate(trainer.get_model_xs()):
        assert x.device.type == torch.device(device).type
        parity_log.close(grp, "x final", x.detach().cpu().numpy(), ref.xs[l], rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref.xs[l]).max())))
    if learn:
        # param.grad as the reference leaves it (normalised by len(accumulate_p_at) * batch) and the weights after SGD(lr_p)
        for j, lin in enumerate(lins):
            g = lin.weight.grad.detach().cpu().numpy()
            parity_log.close(grp, "param.grad (W)", g, ref.gW[j], rtol=2e-4, atol=2e-5 * max(1e-3, float(np.abs(ref.gW[j]).max())))
            want_w = W[j] - np.float32(lr_p) * ref.gW[j]
            parity_log.close(grp, "W after optimizer_p.step()", lin.weight.detach().cpu().numpy(), want_w, rtol=2e-5,
                             atol=2e-6 * max(1.0, float(np.abs(want_w).max())))
            if lin.bias is not None and ref.gb[j] is not None:
                parity_log.close(grp, "param.grad (b)", lin.bias.grad.detach().cpu().numpy(), ref.gb[j], rtol=2e-4,
                                 atol=1e-4 * max(1e-3, float(np.abs(ref.gb[j]).max())))
