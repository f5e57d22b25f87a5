"""Seeded sweep of random network shapes and run modes against the NumPy oracle: ragged widths (not multiples of 16), 1-5
latent layers, batches that are not multiples of the chain tile, every activation / loss / x optimizer / noise mode, non-zero
pseudo-inputs, bias-free Linears, masked losses, ragged accumulation windows -- on the kernel form the engine picks by default
(in-place, wave-specialised) and on the barrier kernel.  Complements the golden fixtures (fixed shapes, pinned by the reference itself)."""
import numpy as np
import pytest
import torch

from oracle import mcpc_oracle as mo
from oracle import philox
from oracle.cases import make_case_inputs
from tests import parity_log

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_CASES = 24


def _random_case(i):
    r = np.random.RandomState(1000 + i)
    L_ = int(r.randint(1, 6))
    sizes = [int(r.randint(1, 70)) for _ in range(L_)]
    if i % 5 == 0:
        sizes[-1] = int(r.randint(100, 200))              # a wider last layer now and then (several tiles per wave)
    n_out = 0 if i % 6 == 5 else int(r.randint(1, 120))
    loss = "none" if n_out == 0 else ["bernoulli", "gaussian", "bernoulli_mask", "gaussian_mask", "zero"][i % 5]
    act = ["relu", "tanh", "identity"][i % 3]
    B = int(r.choice([1, 2, 7, 15, 16, 17, 31, 33, 48, 65, 100]))
    T = int(r.randint(3, 12))
    case = dict(sizes=sizes, acts=[act] * L_, ecoef=[float(np.float32(0.5 + r.rand())) for _ in range(L_)], n_in=int(r.randint(1, 20)),
                n_out=n_out, loss=loss, var=float(np.float32(0.3 + r.rand())), perc=float(r.choice([0.25, 0.5, 0.75])), B=B, seed=500 + i,
                x0_range=1.5, inputs_zero=bool(i % 4), no_bias=[0] if i % 7 == 3 else [], calls=[dict(T=T)])
    mode = dict(adam=(i % 4 == 1), noise=(i % 4 != 1) and (i % 3 != 2), acc_begin=int(r.randint(0, T)), lr=float(r.choice([0.01, 0.03, 0.1])))
    return case, mode


@pytest.mark.parametrize("kernel", ["default", "inplace", "barrier"])
@pytest.mark.parametrize("i", range(N_CASES))
def test_random_shape_matches_oracle(i, kernel):
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    case, mode = _random_case(i)
    sizes, n_out, B, T = case["sizes"], case["n_out"], case["B"], case["calls"][0]["T"]
    W, b, X0, inputs, target = make_case_inputs(case)
    act_o = {"relu": mo.ACT_RELU, "tanh": mo.ACT_TANH, "identity": mo.ACT_IDENTITY}[case["acts"][0]]
    act_l = {"relu": L.ACT_RELU, "tanh": L.ACT_TANH, "identity": L.ACT_IDENTITY}[case["acts"][0]]
    loss = case["loss"]
    kind_o, kind_l, mask = mo.LOSS_NONE, L.LOSS_NONE, 0
    if loss.startswith("bernoulli"):
        kind_o, kind_l = mo.LOSS_BERNOULLI, L.LOSS_BERNOULLI
    elif loss.startswith("gaussian"):
        kind_o, kind_l = mo.LOSS_GAUSSIAN, L.LOSS_GAUSSIAN
    if loss.endswith("_mask"):
        mask = mo.mask_start_from_perc(n_out, case["perc"])
    net = mo.NetSpec(sizes=sizes, acts=[act_o] * len(sizes), W=W, b=b, ecoef=case["ecoef"], has_head=bool(n_out))
    lspec = mo.LossSpec(kind_o, target, case["var"], mask) if kind_o != mo.LOSS_NONE else mo.LossSpec()
    xopt = mo.XOpt(mo.OPT_ADAM if mode["adam"] else mo.OPT_SGD, mode["lr"])
    seed = 40 + i
    noise = (lambda t, l: philox.layer_normals(seed, 7 + t, l, 3, B, sizes[l])) if mode["noise"] else None
    ref = mo.run(net, inputs, X0, lspec, xopt, T, noise=noise, noise_var=1.5, accumulate_p_at=list(range(mode["acc_begin"], T)))

    eng = Engine(sizes, [act_l] * len(sizes), case["n_in"], n_out, B, device=DEV, ecoef=case["ecoef"],
                 tuning={"barrier": "ws=0", "inplace": "ws=2", "default": None}[kernel])
    eng.bind_params([torch.from_numpy(w).to(DEV) for w in W], [None if x is None else torch.from_numpy(x).to(DEV) for x in b])
    eng.bind_inputs(None if case["inputs_zero"] else torch.from_numpy(inputs).to(DEV))
    if target is not None:
        eng.bind_target(torch.from_numpy(target).to(DEV))
    xs = [torch.from_numpy(x).to(DEV) for x in X0]
    eng.load_state(xs)
    res = eng.run(T, loss_kind=kind_l, loss_var=case["var"], mask_start=mask, xopt=L.XOPT_ADAM if mode["adam"] else L.XOPT_SGD, lr=mode["lr"],
                  noise_mode=L.NOISE_PHILOX if mode["noise"] else L.NOISE_NONE, noise_var=1.5, seed=seed, step_base=7, chain_base=3,
                  acc_begin=mode["acc_begin"], acc_end=T, energy_mode=L.ENERGY_ALL)
    eng.store_state(xs)
    eng.sync_check()
    en = res.energies.cpu().numpy()
    scale = max(1.0, float(np.abs(ref.overall).max()))
    # tolerances: energies rel 1e-6, states 1e-5 of the largest value (achieved: 8e-8 / 1e-6, profiles/r04_parity_errors.txt)
    grp = "fuzz (24 random shapes / modes), " + ("Adam-x" if mode["adam"] else "SGD-x")
    e_rtol, x_atol = 1e-6, 1e-5
    parity_log.close(grp, "overall[t]", en[:, -1], ref.overall, rtol=e_rtol, atol=1e-6 * scale)
    parity_log.close(grp, "loss[t]", en[:, 0], ref.loss, rtol=e_rtol, atol=1e-6 * scale)
    for l in range(len(sizes)):
        parity_log.close(grp, "x final", xs[l].cpu().numpy(), ref.xs[l], rtol=0, atol=x_atol * max(1.0, float(np.abs(ref.xs[l]).max())))
    flat = eng.read_param_grads_flat().cpu().numpy()
    want = np.concatenate([np.concatenate([gw.reshape(-1)] + ([] if bb is None else [gb.reshape(-1)])) for gw, gb, bb in zip(ref.gW, ref.gb, b)])
    assert flat.shape == want.shape
    parity_log.close(grp, "dF/dtheta bucket", flat, want, rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(want).max())))
    eng.close()


_DEFAULT_FORM_OF_WIDE_NETS = {}        # shape -> step kernel the DEFAULT tuning chose (test_wide_networks_exercise_the_barrier_fallback)


@pytest.mark.parametrize("sizes,n_out", [([40, 384, 200], 784), ([64, 512, 256], 300), ([100, 600, 96], 64), ([256, 256, 256, 256], 1000),
                                         ([16, 500], 0), ([480, 480, 480, 480], 0)])
def test_wide_networks_against_oracle(sizes, n_out):
    """Widths near the limits of the LDS plans: the in-place plan loses its LDS-resident epilogue operands, then
    stops fitting and the engine falls back to the barrier kernel; results must not depend on which form ran."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    B, T, lr, seed = 40, 5, 0.02, 9
    case = dict(sizes=sizes, acts=["relu"] * len(sizes), ecoef=[1.0] * len(sizes), n_in=sizes[0], n_out=n_out, loss="bernoulli" if n_out else "none",
                var=1.0, perc=0.5, B=B, seed=77, x0_range=1.0, calls=[dict(T=T)])
    W, b, X0, inputs, target = make_case_inputs(case)
    net = mo.NetSpec(sizes=sizes, acts=[mo.ACT_RELU] * len(sizes), W=W, b=b, ecoef=case["ecoef"], has_head=bool(n_out))
    lspec = mo.LossSpec(mo.LOSS_BERNOULLI, target) if n_out else mo.LossSpec()
    ref = mo.run(net, inputs, X0, lspec, mo.XOpt(mo.OPT_SGD, lr), T, noise=lambda t, l: philox.layer_normals(seed, t, l, 0, B, sizes[l]),
                 accumulate_p_at=list(range(1, T)))
    seen = set()
    for tuning in (None, "ws=0"):
        eng = Engine(sizes, [L.ACT_RELU] * len(sizes), sizes[0], n_out, B, device=DEV, tuning=tuning)
        q = eng.query()
        seen.add((q["step_kernel"], q["chains_per_wg"]))
        if tuning is None:
            _DEFAULT_FORM_OF_WIDE_NETS[(tuple(sizes), n_out)] = q["step_kernel"]
        eng.bind_params([torch.from_numpy(w).to(DEV) for w in W], [torch.from_numpy(x).to(DEV) for x in b])
        eng.bind_inputs(None)
        if n_out:
            eng.bind_target(torch.from_numpy(target).to(DEV))
        xs = [torch.from_numpy(x).to(DEV) for x in X0]
        eng.load_state(xs)
        res = eng.run(T, loss_kind=L.LOSS_BERNOULLI if n_out else L.LOSS_NONE, lr=lr, noise_mode=L.NOISE_PHILOX, seed=seed, step_base=0,
                      acc_begin=1, acc_end=T, energy_mode=L.ENERGY_ALL)
        eng.store_state(xs)
        eng.sync_check()
        en = res.energies.cpu().numpy()
        parity_log.close("wide networks (5 shapes)", "overall[t]", en[:, -1], ref.overall, rtol=1e-6, err_msg=f"tuning {tuning}: energies\n{en}")
        for l in range(len(sizes)):
            parity_log.close("wide networks (5 shapes)", "x final", xs[l].cpu().numpy(), ref.xs[l], rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref.xs[l]).max())), err_msg=f"tuning {tuning}")
        flat = eng.read_param_grads_flat().cpu().numpy()
        want = np.concatenate([np.concatenate([gw.reshape(-1), gb.reshape(-1)]) for gw, gb in zip(ref.gW, ref.gb)])
        np.testing.assert_allclose(flat, want, rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(want).max())), err_msg=f"tuning {tuning}")
        eng.close()
    assert len(seen) >= 1


def test_wide_networks_exercise_the_barrier_fallback():
    """the shapes above must PROVE the fallback is taken -- under DEFAULT tuning at least one of them does not fit the in-place
    LDS plan and runs on the barrier kernel (and at least one runs in place), not merely 'some kernel ran'."""
    forms = _DEFAULT_FORM_OF_WIDE_NETS
    assert len(forms) >= 6, "run after test_wide_networks_against_oracle (same module)"
    assert any("mcpc_steps_kernel<1, 4>" in k for k in forms.values()), forms
    assert any("mcpc_steps_ws2_kernel" in k for k in forms.values()), forms


@pytest.mark.parametrize("seed", range(6))
def test_random_slicing_of_a_call_is_invisible(seed):
    """mcpc_run executes a slice [t_begin, t_begin + n_steps) of a call: any cut of a call into slices -- through the accumulation
    window, with Adam's step count carried (adam_step0) or the Philox counter indexed by the step of the CALL -- must leave states and
    energies bitwise those of the single run, and the Hebbian sums equal up to the order of the partial sums of the slices."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    r = np.random.RandomState(70 + seed)
    sizes = [int(r.randint(3, 40)) for _ in range(int(r.randint(1, 4)))]
    n_out, B, T = int(r.randint(5, 60)), int(r.choice([9, 16, 40, 70])), int(r.randint(12, 40))
    adam = seed % 2 == 1
    case = dict(sizes=sizes, acts=["tanh"] * len(sizes), ecoef=[1.0] * len(sizes), n_in=sizes[0], n_out=n_out, loss="bernoulli", var=1.0, perc=0.5,
                B=B, seed=900 + seed, x0_range=1.0, calls=[dict(T=T)])
    W, b, X0, inputs, target = make_case_inputs(case)
    acc_begin = int(r.randint(0, T - 2))
    cuts = sorted(set([0, T] + [int(v) for v in r.randint(1, T, size=int(r.randint(1, 5)))]))
    kw = dict(loss_kind=L.LOSS_BERNOULLI, xopt=L.XOPT_ADAM if adam else L.XOPT_SGD, lr=0.05, noise_mode=L.NOISE_NONE if adam else L.NOISE_PHILOX,
              seed=3, step_base=11, acc_begin=acc_begin, acc_end=T, energy_mode=L.ENERGY_ALL)
    outs = []
    for sliced in (False, True):
        eng = Engine(sizes, [L.ACT_TANH] * len(sizes), sizes[0], n_out, B, device=DEV)
        eng.bind_params([torch.from_numpy(w).to(DEV) for w in W], [torch.from_numpy(x).to(DEV) for x in b])
        eng.bind_inputs(None); eng.bind_target(torch.from_numpy(target).to(DEV))
        xs = [torch.from_numpy(x).to(DEV) for x in X0]
        eng.load_state(xs)
        en = torch.zeros(T, L.ENERGY_COLS, dtype=torch.float64, device=DEV)
        if not sliced:
            eng.run(T, energies_out=en, **kw)
        else:
            for t0, t1 in zip(cuts[:-1], cuts[1:]):
                eng.run(T, t_begin=t0, n_steps=t1 - t0, adam_step0=t0, energies_out=en, acc_reset=(t0 <= acc_begin < t1), **kw)
        eng.store_state(xs)
        flat = eng.read_param_grads_flat()
        eng.sync_check()
        outs.append((en.cpu().numpy(), [x.cpu().numpy() for x in xs], flat.cpu().numpy()))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for a, c in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a, c)
    np.testing.assert_allclose(outs[1][2], outs[0][2], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(outs[0][2]).max())))
    assert np.isfinite(outs[0][0]).all()
