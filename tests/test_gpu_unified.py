"""The unified-wave step kernel (csrc/mcpc_steps_u.h; tuning ws=3, and the default wherever its LDS plan fits) against the other two kernel
forms and the oracle.

It runs the lean paths of a call (fused SGD update with or without the Philox kick, Adam without noise) on nets whose whole read-out
error fits the LDS beside the state -- the reference's own 20-128-128-784 (figure_2.py:159-161, figure_3.py:130-132, table_1.py:31-32) --
with the GEMM core and the epilogue arithmetic of the in-place kernel, operation for operation: final states, recorded trajectories and
outputs must be BITWISE those of the in-place kernel (ws=2) and of the barrier kernel (ws=0) whenever those agree among themselves
(a Gaussian read-out's back-projection is scaled per chunk of the read-out, and the three forms cut it differently: rounding-level
differences there, as between the other two); energies regroup fp32 partial sums (8 waves' shares instead of 4) and agree to 2e-6; the
Hebbian sums come from the same spilled images and are bitwise.  A zero loss (utils/model.py:31-33, figure_3.py:125-161) skips the read-out
on the steps that neither record nor spill it: the test records every k-th output and compares with the kernels that compute it every step.
"""
import numpy as np
import pytest
import torch

from oracle import mcpc_oracle as mo
from oracle import philox
from oracle.cases import make_case_inputs
from tests import parity_log

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (sizes, n_out, act, loss, batch)
NETS = [([20, 128, 128], 784, "relu", "bernoulli", 48),
        ([20, 128, 128], 784, "tanh", "gaussian_mask", 33),
        ([20, 128, 128], 784, "relu", "zero", 16),
        ([6, 16, 16], 24, "tanh", "bernoulli_mask", 7),
        ([30, 200, 72], 100, "relu", "bernoulli", 64),       # widths 8 mod 16 / 16 mod 32: the ragged k range of the GEMM core
        ([1], 1, "identity", "gaussian", 300),                # figure_2's linear-Gaussian toy: one latent unit
        ([16, 64, 64, 64], 300, "relu", "bernoulli", 100),
        ([12, 40], 0, "tanh", "none", 20),                    # no read-out Linear at all
        ([160, 96], 520, "relu", "bernoulli", 32)]            # a top layer of more tiles than waves (the e_1 sum is not register-resident)


def _case(sizes, n_out, act, loss, B, seed=321):
    return dict(sizes=sizes, acts=[act] * len(sizes), ecoef=[1.0] * len(sizes), n_in=sizes[0], n_out=n_out, loss=loss, var=0.7, perc=0.5, B=B,
                seed=seed, x0_range=1.0, calls=[dict(T=8)])


def _kinds(case):
    from montecarlopredictivecoding_amd import _lib as L
    loss, n_out = case["loss"], case["n_out"]
    kind_o, kind_l, mask = mo.LOSS_NONE, L.LOSS_NONE, 0
    if loss.startswith("bernoulli"):
        kind_o, kind_l = mo.LOSS_BERNOULLI, L.LOSS_BERNOULLI
    elif loss.startswith("gaussian"):
        kind_o, kind_l = mo.LOSS_GAUSSIAN, L.LOSS_GAUSSIAN
    if loss.endswith("_mask"):
        mask = mo.mask_start_from_perc(n_out, case["perc"])
    return kind_o, kind_l, mask


def _engine(case, W, b, target, tuning):
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    act_l = {"relu": L.ACT_RELU, "tanh": L.ACT_TANH, "identity": L.ACT_IDENTITY}[case["acts"][0]]
    eng = Engine(case["sizes"], [act_l] * len(case["sizes"]), case["n_in"], case["n_out"], case["B"], device=DEV, tuning=tuning)
    eng.bind_params([torch.from_numpy(w).to(DEV) for w in W], [torch.from_numpy(x).to(DEV) for x in b])
    eng.bind_inputs(None)
    if target is not None:
        eng.bind_target(torch.from_numpy(target).to(DEV))
    return eng


def _run(eng, X0, T, **kw):
    xs = [torch.from_numpy(x).to(DEV) for x in X0]
    eng.load_state(xs)
    res = eng.run(T, **kw)
    eng.store_state(xs)
    eng.sync_check()
    return res, [x.cpu().numpy() for x in xs]


@pytest.mark.parametrize("mode", ["mcpc", "pc_sgd", "map_adam"])
@pytest.mark.parametrize("sizes,n_out,act,loss,B", NETS, ids=["-".join(map(str, n[0])) + f"-{n[1]}-{n[2]}-{n[3]}" for n in NETS])
def test_unified_kernel_matches_the_other_forms_and_the_oracle(sizes, n_out, act, loss, B, mode):
    from montecarlopredictivecoding_amd import _lib as L
    case = _case(sizes, n_out, act, loss, B)
    W, b, X0, inputs, target = make_case_inputs(case)
    kind_o, kind_l, mask = _kinds(case)
    T, acc0, seed = 23, 9, 17
    lr = 0.05 if mode == "map_adam" else 0.03
    kw = dict(loss_kind=kind_l, loss_var=case["var"], mask_start=mask, lr=lr, seed=seed, step_base=5, chain_base=2, noise_var=1.3,
              xopt=L.XOPT_ADAM if mode == "map_adam" else L.XOPT_SGD, noise_mode=L.NOISE_PHILOX if mode == "mcpc" else L.NOISE_NONE,
              acc_begin=acc0, acc_end=T, energy_mode=L.ENERGY_ALL, rec_begin=2, rec_stride=5, rec_count=4, rec_x=True, rec_out=n_out > 0)
    outs = {}
    for key, tuning in (("u", "ws=3"), ("inplace", "ws=2"), ("barrier", "ws=0")):
        eng = _engine(case, W, b, target, tuning)
        name = eng.query()["step_kernel"]
        assert ("steps_u_kernel" in name) == (key == "u"), (key, name)
        res, xs = _run(eng, X0, T, **kw)
        outs[key] = dict(en=res.energies.cpu().numpy(), xs=xs, rec=[r.cpu().numpy() for r in res.rec_x],
                         out=res.rec_out.cpu().numpy() if n_out > 0 else None, g=eng.read_param_grads_flat().cpu().numpy())
        eng.close()
    u = outs["u"]
    assert np.isfinite(u["en"]).all()
    others_agree = all(np.array_equal(a, c) for a, c in zip(outs["inplace"]["xs"], outs["barrier"]["xs"]))
    for key in ("inplace", "barrier"):
        o = outs[key]
        for a, c in zip(u["xs"] + u["rec"] + ([u["out"]] if n_out else []), o["xs"] + o["rec"] + ([o["out"]] if n_out else [])):
            if others_agree:
                assert np.array_equal(a, c), (key, float(np.abs(a - c).max()))
            else:
                np.testing.assert_allclose(a, c, rtol=0, atol=2e-6 * max(1.0, float(np.abs(c).max())))
        np.testing.assert_allclose(u["en"], o["en"], rtol=2e-6, atol=1e-9)
        scale = max(float(np.abs(o["g"]).max()), 1e-30)
        np.testing.assert_allclose(u["g"], o["g"], rtol=0, atol=(0.0 if others_agree and key == "inplace" else 2e-6) * scale)
    # ... and the oracle, driven by the NumPy twin of the device Philox
    act_o = {"relu": mo.ACT_RELU, "tanh": mo.ACT_TANH, "identity": mo.ACT_IDENTITY}[act]
    net = mo.NetSpec(sizes=sizes, acts=[act_o] * len(sizes), W=W, b=b, ecoef=case["ecoef"], has_head=bool(n_out))
    lspec = mo.LossSpec(kind_o, target, case["var"], mask) if kind_o != mo.LOSS_NONE else mo.LossSpec()
    noise = (lambda t, l: philox.layer_normals(seed, 5 + t, l, 2, B, sizes[l])) if mode == "mcpc" else None
    ref = mo.run(net, inputs, X0, lspec, mo.XOpt(mo.OPT_ADAM if mode == "map_adam" else mo.OPT_SGD, lr), T, noise=noise, noise_var=1.3,
                 accumulate_p_at=list(range(acc0, T)))
    grp = "unified-wave kernel vs oracle (9 nets x 3 modes)"
    scale = max(1.0, float(np.abs(ref.overall).max()))
    parity_log.close(grp, "overall[t]", u["en"][:, -1], ref.overall, rtol=1e-6, atol=1e-6 * scale)
    parity_log.close(grp, "loss[t]", u["en"][:, 0], ref.loss, rtol=1e-6, atol=1e-6 * scale)
    for l in range(len(sizes)):
        parity_log.close(grp, "x final", u["xs"][l], ref.xs[l], rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref.xs[l]).max())))
    want = np.concatenate([np.concatenate([gw.reshape(-1), gb.reshape(-1)]) for gw, gb in zip(ref.gW, ref.gb)])
    parity_log.close(grp, "dF/dtheta bucket", u["g"], want, rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize("batch", [1, 256, 4100])
def test_zero_loss_read_out_runs_on_recorded_steps_only_and_changes_nothing(batch):
    """BASELINE config 5 (figure_3.py:125-161: loss_fn = zero_fn): the unified-wave kernel skips the read-out on the steps whose output
    nobody sees; the recorded outputs, the trajectories and the final states are those of the kernels that compute it every step."""
    from montecarlopredictivecoding_amd import _lib as L
    case = _case([20, 128, 128], 784, "relu", "zero", batch, seed=5)
    W, b, X0, inputs, target = make_case_inputs(case)
    T = 61
    kw = dict(loss_kind=L.LOSS_NONE, lr=0.1, seed=3, noise_var=2.0, noise_mode=L.NOISE_PHILOX, energy_mode=L.ENERGY_ALL,
              rec_begin=0, rec_stride=20, rec_count=4, rec_x=True, rec_out=True)
    outs = []
    for tuning in ("ws=3", "ws=2"):
        eng = _engine(case, W, b, None, tuning)
        res, xs = _run(eng, X0, T, **kw)
        outs.append((res.energies.cpu().numpy(), xs + [r.cpu().numpy() for r in res.rec_x] + [res.rec_out.cpu().numpy()]))
        eng.close()
    for a, c in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a, c)
    assert np.abs(outs[0][1][-1]).max() > 0                       # the recorded outputs are there
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=2e-6, atol=1e-9)
    assert np.all(outs[0][0][:, 0] == 0)                          # a zero loss


def test_default_tuning_picks_the_unified_kernel_where_its_plan_fits_and_keeps_the_other_kernels_elsewhere():
    from montecarlopredictivecoding_amd import _lib as L
    # the reference's net: unified; cfg-M's net (its e_o + state rows do not fit beside each other): in-place kernel
    for sizes, n_out, want in (([20, 128, 128], 784, "steps_u_kernel"), ([30, 256, 256], 784, "steps_ws2_kernel")):
        case = _case(sizes, n_out, "relu", "bernoulli", 32)
        W, b, X0, inputs, target = make_case_inputs(case)
        eng = _engine(case, W, b, target, None)
        assert want in eng.query()["step_kernel"], eng.query()
        # a run the unified kernel does not serve (injected noise: the golden fixtures' path) still works on such an engine
        g = torch.Generator().manual_seed(1)
        ext = [torch.randn(4, 32, n, generator=g).to(DEV) for n in sizes]
        res, xs = _run(eng, X0, 4, loss_kind=L.LOSS_BERNOULLI, lr=0.03, noise_mode=L.NOISE_EXTERNAL, ext_noise=ext, noise_var=2.0,
                       energy_mode=L.ENERGY_ALL)
        assert np.isfinite(res.energies.cpu().numpy()).all()
        eng.close()
    # a net whose whole read-out error and state rows do not fit the LDS together has no unified plan: ws=3 says so
    wide = _case([256, 256, 256, 256], 1000, "relu", "bernoulli", 32)
    Ww, bw, _, _, tw = make_case_inputs(wide)
    with pytest.raises(Exception, match="unified-wave"):
        _engine(wide, Ww, bw, tw, "ws=3")


@pytest.mark.parametrize("loss", ["bernoulli", "zero"])
def test_cfg_m_width_on_the_unified_kernel(loss):
    """cfg-M's net (30-256-256-784): the unified plan fits the 160 KiB only with the bit-packed target rows left in global memory (the
    read-out's epilogue requests its words in front of the row's GEMM).  The automatic choice keeps the in-place kernel for its MCPC /
    MAP calls (profiles/r06_small_net.txt) and takes the unified kernel for a ZERO-LOSS call, whose read-out only that kernel skips."""
    from montecarlopredictivecoding_amd import _lib as L
    case = _case([30, 256, 256], 784, "relu", loss, 80, seed=12)
    W, b, X0, inputs, target = make_case_inputs(case)
    kind = L.LOSS_BERNOULLI if loss == "bernoulli" else L.LOSS_NONE
    T = 19
    kw = dict(loss_kind=kind, lr=0.03, seed=9, step_base=0, noise_var=2.0, noise_mode=L.NOISE_PHILOX, energy_mode=L.ENERGY_ALL, acc_begin=7, acc_end=T if loss == "bernoulli" else 7,
              rec_begin=3, rec_stride=6, rec_count=3, rec_x=True, rec_out=True)
    outs = []
    for tuning in ("ws=3", "ws=2", None):
        eng = _engine(case, W, b, target if loss == "bernoulli" else None, tuning)
        assert ("steps_u_kernel" in eng.query()["step_kernel"]) == (tuning == "ws=3")
        res, xs = _run(eng, X0, T, **kw)
        outs.append((res.energies.cpu().numpy(), xs + [r.cpu().numpy() for r in res.rec_x] + [res.rec_out.cpu().numpy()], eng.read_param_grads_flat().cpu().numpy()))
        eng.close()
    for o in outs[1:]:
        for a, c in zip(outs[0][1], o[1]):
            assert np.array_equal(a, c)
        np.testing.assert_allclose(outs[0][0], o[0], rtol=2e-6, atol=1e-9)
        assert np.array_equal(outs[0][2], o[2])


def test_unified_kernel_on_the_round_schedule_and_sliced_calls():
    """More 16-chain units than CUs (the round schedule's per-launch unit lists) and a call cut into launches at arbitrary steps."""
    from montecarlopredictivecoding_amd import _lib as L
    case = _case([20, 128, 128], 784, "relu", "bernoulli", 4400, seed=8)
    W, b, X0, inputs, target = make_case_inputs(case)
    T = 45
    kw = dict(loss_kind=L.LOSS_BERNOULLI, lr=0.03, seed=9, step_base=0, noise_var=2.0, noise_mode=L.NOISE_PHILOX, energy_mode=L.ENERGY_ALL, acc_begin=20, acc_end=T)
    outs = []
    for tuning in ("ws=3", "ws=3,rr=0", "ws=2"):
        eng = _engine(case, W, b, target, tuning)
        if tuning == "ws=3":
            assert "round schedule" in eng.query()["step_kernel"] and "steps_u_kernel" in eng.query()["step_kernel"]
        res, xs = _run(eng, X0, T, **kw)
        outs.append((res.energies.cpu().numpy(), xs, eng.read_param_grads_flat().cpu().numpy()))
        eng.close()
    for o in outs[1:]:
        for a, c in zip(outs[0][1], o[1]):
            assert np.array_equal(a, c)
        np.testing.assert_allclose(outs[0][0], o[0], rtol=2e-6)
        np.testing.assert_allclose(outs[0][2], o[2], rtol=0, atol=2e-6 * float(np.abs(o[2]).max()))
    # the same call in three slices (t_begin / n_steps) on the unified kernel
    eng = _engine(case, W, b, target, "ws=3")
    xs = [torch.from_numpy(x).to(DEV) for x in X0]
    eng.load_state(xs)
    t = 0
    for n in (7, 21, 17):
        eng.run(T, t_begin=t, n_steps=n, **kw)
        t += n
    eng.store_state(xs)
    eng.sync_check()
    for a, c in zip(outs[0][1], [x.cpu().numpy() for x in xs]):
        assert np.array_equal(a, c)
    eng.close()


@pytest.mark.parametrize("y_lo,y_hi", [(-1.0, 2.0), (0.0, 255.0), (-40.0, 3.0)], ids=["bounded", "pixels-0-255", "negative"])
@pytest.mark.parametrize("tuning", [None, "ws=2", "ws=0"], ids=["default", "inplace", "barrier"])
def test_bernoulli_read_out_with_targets_outside_the_unit_interval(tuning, y_lo, y_hi):
    """the reference's BCEWithLogitsLoss (utils/model.py:17-22) takes any target and stays finite; the back-projection of the
    Bernoulli read-out's error used a CONSTANT fp16 scale that assumes |sigmoid(o) - y| <= 2 and overflowed to Inf from |y| ~ 8 on
    (un-normalised 0..255 pixels).  mcpc_bind_target now records whether the target lies in [-1, 2]; outside, the rows scale by their own
    maximum like every unbounded read-out error.  Every kernel form against the oracle, a learning call."""
    from montecarlopredictivecoding_amd import _lib as L
    case = _case([20, 128, 128], 784, "relu", "bernoulli", 40, seed=77)
    W, b, X0, inputs, target = make_case_inputs(case)
    r = np.random.RandomState(4)
    target = (y_lo + (y_hi - y_lo) * r.rand(*target.shape)).astype(np.float32)
    T, acc0, seed, lr = 12, 4, 21, 0.002
    eng = _engine(case, W, b, target, tuning)
    res, xs = _run(eng, X0, T, loss_kind=L.LOSS_BERNOULLI, lr=lr, seed=seed, step_base=0, noise_var=1.0, noise_mode=L.NOISE_PHILOX,
                   acc_begin=acc0, acc_end=T, energy_mode=L.ENERGY_ALL)
    g = eng.read_param_grads_flat().cpu().numpy()
    eng.close()
    en = res.energies.cpu().numpy()
    assert np.isfinite(en).all() and all(np.isfinite(x).all() for x in xs) and np.isfinite(g).all()
    net = mo.NetSpec(sizes=case["sizes"], acts=[mo.ACT_RELU] * 3, W=W, b=b, ecoef=case["ecoef"], has_head=True)
    ref = mo.run(net, inputs, X0, mo.LossSpec(mo.LOSS_BERNOULLI, target, 1.0, 0), mo.XOpt(mo.OPT_SGD, lr), T,
                 noise=lambda t, l: philox.layer_normals(seed, t, l, 0, 40, case["sizes"][l]), noise_var=1.0, accumulate_p_at=list(range(acc0, T)))
    grp = "Bernoulli read-out, targets outside [0, 1] (3 ranges x 3 kernels)"
    scale = max(1.0, float(np.abs(ref.overall).max()))
    parity_log.close(grp, "overall[t]", en[:, -1], ref.overall, rtol=1e-6, atol=1e-6 * scale)
    for l in range(3):
        parity_log.close(grp, "x final", xs[l], ref.xs[l], rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref.xs[l]).max())))
    want = np.concatenate([np.concatenate([gw.reshape(-1), gb.reshape(-1)]) for gw, gb in zip(ref.gW, ref.gb)])
    parity_log.close(grp, "dF/dtheta bucket", g, want, rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(want).max())))
