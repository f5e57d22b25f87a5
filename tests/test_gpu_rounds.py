"""The round schedule (csrc/mcpc_api.hip: setup_rounds / run_round_cycle): a shard of more 16-chain units than CUs runs as k launches
per cycle, every unit in m of them, instead of ceil(U / CUs) hardware rounds.

Chains are independent and every schedule runs the same per-chain arithmetic, so against
  * the hardware rounds of the same kernel (`rr=0`) and
  * the barrier kernel (`ws=0`: another program, generic epilogues, hardware rounds of two workgroups per CU)
final states and records must be BITWISE equal, energies equal up to the regrouping of fp32 partial sums, and the Hebbian sums
bitwise (every unit has filled its own rows of a ring part before the part is flushed; the running sum of e_1 is one sequential
chain per element whatever the launches, mcpc_ws2_lean.h: lean_load_e0).
"""
import re

import numpy as np
import pytest
import torch

from tests.test_gpu_fullsize import B, DEV, SIZES, _engine, _problem, _run

pytestmark = pytest.mark.gpu
HW_ROUNDS = "rr=0"
BARRIER = "ws=0"


def _plan(units, n_cu=256):
    """Mirror of setup_rounds: per k <= 16 the largest m whose launches of m consecutive groups fit the CUs; the smallest k within
    3 % of the best k / m."""
    fit = {}
    for k in range(2, 17):
        sizes = [(g + 1) * units // k - g * units // k for g in range(k)]
        for m in range(k - 1, 0, -1):
            if max(sum(sizes[(i + j) % k] for j in range(m)) for i in range(k)) <= n_cu:
                fit[k] = m
                break
    if not fit:
        return ((units + n_cu - 1) // n_cu, 1)
    bk = min(fit, key=lambda k: (k / fit[k], k))
    for k in sorted(fit):
        if k < bk and 100.0 * k * fit[bk] <= 103.0 * bk * fit[k]:
            return (k, fit[k])
    return (bk, fit[bk])


def _km(eng):
    q = eng.query()
    mt = re.search(r"round schedule: k=(\d+) .* m=(\d+)", q["step_kernel"])
    return (int(mt.group(1)), int(mt.group(2))) if mt else None


def test_plan_mirror_of_known_cases():
    assert _plan(375) == (3, 2)          # 6000 chains: 250 workgroups per launch
    assert _plan(512) == (2, 1) and _plan(1024) == (4, 1)
    assert _plan(300) == (6, 5)          # 1.200 launch times per step; (13, 11) would be 1.182 with launches of 5 steps in a Hebbian segment
    assert _plan(258) == (12, 11) and _plan(3000) == (12, 1)


@pytest.mark.parametrize("batch,T,acc0", [(6000, 331, 97), (4800, 140, 33), (4112, 90, 20), (8192, 150, 61), (12000, 70, 30),
                                          (4097, 41, 11), (5555, 41, 11), (9999, 41, 11)])
def test_round_schedule_matches_other_schedules_bitwise(batch, T, acc0):
    """A learning call (inference stretch, then Hebbian segments through the spill ring) on the default plan against the hardware
    rounds of the same kernel and against the barrier kernel.  The last three sizes end in a partly filled workgroup (1, 3 and 15
    chains of 16) and one that is all padding (4097: Bpad = 4128)."""
    W, b, y, xs = _problem(batch)
    outs = {}
    for key, tuning in (("rounds", None), ("hw", HW_ROUNDS), ("barrier", BARRIER)):
        eng = _engine(batch, W, b, y, tuning=tuning)
        q = eng.query()
        if key == "rounds":
            assert q["chains_per_wg"] == 16 and q["n_workgroups"] == (batch + 15) // 16 and _km(eng) == _plan(q["n_workgroups"]), q
            m = _km(eng)[1]
        elif key == "hw":
            assert q["chains_per_wg"] == 16 and _km(eng) is None
        else:
            assert _km(eng) is None
        res, out = _run(eng, xs, T, acc_begin=acc0, acc_end=T, rec_begin=0, rec_stride=37, rec_count=(T + 36) // 37, rec_x=True)
        outs[key] = (res.energies.cpu().numpy(), [o.cpu().numpy() for o in out], [r.cpu().numpy() for r in res.rec_x],
                     eng.read_param_grads_flat().cpu().numpy())
        eng.close()
    en = outs["rounds"][0]
    assert np.all(np.isfinite(en))
    np.testing.assert_allclose(en[:, -1], en[:, 0] + en[:, 1:4].sum(1), rtol=1e-12)
    for key in ("hw", "barrier"):
        for a, c in zip(outs["rounds"][1] + outs["rounds"][2], outs[key][1] + outs[key][2]):
            assert np.array_equal(a, c), key
        np.testing.assert_allclose(en, outs[key][0], rtol=2e-6)
        if 128 % m == 0 and key == "hw":       # Hebbian segments of 128 steps in both schedules: the flushes add the same partial sums in the same order
            assert np.array_equal(outs["rounds"][3], outs[key][3]), key
        else:                  # segments of m * (128 // m) steps: other partial sums of the same fp32 terms
            scale = np.abs(outs[key][3]).max()
            np.testing.assert_allclose(outs["rounds"][3], outs[key][3], rtol=0, atol=2e-6 * scale)
    assert np.abs(outs["rounds"][3]).max() > 0


@pytest.mark.parametrize("mode", ["adam", "external_noise"])
def test_round_schedule_with_per_step_tables(mode):
    """Adam's bias-correction table and injected normals are indexed by the step, which differs between the units of a launch of
    the round schedule (a unit starts where its previous launch left it)."""
    from montecarlopredictivecoding_amd import _lib as L
    W, b, y, xs = _problem()
    xs_small = [x * 0.1 for x in xs]
    T = 57
    kw = dict(noise_mode=L.NOISE_NONE, xopt=L.XOPT_ADAM, lr=0.05)
    if mode == "external_noise":
        g = torch.Generator().manual_seed(3)
        ext = [torch.randn(T, B, n, generator=g).to(DEV) for n in SIZES]
        kw = dict(noise_mode=L.NOISE_EXTERNAL, ext_noise=ext, noise_var=2.0, lr=0.03)
    outs = []
    for tuning in (None, "rr_qmax=5", HW_ROUNDS):
        eng = _engine(B, W, b, y, tuning=tuning)
        res, out = _run(eng, xs_small, T, **kw)
        extra = []
        if mode == "adam":
            m = [torch.empty_like(x) for x in xs]; v = [torch.empty_like(x) for x in xs]
            eng.store_adam_state(m, v)
            eng.sync_check()
            extra = [t.cpu().numpy() for t in m + v]
        outs.append((res.energies.cpu().numpy(), [o.cpu().numpy() for o in out] + extra))
        eng.close()
    for other in outs[1:]:
        for a, c in zip(outs[0][1], other[1]):
            assert np.array_equal(a, c)
        np.testing.assert_allclose(outs[0][0], other[0], rtol=2e-6)
    assert np.all(np.isfinite(outs[0][0]))


def test_round_schedule_sliced_calls_continue_each_other():
    """Two calls of 40 + 53 steps (the second starts at t_begin = 40 of T = 93) against one call of 93: the cycles of a call start
    at its own first step, so slices of any length compose."""
    from montecarlopredictivecoding_amd import _lib as L
    W, b, y, xs = _problem()
    eng = _engine(B, W, b, y)
    res, out_one = _run(eng, xs, 93)
    en_one = res.energies.cpu().numpy()
    eng.load_state(xs)
    args = dict(loss_kind=L.LOSS_BERNOULLI, lr=0.03, noise_mode=L.NOISE_PHILOX, seed=77, step_base=1000, energy_mode=L.ENERGY_ALL)
    r1 = eng.run(93, t_begin=0, n_steps=40, **args)
    r2 = eng.run(93, t_begin=40, n_steps=53, **args)
    out_two = [torch.empty_like(x) for x in xs]
    eng.store_state(out_two)
    eng.sync_check()
    en_two = np.concatenate([r1.energies.cpu().numpy()[:40], r2.energies.cpu().numpy()[40:]])
    eng.close()
    for a, c in zip(out_one, out_two):
        assert torch.equal(a, c)
    np.testing.assert_array_equal(en_one, en_two)


@pytest.mark.parametrize("sizes,act,loss,xopt", [([30, 256, 256], "relu", "bernoulli", "sgd_noise"), ([20, 128, 128], "tanh", "gaussian", "sgd_noise"),
                                                 ([30, 256, 256], "relu", "bernoulli", "adam"), ([24, 96], "tanh", "gaussian", "sgd"),
                                                 ([20, 128, 128], "relu", "none", "sgd_noise")])
def test_epilogue_operands_in_lds_match_global_memory(sizes, act, loss, xopt):
    """16-chain plans with the room keep the state rows, biases, mu_1 rows and bit-packed target rows of a workgroup's chains in LDS
    for the whole launch (KParams::xl, mcpc_ws2_lean.h); `no_xl=1` leaves them in global memory.  Same arithmetic either way: states,
    records, energies, Hebbian sums and Adam moments bitwise equal -- 0/1 targets (bit words from LDS), fp32 targets (still read from
    global memory), no loss, every lean x update (SGD, SGD + Philox kick, Adam), a partially filled last workgroup (lean epilogues
    with per-lane masks for its padding chains) and a call cut into launches by a Hebbian window.  `no_lean=1` runs the generic
    epilogues everywhere: the same bits again, energies included."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    batch, T, n_out = 1000, 45, 784 if len(sizes) == 3 else 40            # 1000 = 62 full workgroups + one of 8 chains
    g = torch.Generator().manual_seed(11)
    dims = [sizes[0]] + sizes + [n_out]
    W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) / dims[j] ** 0.5).to(DEV) for j in range(len(dims) - 1)]
    b = [((torch.rand(dims[j + 1], generator=g) * 2 - 1) / dims[j] ** 0.5).to(DEV) for j in range(len(dims) - 1)]
    y = (torch.rand(batch, n_out, generator=g) < 0.2).float().to(DEV) if loss == "bernoulli" else torch.rand(batch, n_out, generator=g).to(DEV)
    xs = [((torch.rand(batch, n, generator=g) * 2 - 1) * 0.5).to(DEV) for n in sizes]
    inputs = torch.rand(batch, sizes[0], generator=g).to(DEV)
    a_dev = L.ACT_TANH if act == "tanh" else L.ACT_RELU
    kw = dict(loss_kind={"bernoulli": L.LOSS_BERNOULLI, "gaussian": L.LOSS_GAUSSIAN, "none": L.LOSS_NONE}[loss], energy_mode=L.ENERGY_ALL,
              seed=5, step_base=7, acc_begin=13, acc_end=T, rec_begin=0, rec_stride=9, rec_count=5, rec_x=True)
    if loss == "gaussian":
        kw.update(loss_var=0.7, mask_start=3)
    if xopt == "adam":
        kw.update(xopt=L.XOPT_ADAM, lr=0.05, noise_mode=L.NOISE_NONE)
    else:
        kw.update(lr=0.03, noise_mode=L.NOISE_PHILOX if xopt == "sgd_noise" else L.NOISE_NONE, noise_var=1.5)
    outs = []
    for tuning in ("ws=2", "ws=2,no_xl=1", "ws=2,no_lean=1"):      # (ws=2: the in-place kernel also where the unified-wave kernel would serve the call)
        eng = Engine(sizes, [a_dev] * len(sizes), sizes[0], n_out, batch, device=DEV, tuning=tuning)
        assert eng.query()["chains_per_wg"] == 16
        lds = eng.query()["lds_bytes"]
        eng.bind_params(W, b); eng.bind_inputs(inputs); eng.bind_target(y)
        eng.load_state(xs)
        res = eng.run(T, **kw)
        out = [torch.empty_like(x) for x in xs]
        eng.store_state(out)
        extra = []
        if xopt == "adam":
            m = [torch.empty_like(x) for x in xs]; v = [torch.empty_like(x) for x in xs]
            eng.store_adam_state(m, v)
            extra = m + v
        flat = eng.read_param_grads_flat(scale=1.0)
        eng.sync_check()
        outs.append((lds, [t.cpu().numpy() for t in out + list(res.rec_x) + extra + [flat]], res.energies.cpu().numpy()))
        eng.close()
    assert outs[0][0] > outs[1][0]                      # the default plan did take the extra LDS
    for other in outs[1:]:
        for a, c in zip(outs[0][1], other[1]):
            assert np.array_equal(a, c)
        assert np.array_equal(outs[0][2], other[2])
    assert np.all(np.isfinite(outs[0][2]))
    assert np.abs(outs[0][1][-1]).max() > 0
