"""The headline call itself, verified end to end ("next round" item 1).

`bench.py` quotes its number on ONE call shape: cfg-M (30-256-256-784 ReLU, Bernoulli read-out), 6000 chains,
T = 5000 = 1000 mixing + 4000 sampling Langevin steps with the fused Philox kick, Hebbian sums over the sampling steps
(reference pc_trainer.py:853-862: `optimizer_p.zero_grad()` at accumulate_p_at[0], autograd adds dF/dtheta of every later
step; :904-914 normalise and step), loss + energies every step, x every 100 steps.  Default tuning = 375 16-chain workgroups on
the round schedule (three launches per cycle, every unit in two of them), 31 + 1 Hebbian segments of 2 x 64 steps through a
three-part spill ring that wraps 10 times, flushes overlapped on two low-priority streams.  This file runs exactly that call and
checks it against

  (1) the same call on the serial tuning (`ws=0,no_overlap=1,slot_cap=128`: the BARRIER kernel -- another program: four waves per
      workgroup, s_barrier hand-overs, generic epilogues, 376 workgroups in hardware rounds -- one launch per segment, one stream,
      one ring part, nothing can race): final state, all 50 records
      and the 276 146-float gradient bucket BITWISE, loss / energies rel 2e-6 (fp32 partial sums are grouped per workgroup);
  (2) an fp64 recomputation of dF/dtheta of every Linear over the 200-step window [4700, 4900) -- late in the call, after
      57 segments, spanning four ring parts -- from the recorded states of that window.  The window's share of the bucket is
      the difference of two more runs of the same call whose accumulation ends at 4900 and at 4700 (the trajectories do not
      depend on what is accumulated: asserted bitwise).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
SIZES, N_OUT, B = [30, 256, 256], 784, 6000
T, MIXING = 5000, 1000
SERIAL_TUNING = "ws=0,no_overlap=1,slot_cap=128"


def _engine(W, b, y, tuning=None):
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=DEV, tuning=tuning)
    eng.bind_params(W, b)
    eng.bind_inputs(None)
    eng.bind_target(y)
    return eng


def _headline_call(eng, xs, acc_end=T, **kw):
    """bench.py's `one_call(learning=True)`, with a fixed Philox step base."""
    from montecarlopredictivecoding_amd import _lib as L
    eng.load_state(xs)
    args = dict(loss_kind=L.LOSS_BERNOULLI, xopt=L.XOPT_SGD, lr=0.03, noise_mode=L.NOISE_PHILOX, noise_var=2.0, seed=30,
                step_base=0, chain_base=0, acc_begin=MIXING, acc_end=acc_end, energy_mode=L.ENERGY_ALL,
                rec_begin=0, rec_stride=100, rec_count=(T + 99) // 100, rec_x=True)
    args.update(kw)
    res = eng.run(T, **args)
    out = [torch.empty_like(x) for x in xs]
    eng.store_state(out)
    flat = eng.read_param_grads_flat(scale=1.0)
    eng.sync_check()
    return res, out, flat


@pytest.fixture(scope="module")
def problem():
    from bench import make_problem
    return make_problem(B, 30, DEV)


@pytest.fixture(scope="module")
def default_run(problem):
    W, b, y, xs = problem
    eng = _engine(W, b, y)
    q = eng.query()
    assert (q["chains_per_wg"], q["n_workgroups"]) == (16, 375) and "round schedule: k=3 " in q["step_kernel"] and "m=2 " in q["step_kernel"]
    res, out, flat = _headline_call(eng, xs)
    assert eng.query()["spill_slots"] == 384
    eng.close()
    return res.energies.cpu().numpy(), out, [r.clone() for r in res.rec_x], flat


def test_headline_call_matches_serial_plain_schedule_bitwise(problem, default_run):
    W, b, y, xs = problem
    en_d, out_d, rec_d, flat_d = default_run
    eng = _engine(W, b, y, tuning=SERIAL_TUNING)
    res, out_s, flat_s = _headline_call(eng, xs)
    assert eng.query()["spill_slots"] == 128 and "mcpc_steps_kernel<1, 4>" in eng.query()["step_kernel"]
    eng.close()
    for a, c in zip(out_d, out_s):
        assert torch.equal(a, c)
    assert len(rec_d) == 3 and rec_d[0].shape[0] == 50
    for a, c in zip(rec_d, res.rec_x):
        assert torch.equal(a, c)
    assert flat_d.numel() == 276146 and torch.equal(flat_d, flat_s)
    assert bool(flat_d[900:].abs().max() > 0) and not bool(flat_d[:900].any())      # dF/dW0 == 0: the pseudo-input is zero
    en_s = res.energies.cpu().numpy()
    assert np.all(np.isfinite(en_d))
    np.testing.assert_allclose(en_d, en_s, rtol=2e-6)
    np.testing.assert_allclose(en_d[:, -1], en_d[:, 0] + en_d[:, 1:4].sum(1), rtol=1e-12)     # overall = loss + energies


def test_headline_call_bucket_of_both_hebbian_kernels_agrees(problem, default_run):
    """The tiled Hebbian GEMM runs the fp16 form by default (mcpc_hebbian.h: mcpc_heb7_kernel; rounds 3-4: bf16x6) and the fp32-MFMA form under
    `heb_fp32=1`: same trajectories bitwise (the flush never writes state), buckets equal to fp32 summation-order noise --
    both carry about 2e-7 of sum|terms| against fp64 (scripts/heb_bf16_ubench.hip), the window test below holds either to it."""
    W, b, y, xs = problem
    _, out_d, _, flat_d = default_run
    eng = _engine(W, b, y, tuning="heb_fp32=1")
    _, out_f, flat_f = _headline_call(eng, xs)
    eng.close()
    for a, c in zip(out_d, out_f):
        assert torch.equal(a, c)
    assert not torch.equal(flat_d, flat_f)            # the key does select another kernel
    off = 0
    for j in range(4):
        for n in (W[j].numel(), b[j].numel()):
            d, f = flat_d[off:off + n].double(), flat_f[off:off + n].double()
            scale = float(f.abs().max())
            if scale > 0:
                assert float((d - f).abs().max()) <= 2e-6 * scale, (j, n, float((d - f).abs().max()) / scale)
            else:
                assert not bool(d.any())
            off += n


@pytest.mark.parametrize("tuning", [None, "heb_fp32=1"], ids=["f16x3", "fp32"])
def test_headline_call_hebbian_window_matches_fp64_recomputation(problem, default_run, tuning):
    W, b, y, xs = problem
    _, out_d, _, _ = default_run
    w0, w1 = 4700, 4900
    eng = _engine(W, b, y, tuning=tuning)
    # accumulation ends at w1; every state of the window is recorded
    res_c, out_c, flat_c = _headline_call(eng, xs, acc_end=w1, rec_begin=w0, rec_stride=1, rec_count=w1 - w0)
    rec = res_c.rec_x
    _, out_e, flat_e = _headline_call(eng, xs, acc_end=w0, rec_count=0)
    eng.close()
    for a, c, d_ in zip(out_d, out_c, out_e):
        assert torch.equal(a, c) and torch.equal(a, d_)               # what is accumulated never changes a trajectory
    got = (flat_c.double() - flat_e.double()).cpu().numpy()
    total = flat_c.double().abs().cpu().numpy()
    # fp64 anchor from the recorded states x_t, t in [w0, w1)  (dF/dW_l = -e_{l+1}^T f(x_l), read-out +e_o^T f(x_3))
    Wd, bd, yd = [w.double() for w in W], [v.double() for v in b], y.double()
    gW = [torch.zeros_like(w) for w in Wd]
    gb = [torch.zeros_like(v) for v in bd]
    aW = [torch.zeros_like(w) for w in Wd]
    ab = [torch.zeros_like(v) for v in bd]
    for k in range(0, w1 - w0, 20):
        x1, x2, x3 = (r[k:k + 20].double() for r in rec)
        f1, f2, f3 = x1.clamp_min(0), x2.clamp_min(0), x3.clamp_min(0)
        e1 = x1 - bd[0]
        e2 = x2 - (f1 @ Wd[1].T + bd[1])
        e3 = x3 - (f2 @ Wd[2].T + bd[2])
        eo = torch.sigmoid(f3 @ Wd[3].T + bd[3]) - yd
        gb[0] -= e1.sum((0, 1))
        ab[0] += e1.abs().sum((0, 1))
        for j, (e, f, sgn) in enumerate(((e2, f1, -1.0), (e3, f2, -1.0), (eo, f3, 1.0)), start=1):
            gW[j] += sgn * torch.einsum("tbu,tbi->ui", e, f)
            gb[j] += sgn * e.sum((0, 1))
            aW[j] += torch.einsum("tbu,tbi->ui", e.abs(), f)
            ab[j] += e.abs().sum((0, 1))
    want = torch.cat([torch.cat([gw.reshape(-1), g_.reshape(-1)]) for gw, g_ in zip(gW, gb)]).cpu().numpy()
    mag = torch.cat([torch.cat([gw.reshape(-1), g_.reshape(-1)]) for gw, g_ in zip(aW, ab)]).cpu().numpy()
    off = 0
    for j in range(4):
        for n in (W[j].numel(), b[j].numel()):
            sl = slice(off, off + n)
            if j == 0 and n == W[0].numel():
                assert not got[sl].any()
            else:
                # the window is a difference of two fp32 sums over 3900 / 3700 steps: half an ulp of the LARGER sum per
                # element (6e-8 |G(4900)|) on top of the 1e-5 of the window's own sum of |terms| the 400-step test allows
                tol = 1e-5 * mag[sl].max() + 2e-7 * total[sl].max()
                np.testing.assert_allclose(got[sl], want[sl], rtol=0, atol=tol)
                if j == 3 and n == W[3].numel():
                    assert np.abs(want[sl]).max() > 50 * tol         # the window is far above that floor: the check has teeth
            off += n
