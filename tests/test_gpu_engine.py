"""GPU parity: the HIP engine (through the C ABI) vs the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from oracle import philox
from tests import parity_log
from tests.golden_util import Golden, fixture_names

pytestmark = pytest.mark.gpu

ACT = {"identity": 0, "relu": 1, "tanh": 2}


def _dev():
    return torch.device("cuda", 0)


def make_engine(g, **kw):
    from montecarlopredictivecoding_amd.engine import Engine
    c = g.case
    return Engine(c["sizes"], [ACT[a] for a in c["acts"]], c["n_in"], c["n_out"], c["B"], device=_dev(),
                  ecoef=c["ecoef"], **kw)


def bind(eng, g, W=None, b=None):
    dev = _dev()
    W = g.W if W is None else W
    b = g.b if b is None else b
    Wt = [torch.from_numpy(np.ascontiguousarray(w)).to(dev) for w in W]
    bt = [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in b]
    eng.bind_params(Wt, bt)
    inputs = None if not g.inputs.any() else torch.from_numpy(g.inputs).to(dev)
    eng.bind_inputs(inputs)
    if g.target is not None:
        eng.bind_target(torch.from_numpy(g.target).to(dev))
    return Wt, bt


def run_call(eng, g, ci, call, acc_window=None, **kw):
    """One golden call on the engine with the SAME injected normals that drove the reference."""
    from montecarlopredictivecoding_amd import _lib as L
    dev = _dev()
    c = g.case
    T = call["T"]
    ls = g.loss_spec()
    noise = g.noise(ci)
    ext = None
    if noise is not None:
        ext = [torch.from_numpy(np.stack([noise(t, l) for t in range(T)])).to(dev) for l in range(len(c["sizes"]))]
    up, acc = g.schedules(call)
    if acc_window is None:
        if acc:
            acc_window = (acc[0], T)          # autograd keeps adding until the call ends (pc_trainer.py:853-862)
        elif up:
            acc_window = (up[-1], up[-1] + 1)
        else:
            acc_window = (0, T)               # .grad is never zeroed: sum over the whole call
    res = eng.run(T, loss_kind=ls.kind, loss_var=ls.var, mask_start=ls.mask_start,
                  xopt=L.XOPT_ADAM if call["xopt"] == "adam" else L.XOPT_SGD, lr=call["lr"],
                  noise_mode=L.NOISE_EXTERNAL if ext is not None else L.NOISE_NONE,
                  noise_var=call.get("noise_var", 2.0), ext_noise=ext,
                  acc_begin=acc_window[0], acc_end=acc_window[1],
                  # .grad is only zeroed by the schedules (pc_trainer.py:853-859): with neither an update
                  # nor an accumulate list, a later call keeps adding to what the earlier call left
                  acc_reset=not (ci > 0 and not acc and not up),
                  energy_mode=L.ENERGY_ALL, rec_begin=0, rec_stride=1, rec_count=T, rec_x=True, rec_out=True, **kw)
    return res, (up, acc)


def check_against_golden(g, ci, call, res, eng, xs_final, sched, x_atol=None, e_rtol=None, group=None):
    """Tolerances: BASELINE.md section 3's contract -- energies rel 1e-6, states 1e-5 absolute -- against the reference's own fp32
    trajectories (50-100 steps).  Achieved on an MI355X (profiles/r04_parity_errors.txt): energies 2.1e-7 (SGD-x) / 6.6e-7 (Adam-x),
    states 1.4e-6 / 1.9e-6, read-out 4.8e-6.  Round 5: Adam-x energies at 1e-6 as well -- the kernel's Adam update is operation for
    operation torch.optim.Adam's (correctly rounded square root and divisions, csrc/mcpc_device.h: adam_x); rounds 1-4 used
    v_rcp_f32 / v_sqrt_f32 and held them to 3e-6."""
    adam = call["xopt"] == "adam"
    if x_atol is None:
        x_atol = 1e-5
    if e_rtol is None:
        e_rtol = 1e-6
    group = group or ("fixtures, Adam-x" if adam else "fixtures, SGD-x")
    nc = g.case.get("rec_chains", None)
    L_ = len(g.case["sizes"])
    en = res.energies.cpu().numpy()
    parity_log.close(group, "energy[t]", en[:, 1:1 + L_].sum(1), g.get(ci, "energy"), rtol=e_rtol, atol=1e-6)
    parity_log.close(group, "loss[t]", en[:, 0], g.get(ci, "loss"), rtol=e_rtol, atol=1e-6)
    parity_log.close(group, "overall[t]", en[:, -1], g.get(ci, "overall"), rtol=e_rtol, atol=1e-6)
    for t in call.get("record_at", []):
        for l in range(L_):
            parity_log.close(group, "x[t] (records)", res.rec_x[l][t].cpu().numpy()[:nc], g.get(ci, f"x_t{t}_l{l}"), rtol=0, atol=x_atol)
        if g.case["n_out"]:
            parity_log.close(group, "outputs[t]", res.rec_out[t].cpu().numpy()[:nc], g.get(ci, f"out_t{t}"), rtol=0, atol=3 * x_atol)
    for l in range(L_):
        parity_log.close(group, "x final", xs_final[l].cpu().numpy()[:nc], g.get(ci, f"x_final_l{l}"), rtol=0, atol=x_atol)
    # parameter gradients as the reference leaves them in .grad
    up, acc = sched
    B = g.case["B"]
    scale = 1.0
    if up:
        scale = 1.0 / (len(acc) * B) if acc else 1.0 / B
    for j in range(eng.n_lin):
        n_out, n_in = eng.lin_shape(j)
        dW = torch.empty(n_out, n_in, device=_dev())
        db = torch.empty(n_out, device=_dev())
        eng.read_param_grads(j, dW, db, scale=scale)
        dW, db = dW.cpu().numpy(), db.cpu().numpy()
        if g.has(ci, f"gW{j}"):
            ref = g.get(ci, f"gW{j}")
            parity_log.close(group, "dF/dW", dW, ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))
        if g.has(ci, f"gb{j}"):
            ref = g.get(ci, f"gb{j}")
            parity_log.close(group, "dF/db", db, ref, rtol=2e-4, atol=1e-4 * max(1.0, np.abs(ref).max()))
        if g.has(ci, f"gW{j}_idx"):
            idx, ref = g.get(ci, f"gW{j}_idx"), g.get(ci, f"gW{j}_val")
            scale_abs = g.get(ci, f"gW{j}_abs") / dW.size
            parity_log.close(group, "dF/dW (sampled entries)", dW.reshape(-1)[idx], ref, rtol=2e-3, atol=2e-3 * scale_abs)
            parity_log.close(group, "sum |dF/dW|", np.abs(dW.astype(np.float64)).sum(), g.get(ci, f"gW{j}_abs"), rtol=2e-4)


@pytest.mark.parametrize("kernel", ["default", "barrier"])
@pytest.mark.parametrize("name", fixture_names(exclude=("g9_",)))
def test_engine_matches_reference_golden(name, kernel, monkeypatch):
    """Every golden fixture on the default kernel (in-place wave-specialised: the kernel the benchmark runs) AND on the barrier
    kernel that remains the fallback and the parity checks' independent form (tuning ws=0): every loss / optimizer / noise / schedule
    variant the reference's fixtures hold is pinned on both."""
    if kernel == "barrier":
        monkeypatch.setenv("MCPC_TUNING", "ws=0")
    g = Golden(name)
    eng = make_engine(g)
    assert ("mcpc_steps_kernel<1, 4>" in eng.query()["step_kernel"]) == (kernel == "barrier")
    W, b = g.W, g.b
    keep = bind(eng, g)
    dev = _dev()
    xs = [torch.from_numpy(x).to(dev) for x in g.X0]
    for ci, call in enumerate(g.case["calls"]):
        if call.get("sample_x", True):
            xs = [torch.from_numpy(x).to(dev) for x in g.X0]
        eng.load_state(xs)
        res, sched = run_call(eng, g, ci, call)
        xs = [torch.empty_like(x) for x in xs]
        eng.store_state(xs)
        torch.cuda.synchronize()
        check_against_golden(g, ci, call, res, eng, xs, sched)
        if sched[0]:   # the reference's optimizer_p changed the parameters: continue from its values
            W = [g.get(ci, f"W{j}_after") for j in range(len(W))]
            b = [g.get(ci, f"b{j}_after") if bb is not None else None for j, bb in enumerate(b)]
            keep = bind(eng, g, W, b)   # noqa: F841
    eng.close()


def _per_slot_bytes(case):
    """Bytes of one spill-ring slot (mcpc_create: activations of every latent layer, errors of layers >= 2, read-out error)."""
    pad16 = lambda n: (n + 15) // 16 * 16   # noqa: E731
    bpad = (case["B"] + 31) // 32 * 32
    floats = sum(pad16(n) * (2 if l >= 1 else 1) for l, n in enumerate(case["sizes"])) + pad16(case["n_out"])
    return bpad * floats * 4


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("slots,parts", [(2, 3), (3, 3), (6, 3), (4, 2), (8, 2), (8, 4)])
@pytest.mark.parametrize("name", ["g2_cfgM_b64", "g8_ragged", "g6_nonzero_inputs"])
def test_hebbian_ring_reuse_matches_reference_golden(name, slots, parts, overlap):
    """The learning call at T = 5000 re-uses each half of the Hebbian spill ring dozens of times, with the flush of one half
    overlapped with the steps that fill the other (pc_trainer.py:853-862,904-914 is what the sums must equal).  Here the ring
    is shrunk through mcpc_net_desc::spill_budget_bytes to 2 ... 8 slots in 2, 3 (the default) or 4 parts, so the 80 / 15 / 8
    accumulating steps of the fixtures wrap it up to 40 times -- with the overlapped flush (parts flushed one by one on the
    low-priority stream behind ev_flush waits; 2 slots: two halves of one) and with the serial one -- and dW / db must still be the
    reference's.  g2_cfgM_b64 runs the LDS-tiled Hebbian kernel (784 x 256 and
    256 x 256) plus the streaming one (256 x 32); the small nets the streaming kernel only."""
    g = Golden(name)
    eng = make_engine(g, spill_budget_bytes=slots * _per_slot_bytes(g.case), tuning=f"ring_parts={parts}" + ("" if overlap else ",no_overlap=1"))
    assert eng.query()["spill_slots"] == slots
    bind(eng, g)
    dev = _dev()
    xs = [torch.from_numpy(x).to(dev) for x in g.X0]
    call = g.case["calls"][0]
    eng.load_state(xs)
    res, sched = run_call(eng, g, 0, call)
    xs = [torch.empty_like(x) for x in xs]
    eng.store_state(xs)
    eng.sync_check()
    check_against_golden(g, 0, call, res, eng, xs, sched)
    eng.close()


def test_philox_bits_match_numpy():
    from montecarlopredictivecoding_amd.engine import philox_normals
    for (seed, step, layer, base, B, n) in [(0, 0, 0, 0, 7, 5), (0x123456789ABCDEF, 2 ** 33 + 5, 2, 4000, 64, 257),
                                            (30, 4999, 1, 0, 129, 256)]:
        raw = philox_normals(seed, step, layer, base, B, n, _dev(), raw=True).cpu().numpy().view(np.uint32)
        ref = philox.layer_u32(seed, step, layer, base, B, n)
        assert np.array_equal(raw, ref)
        z = philox_normals(seed, step, layer, base, B, n, _dev()).cpu().numpy()
        zr = philox.layer_normals(seed, step, layer, base, B, n)
        np.testing.assert_allclose(z, zr, rtol=0, atol=2e-5)


def test_philox_statistics():
    from montecarlopredictivecoding_amd.engine import philox_normals
    z = philox_normals(99, 12345, 0, 0, 8192, 512, _dev()).double().cpu().numpy().reshape(-1)
    n = z.size
    assert abs(z.mean()) < 5.0 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 5.0 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 5.0 * np.sqrt(15.0 / n)
    assert abs((z ** 4).mean() - 3.0) < 5.0 * np.sqrt(96.0 / n)
    assert np.abs(z).max() < 6.5


def test_two_engines_on_two_streams_run_concurrently():
    """include/mcpc.h threading rule: one engine per (device, stream), launches asynchronous.  Two engines of different shapes
    driven from two torch streams at the same time (their kernels overlap on the GPU: each leaves most CUs idle) must give
    exactly what each gives alone."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    dev = _dev()
    specs = [([12, 40, 24], 8 + 4, 60, 200), ([6, 20], 6, 0, 90)]      # (sizes, n_in, n_out, batch)
    gen = torch.Generator().manual_seed(11)

    def make(sizes, n_in, n_out, batch):
        dims = [n_in] + sizes + ([n_out] if n_out else [])
        W = [((torch.rand(dims[j + 1], dims[j], generator=gen) - 0.5) * 0.6).to(dev) for j in range(len(dims) - 1)]
        b = [((torch.rand(dims[j + 1], generator=gen) - 0.5) * 0.2).to(dev) for j in range(len(dims) - 1)]
        y = (torch.rand(batch, n_out, generator=gen) < 0.3).float().to(dev) if n_out else None
        xs = [torch.rand(batch, n, generator=gen).to(dev) for n in sizes]
        return W, b, y, xs

    probs = [make(*s) for s in specs]

    def launch(i, stream):
        sizes, n_in, n_out, batch = specs[i]
        W, b, y, xs = probs[i]
        with torch.cuda.stream(stream):
            eng = Engine(sizes, [L.ACT_TANH] * len(sizes), n_in, n_out, batch, device=dev)
            eng.bind_params(W, b); eng.bind_inputs(None)
            if y is not None:
                eng.bind_target(y)
            eng.load_state(xs)
            res = eng.run(300, loss_kind=L.LOSS_BERNOULLI if n_out else L.LOSS_NONE, lr=0.03, noise_mode=L.NOISE_PHILOX, seed=5 + i,
                          step_base=0, acc_begin=100, acc_end=300, energy_mode=L.ENERGY_ALL)
            out = [torch.empty_like(x) for x in xs]
            eng.store_state(out)
            flat = eng.read_param_grads_flat()
        return eng, res, out, flat

    torch.cuda.synchronize()
    alone = []
    for i in range(2):
        eng, res, out, flat = launch(i, torch.cuda.current_stream())
        eng.sync_check()
        alone.append((res.energies.clone(), [o.clone() for o in out], flat.clone()))
        eng.close()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    live = [launch(i, streams[i]) for i in range(2)]          # both in flight before either is waited for
    for i, (eng, res, out, flat) in enumerate(live):
        with torch.cuda.stream(streams[i]):
            eng.sync_check()
        assert torch.equal(res.energies, alone[i][0])
        for a, c in zip(out, alone[i][1]):
            assert torch.equal(a, c)
        assert torch.equal(flat, alone[i][2])
        eng.close()


def test_gemm_core_accuracy_against_fp64():
    """The step kernel multiplies fp32 operands as products of scaled fp16 pieces with fp32 accumulation (csrc/mcpc_gemm_f16.h: three
    MFMAs per product, four in contractions of at most 64 terms; rounds 3-4: six bf16 MFMAs).  ONE step with
    lr = 1 and no noise exposes every contraction of a step -- three forward GEMMs, the read-out and four back-projections, K = 32 ..
    784 -- in x_new = x - g.  Against an fp64 evaluation of the same step (reference pc_layer.py:266-300 for the errors,
    pc_trainer.py:862 for dF/dx) every element must sit within 1e-6 of the sum of the ABSOLUTE values of the terms that make it up:
    what a chain of two fp32 dot products may lose (the core alone: max 0.5-2.1e-7 of sum|terms| for K = 32 .. 784, the fp32 MFMA chain
    0.8-2.2e-7: profiles/r05_f16x4_study.txt)."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    dev = _dev()
    sizes, n_in, n_out, B = [30, 256, 256], 30, 784, 64
    g = torch.Generator().manual_seed(41)
    dims = [n_in] + sizes + [n_out]
    W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) * (3.0 / dims[j] ** 0.5)).to(dev) for j in range(4)]
    b = [(torch.rand(dims[j + 1], generator=g) * 2 - 1).to(dev) for j in range(4)]
    xs = [((torch.rand(B, n, generator=g) * 2 - 1) * 1.5).to(dev) for n in sizes]
    inputs = (torch.rand(B, n_in, generator=g) * 2 - 1).to(dev)
    y = (torch.rand(B, n_out, generator=g) * 2 - 1).to(dev)
    var = 0.8
    eng = Engine(sizes, [L.ACT_TANH] * 3, n_in, n_out, B, device=dev)
    eng.bind_params(W, b); eng.bind_inputs(inputs); eng.bind_target(y)
    eng.load_state(xs)
    eng.run(1, loss_kind=L.LOSS_GAUSSIAN, loss_var=var, lr=1.0, noise_mode=L.NOISE_NONE)
    out = [torch.empty_like(x) for x in xs]
    eng.store_state(out)
    eng.sync_check()
    eng.close()
    Wd, bd, xd = [w.double() for w in W], [v.double() for v in b], [x.double() for x in xs]
    f = [torch.tanh(x) for x in xd]
    fp = [1.0 - t * t for t in f]
    pre = [inputs.double()] + f                                   # input of Linear j
    mu = [pre[j] @ Wd[j].T + bd[j] for j in range(4)]
    amu = [pre[j].abs() @ Wd[j].abs().T + bd[j].abs() for j in range(4)]        # sum of |terms| of each prediction
    e = [xd[l] - mu[l] for l in range(3)] + [(mu[3] - y.double()) / var]
    ae = [xd[l].abs() + amu[l] for l in range(3)] + [(amu[3] + y.double().abs()) / var]
    # the same step in torch's fp32 on this GPU (rocBLAS / hipBLASLt fp32 GEMMs: what the reference's loop would run here), for scale
    f32 = [torch.tanh(x) for x in xs]
    fp32 = [1.0 - t * t for t in f32]
    pre32 = [inputs] + f32
    mu32 = [pre32[j] @ W[j].T + b[j] for j in range(4)]
    e32 = [xs[l] - mu32[l] for l in range(3)] + [(mu32[3] - y) / var]
    worst_engine, worst_torch = 0.0, 0.0
    for l in range(3):
        sign = 1.0 if l == 2 else -1.0
        back = e[l + 1] @ Wd[l + 1]
        gl = e[l] + sign * fp[l] * back
        bound = ae[l] + fp[l].abs() * (ae[l + 1] @ Wd[l + 1].abs())
        err = (out[l].double() - (xd[l] - gl)).abs()
        assert float((err / bound).max()) < 1e-6, (l, float((err / bound).max()))
        assert float((err / bound).max()) > 0                      # (not vacuous: the step did change x)
        assert float(gl.abs().max()) > 0.1
        out32 = xs[l] - (e32[l] + sign * fp32[l] * (e32[l + 1] @ W[l + 1]))
        worst_engine = max(worst_engine, float((err / bound).max()))
        worst_torch = max(worst_torch, float(((out32.double() - (xd[l] - gl)).abs() / bound).max()))
    # the fp16-piece arithmetic is in the error class of the fp32 GEMMs torch runs on the same device: within a factor of two of them
    print(f"[gemm core] max error / sum|terms| against fp64: engine {worst_engine:.3e}, torch fp32 on this GPU {worst_torch:.3e}")
    assert worst_engine <= max(2.0 * worst_torch, 3e-7), (worst_engine, worst_torch)


def _mixed_scale_problem(dev, act, seed=43):
    """cfg-M's widths, 64 chains whose states (and targets) differ in scale by 10^(c mod 7 - 3), no biases: every B row of every GEMM has a
    scale of its own."""
    sizes, n_in, n_out, B = [30, 256, 256], 30, 784, 64
    g = torch.Generator().manual_seed(seed)
    dims = [n_in] + sizes + [n_out]
    W = [((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) * (3.0 / dims[j] ** 0.5)).to(dev) for j in range(4)]
    b = [torch.zeros(dims[j + 1], device=dev) for j in range(4)]
    scale = (10.0 ** (torch.arange(B) % 7 - 3).float()).unsqueeze(1)
    xs = [((torch.rand(B, n, generator=g) * 2 - 1) * 1.5 * scale).to(dev) for n in sizes]
    inputs = ((torch.rand(B, n_in, generator=g) * 2 - 1) * scale).to(dev)
    y = ((torch.rand(B, n_out, generator=g) * 2 - 1) * scale).to(dev)
    return sizes, n_in, n_out, B, W, b, xs, inputs, y, scale.to(dev)


@pytest.mark.parametrize("tuning", [None, "ws=0"], ids=["in-place kernel (row words)", "barrier kernel (row scans)"])
def test_rows_of_mixed_scale_keep_their_own_precision(tuning):
    """The fp16 GEMM core scales every CHAIN ROW of its LDS operand by a power of two of its own (csrc/mcpc_gemm_f16.h; in the in-place
    kernel the exponent comes from the epilogue waves that wrote the row: mcpc_kernels.h, rowexp_track).  Chains whose values differ by six
    orders of magnitude sit side by side in one 16-chain tile here; ONE step with lr = 1 against an fp64 evaluation, every element within
    1e-6 of the sum of the absolute values of ITS OWN terms -- a scale shared by the tile would leave the small chains a few bits
    (profiles/r05_f16x4_study.txt: ten times the error on rows of mixed scale, and that with a spread of 1e6 instead of 1e3)."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    dev = _dev()
    sizes, n_in, n_out, B, W, b, xs, inputs, y, scale = _mixed_scale_problem(dev, "identity")
    var = 0.8
    eng = Engine(sizes, [L.ACT_IDENTITY] * 3, n_in, n_out, B, device=dev, tuning=tuning)
    eng.bind_params(W, b); eng.bind_inputs(inputs); eng.bind_target(y)
    eng.load_state(xs)
    eng.run(1, loss_kind=L.LOSS_GAUSSIAN, loss_var=var, lr=1.0, noise_mode=L.NOISE_NONE)
    out = [torch.empty_like(x) for x in xs]
    eng.store_state(out)
    eng.sync_check()
    eng.close()
    Wd, xd = [w.double() for w in W], [x.double() for x in xs]
    pre = [inputs.double()] + xd
    mu = [pre[j] @ Wd[j].T for j in range(4)]
    amu = [pre[j].abs() @ Wd[j].abs().T for j in range(4)]
    e = [xd[l] - mu[l] for l in range(3)] + [(mu[3] - y.double()) / var]
    ae = [xd[l].abs() + amu[l] for l in range(3)] + [(amu[3] + y.double().abs()) / var]
    worst = 0.0
    for l in range(3):
        sign = 1.0 if l == 2 else -1.0
        gl = e[l] + sign * (e[l + 1] @ Wd[l + 1])
        bound = ae[l] + ae[l + 1] @ Wd[l + 1].abs()
        err = (out[l].double() - (xd[l] - gl)).abs()
        rel = err / bound
        worst = max(worst, float(rel.max()))
        # the bound of a chain follows the chain's own scale (six orders of magnitude between the chains of a tile) ...
        per_chain = bound.max(dim=1).values / scale[:, 0].double()
        assert float(per_chain.max() / per_chain.min()) < 50.0
        # ... and every chain meets it, the smallest like the largest
        assert float(rel.max()) < 1e-6, (l, float(rel.max()), int(rel.max(dim=1).values.argmax()))
    assert worst > 0


def test_row_words_equal_row_scans_over_many_steps():
    """The in-place kernel takes a row's exponent from the word its producers keep, the barrier kernel scans the row: the same exponent,
    so the trajectories of rows of mixed scale agree BITWISE after 40 steps (noise included) -- any word that was stale, early, or short of
    a tile would change a scale and with it the bits."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    dev = _dev()
    sizes, n_in, n_out, B, W, b, xs, inputs, y, scale = _mixed_scale_problem(dev, "relu", seed=44)
    y01 = (y > 0).float()
    res = []
    for tuning in (None, "ws=0"):
        eng = Engine(sizes, [L.ACT_RELU] * 3, n_in, n_out, B, device=dev, tuning=tuning)
        eng.bind_params(W, b); eng.bind_inputs(inputs); eng.bind_target(y01)
        eng.load_state(xs)
        eng.run(40, loss_kind=L.LOSS_BERNOULLI, lr=0.03, noise_mode=L.NOISE_PHILOX, noise_var=2.0, seed=5, step_base=0, energy_mode=L.ENERGY_ALL)
        out = [torch.empty_like(x) for x in xs]
        eng.store_state(out)
        eng.sync_check()
        res.append((out, None, eng.query()["step_kernel"]))
        eng.close()
    assert "ws2" in res[0][2] and "ws2" not in res[1][2], (res[0][2], res[1][2])
    for a, c in zip(res[0][0], res[1][0]):
        assert torch.equal(a, c)
