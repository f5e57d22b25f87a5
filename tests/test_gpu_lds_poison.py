"""No result of the step kernels depends on LDS content they did not produce.

Round 3's intermittent NaN: the GEMM core (bf16x6 then, fp16 x 3 now) reads its LDS operand in whole 32-deep k-blocks, 12 floats beyond a row whose width
is 16 mod 32; zero weights made those products zero only while the over-read values were FINITE, and for the last row of a plan the
over-read left the plan and returned whatever another kernel had left in the CU's LDS (scripts/nan_repro.py reproduces it on the old
builds; profiles/r04_nan_repro.txt).  Since round 4 the over-reading lanes take their address from a zero region of the plan
(csrc/mcpc_gemm_f16.h, KParams::lds_zero; the fp16 core of round 5 reads its operand the same way, in the GEMM and in the row pre-pass): the products are zeros whatever lies behind the row.  This file pins that by construction: the LDS of EVERY compute unit is filled with a poison
pattern (mcpc_debug_poison_lds: signalling NaN, +Inf, a huge finite value, zeros) immediately before each engine call, on networks
whose widths are not multiples of 32 (200, 33, 17, 21, 40 ...), on every kernel form and LDS plan the library has; results must match
the NumPy oracle AND be bitwise independent of the pattern.
"""
import numpy as np
import pytest
import torch

from oracle import mcpc_oracle as mo
from oracle import philox
from oracle.cases import make_case_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
POISONS = {"snan": 0x7FA00000, "inf": 0x7F800000, "huge": 0x7F7FFFFF, "zero": 0x0}
# (latent sizes, n_out, batch): batches are multiples of 16 so that the LAST row of every workgroup is a live chain
CASES = [([40, 384, 200], 784, 48), ([33, 17, 21], 50, 32), ([200], 0, 16), ([17, 200, 33], 120, 64), ([30, 256, 256], 784, 32),
         ([21, 40], 17, 80), ([16, 496], 0, 16)]
FORMS = [None, "ws=2", "ws=0", "overlay16=1", "no_xl=1", "no_lean=1"]      # (None: the unified-wave kernel where its plan fits, else the in-place kernel)


def test_poison_reaches_every_compute_unit():
    from montecarlopredictivecoding_amd.engine import debug_poison_lds
    for word in POISONS.values():
        debug_poison_lds(DEV, word)          # raises unless all CUs were visited and read their pattern back


@pytest.mark.parametrize("tuning", FORMS, ids=[f or "default" for f in FORMS])
@pytest.mark.parametrize("sizes,n_out,B", CASES, ids=["-".join(map(str, c[0])) + f"-{c[1]}" for c in CASES])
def test_results_do_not_depend_on_foreign_lds_content(sizes, n_out, B, tuning):
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine, debug_poison_lds
    T, lr, seed = 6, 0.02, 9
    case = dict(sizes=sizes, acts=["relu"] * len(sizes), ecoef=[1.0] * len(sizes), n_in=sizes[0], n_out=n_out, loss="bernoulli" if n_out else "none",
                var=1.0, perc=0.5, B=B, seed=77, x0_range=1.0, calls=[dict(T=T)])
    W, b, X0, inputs, target = make_case_inputs(case)
    net = mo.NetSpec(sizes=sizes, acts=[mo.ACT_RELU] * len(sizes), W=W, b=b, ecoef=case["ecoef"], has_head=bool(n_out))
    lspec = mo.LossSpec(mo.LOSS_BERNOULLI, target) if n_out else mo.LossSpec()
    ref = mo.run(net, inputs, X0, lspec, mo.XOpt(mo.OPT_SGD, lr), T, noise=lambda t, l: philox.layer_normals(seed, t, l, 0, B, sizes[l]),
                 accumulate_p_at=list(range(1, T)))
    outs = {}
    for name, word in POISONS.items():
        eng = Engine(sizes, [L.ACT_RELU] * len(sizes), sizes[0], n_out, B, device=DEV, tuning=tuning)
        eng.bind_params([torch.from_numpy(w).to(DEV) for w in W], [torch.from_numpy(x).to(DEV) for x in b])
        eng.bind_inputs(None)
        if n_out:
            eng.bind_target(torch.from_numpy(target).to(DEV))
        xs = [torch.from_numpy(x).to(DEV) for x in X0]
        eng.load_state(xs)
        torch.cuda.synchronize()
        debug_poison_lds(DEV, word)
        res = eng.run(T, loss_kind=L.LOSS_BERNOULLI if n_out else L.LOSS_NONE, lr=lr, noise_mode=L.NOISE_PHILOX, seed=seed, step_base=0,
                      acc_begin=1, acc_end=T, energy_mode=L.ENERGY_ALL)
        eng.store_state(xs)
        eng.sync_check()
        outs[name] = (res.energies.cpu().numpy(), [x.cpu().numpy() for x in xs], eng.read_param_grads_flat().cpu().numpy())
        eng.close()
    en, xs_o, flat = outs["snan"]
    assert np.isfinite(en).all() and all(np.isfinite(x).all() for x in xs_o) and np.isfinite(flat).all(), f"tuning {tuning}: NaN / Inf leaked from foreign LDS"
    for name in ("inf", "huge", "zero"):
        assert np.array_equal(en, outs[name][0]), (tuning, name)
        for a, c in zip(xs_o, outs[name][1]):
            assert np.array_equal(a, c), (tuning, name)
        assert np.array_equal(flat, outs[name][2]), (tuning, name)
    np.testing.assert_allclose(en[:, -1], ref.overall, rtol=1e-6, err_msg=f"tuning {tuning}")
    for l in range(len(sizes)):
        np.testing.assert_allclose(xs_o[l], ref.xs[l], rtol=0, atol=1e-5 * max(1.0, float(np.abs(ref.xs[l]).max())), err_msg=f"tuning {tuning}")
    want = np.concatenate([np.concatenate([gw.reshape(-1), gb.reshape(-1)]) for gw, gb in zip(ref.gW, ref.gb)])
    np.testing.assert_allclose(flat, want, rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(want).max())), err_msg=f"tuning {tuning}")
