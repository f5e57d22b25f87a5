"""Adversarial accuracy tests of the fp16-piece arithmetic: correlated roundings, weights far below their Linear's
maximum, and the Hebbian flush's one-exponent-per-image scaling -- each against fp64 with the bound DERIVED below, and torch's fp32 GEMM
on the same GPU printed beside it (the arithmetic being matched is torch's fp32 `addmm` / `mm`: reference pc_trainer.py:733,862).

The arithmetic (csrc/mcpc_gemm_f16.h).  An operand a is scaled by a power of two (exact) and cut into two fp16 pieces, a 2^s = a_h + a_m + r:
a_h = fp16(a 2^s) has relative error <= 2^-11 (11 significant bits, round to nearest), the residual is exact in fp32, a_m = fp16(residual)
leaves |r| <= 2^-11 |residual| <= 2^-22 |a 2^s| as long as a_m is a NORMAL fp16 number.  So every operand enters as a (1 + d), |d| <= 2^-22
(fp32 keeps 2^-24).  The piece products a_h b_h, a_h b_m, a_m b_h are exact in the MFMA's fp32 input; the fourth, a_m b_m <= 2^-22 |a b|,
is dropped in contractions of more than 64 terms.  Per TERM therefore

    |computed term - a b|  <=  (2^-22 + 2^-22 + 2^-22 + 2^-44) |a b|  ~  3 x 2^-22 |a b|  =  7.2e-7 |a b|      (K > 64)
                           <=  (2^-22 + 2^-22 + 2^-44) |a b|          ~  2 x 2^-22 |a b|  =  4.8e-7 |a b|      (K <= 64)

WORST CASE -- every term with the same sign of error: a B row of one repeated value against same-sign weights of one repeated value.  On
independent operands these errors average out over K (the i.i.d. figure of profiles/r05_f16x4_study.txt: 0.5-1.6e-7 of sum |terms|); here
they do not, and the bound above is what holds -- beside the fp32 accumulation of the MFMA chain (one rounding of the running sum per
MFMA: 3 K / 32 roundings of <= 2^-24 each -- ~sqrt of that count on independent terms, ALL of it on identical terms), which torch's fp32
GEMM has as well, and more of: measured on these inputs (K = 96 / 256 / 784) the engine is at 0.58 / 1.26 / 3.65e-6 of sum |terms|, torch's
fp32 GEMM on the same MI355X at 1.43 / 3.80 / 11.7e-6.  The tests assert |error| <= (3 x 2^-22 + 3 ceil(K / 32) 2^-24) sum |terms|.

RANGE.  The scale brings the largest |value| of a B row (of a whole Linear for the packed weights, of a whole spilled image in the flush)
into [2^14, 2^15).  A value 2^-rho of that maximum keeps a normal second piece while rho <= 18; below, the second piece is an fp16
subnormal with ABSOLUTE error 2^-25, i.e. a relative error of 2^(rho - 39) of the value: 22 bits at rho = 17, 19 at rho = 20, 11 (the first
piece alone) from rho = 28.  Per-row scaling keeps this from crossing chains in the step GEMMs; it does cross units inside a Linear (one
exponent per weight matrix) and chains / steps inside a spilled image of the flush (one exponent per image): see the two tests below.
"""
import numpy as np
import pytest
import torch

from tests import parity_log

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
C_OP = 3.0 * 2.0 ** -22          # operand representation, K > 64 (derivation above)


def c_acc(K):
    """fp32 accumulation of the MFMA chain: ONE rounding of the running sum per MFMA, 3 per 32-deep k-block, each <= 2^-24 of the sum.  On
    independent terms they average out (~sqrt of their count); with every term the same value the running sum steps through a regular
    pattern and they add up -- torch's fp32 GEMM on the same GPU loses three times as much on these inputs (printed beside)."""
    return 3.0 * ((K + 31) // 32) * 2.0 ** -24


def _adversarial_values(n, g):
    """24-bit values whose two fp16 roundings both go the same way, mixed with random 24-bit mantissas: 1 + 2^-11 - 2^-23 sits just
    below the first piece's rounding boundary AND leaves a residual just below the second's."""
    crafted = torch.tensor([1.0 + 2.0 ** -11 - 2.0 ** -23, 1.0 + 2.0 ** -12 + 2.0 ** -23, 1.9999999, 1.0 + 2.0 ** -11 + 2.0 ** -22 - 2.0 ** -23,
                            1.5 + 2.0 ** -12 - 2.0 ** -23, 1.25 + 3 * 2.0 ** -13 + 2.0 ** -23], dtype=torch.float32)
    rnd = 1.0 + torch.rand(max(n - crafted.numel(), 0), generator=g)
    v = torch.cat([crafted, rnd])[:n]
    return v * (2.0 ** torch.randint(-3, 4, (n,), generator=g).float())


@pytest.mark.parametrize("tuning", [None, "ws=2", "ws=0"], ids=["default", "inplace", "barrier"])
@pytest.mark.parametrize("K", [96, 256, 784])
def test_correlated_roundings_one_repeated_value_against_same_sign_weights(K, tuning):
    """x_0 rows of ONE repeated 24-bit value per chain, Linear 1 rows of ONE repeated positive 24-bit value per unit, no a_m b_m term
    (K > 64): every term of mu_1 = W_1 x_0 carries the same relative error.  One step with lr = 1 from x_1 = 0 returns mu_1."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    g = torch.Generator().manual_seed(K)
    B, n1 = 32, 64
    v = _adversarial_values(B, g)                     # per chain
    wv = _adversarial_values(n1, g) * 0.01            # per unit
    x0 = v[:, None].expand(B, K).contiguous().to(DEV)
    W1 = wv[:, None].expand(n1, K).contiguous().to(DEV)
    W0 = torch.zeros(K, 4, device=DEV)
    eng = Engine([K, n1], [L.ACT_IDENTITY] * 2, 4, 0, B, device=DEV, tuning=tuning)
    eng.bind_params([W0, W1], [torch.zeros(K, device=DEV), torch.zeros(n1, device=DEV)])
    eng.bind_inputs(None)
    eng.load_state([x0, torch.zeros(B, n1, device=DEV)])
    eng.run(1, loss_kind=L.LOSS_NONE, lr=1.0, noise_mode=L.NOISE_NONE)
    out = [torch.empty(B, K, device=DEV), torch.empty(B, n1, device=DEV)]
    eng.store_state(out)
    eng.sync_check()
    eng.close()
    exact = x0.double() @ W1.double().T                                       # = K v_c w_u
    terms = x0.double().abs() @ W1.double().abs().T
    err = ((out[1].double() - exact).abs() / terms).max().item()
    err_torch = (((x0 @ W1.T).double() - exact).abs() / terms).max().item()
    bound = C_OP + c_acc(K)
    print(f"[correlated roundings] K={K} tuning={tuning}: engine {err:.3e} of sum|terms| (bound {bound:.2e}), torch fp32 on this GPU {err_torch:.3e}")
    assert err <= bound, (err, err_torch)
    assert err > 2.0 ** -26                     # (not vacuous: the operands are not exactly representable in 22 bits)


@pytest.mark.parametrize("tuning", [None, "ws=2"], ids=["default", "inplace"])
@pytest.mark.parametrize("rho", [17, 20])
def test_weights_far_below_the_maximum_of_their_linear(rho, tuning):
    """One weight of Linear 1 is 2^rho x the others (rho = 17: the others' second pieces are fp16 subnormals, still 22 bits; rho = 20:
    19 bits -- RANGE above).  Rows WITHOUT the large weight see only small weights: their bound is max(3 x 2^-22, 2^-22 + 2^(rho - 39))."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    g = torch.Generator().manual_seed(rho)
    B, K, n1 = 32, 256, 64
    x0 = ((torch.rand(B, K, generator=g) + 0.5) * (2.0 ** torch.randint(-2, 3, (B, 1), generator=g).float())).to(DEV)
    W1 = ((torch.rand(n1, K, generator=g) + 0.5) * 0.01)
    W1[5, 7] = float(W1.median()) * 2.0 ** rho
    W1 = W1.to(DEV)
    eng = Engine([K, n1], [L.ACT_IDENTITY] * 2, 4, 0, B, device=DEV, tuning=tuning)
    eng.bind_params([torch.zeros(K, 4, device=DEV), W1], [torch.zeros(K, device=DEV), torch.zeros(n1, device=DEV)])
    eng.bind_inputs(None)
    eng.load_state([x0, torch.zeros(B, n1, device=DEV)])
    eng.run(1, loss_kind=L.LOSS_NONE, lr=1.0, noise_mode=L.NOISE_NONE)
    out = [torch.empty(B, K, device=DEV), torch.empty(B, n1, device=DEV)]
    eng.store_state(out)
    eng.sync_check()
    eng.close()
    exact = x0.double() @ W1.double().T
    terms = x0.double().abs() @ W1.double().abs().T
    rel = (out[1].double() - exact).abs() / terms
    small_rows = [u for u in range(n1) if u != 5]
    err_small, err_big = rel[:, small_rows].max().item(), rel[:, 5].max().item()
    bound_small = max(C_OP, 2.0 ** -22 + 2.0 ** (rho - 39)) + 2e-7        # (independent terms: the accumulation's roundings average out)
    print(f"[weights 2^-{rho} of the maximum] tuning={tuning}: rows of small weights {err_small:.3e} (bound {bound_small:.2e}), the row with the maximum {err_big:.3e}")
    assert err_small <= bound_small and err_big <= C_OP + 2e-7, (err_small, err_big)


def _flush_problem(scale_chain=None, scale_unit=None, seed=3):
    """cfg-M-like widths so that the tiled fp16 Hebbian kernel serves every Linear; identity activations and ALL-ZERO weights: every
    prediction is 0, so the spilled images are exactly e_l = x_l, f(x_l) = x_l and e_o = -y (var = 1) -- no rounding before the flush, whose
    arithmetic alone is then what an fp64 product of the same fp32 values measures."""
    g = torch.Generator().manual_seed(seed)
    sizes, n_in, n_out, B = [32, 128, 128], 8, 272, 64
    dims = [n_in] + sizes + [n_out]
    W = [torch.zeros(dims[j + 1], dims[j], device=DEV) for j in range(4)]
    b = [torch.zeros(dims[j + 1], device=DEV) for j in range(4)]
    xs = [(torch.rand(B, n, generator=g) * 2 - 1) for n in sizes]
    y = (torch.rand(B, n_out, generator=g) * 2 - 1)
    if scale_chain is not None:                  # one chain 10^6 x the others, in every image
        for x in xs:
            x[3] *= scale_chain
        y[3] *= scale_chain
    if scale_unit is not None:                   # ONE value (chain 3, unit 5 of the middle layer) 2^20 x the others
        xs[1][3, 5] *= scale_unit
    return sizes, n_in, n_out, B, W, b, [x.to(DEV) for x in xs], y.to(DEV)


@pytest.mark.parametrize("case", ["one chain 1e6 x the others", "one value 2^20 x the others"])
def test_hebbian_flush_with_one_exponent_per_image(case):
    """The flush scales a whole spilled image (all chains, all steps of a segment) by ONE power of two (csrc/mcpc_hebbian.h).  A chain that
    is 10^6 x the others raises the scale for everybody -- and sits in the sum of every dW entry, so relative to sum |terms| nothing is
    lost.  ONE large value (a single chain's single unit) raises the scale as well but is a term of only the entries of its unit's row /
    column: every other entry of that image's Linears keeps max(3 x 2^-22, 2^-22 + 2^(rho - 39)) of ITS OWN sum |terms| (rho = 20: 19
    bits; ADVICE r5 -- the fp32-MFMA flush, tuning heb_fp32=1, is the form without that limit, asserted at 1e-6 here)."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    big_unit = case.startswith("one value")
    sizes, n_in, n_out, B, W, b, xs, y = _flush_problem(scale_chain=None if big_unit else 1e6, scale_unit=2.0 ** 20 if big_unit else None)
    outs = {}
    for tuning in (None, "heb_fp32=1"):
        eng = Engine(sizes, [L.ACT_IDENTITY] * 3, n_in, n_out, B, device=DEV, tuning=tuning)
        eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
        eng.load_state(xs)
        eng.run(1, loss_kind=L.LOSS_GAUSSIAN, loss_var=1.0, lr=1e-30, noise_mode=L.NOISE_NONE, acc_begin=0, acc_end=1)       # (the state does not move)
        gW = [torch.empty_like(w) for w in W]
        for j in range(1, 4):
            eng.read_param_grads(j, gW[j], None)
        eng.sync_check()
        eng.close()
        outs[tuning] = gW
    xd = [x.double() for x in xs]
    pre = [None] + xd
    e = [None, xd[1], xd[2], -y.double()]              # zero weights: e_l = x_l - 0, e_o = (0 - y) / 1
    # dF/dW_j = -e_j^T f(x_{j-1}) for the latent Linears, +e_o^T f(x_{L-1}) for the read-out (reference pc_trainer.py:862 via autograd)
    sign = [None, -1.0, -1.0, 1.0]
    rho = 20
    worst = {}
    for j in range(1, 4):
        exact = sign[j] * (e[j].T @ pre[j])
        terms = e[j].abs().T @ pre[j].abs()
        for tuning, name in ((None, "fp16 x 3 flush"), ("heb_fp32=1", "fp32-MFMA flush")):
            rel = (outs[tuning][j].double() - exact).abs() / terms
            # the large value sits in x_2's image: operand of Linear 2 (as its error) and of Linear 3 (as its activation)
            touched = big_unit and j in (2, 3)
            bound = (max(C_OP, 2.0 ** -22 + 2.0 ** (rho - 39)) if touched else C_OP) + 2e-7
            worst[(j, name)] = rel.max().item()
            print(f"[flush, {case}] Linear {j}, {name}: max error {rel.max().item():.3e} of sum|terms| (bound {bound:.2e})")
            if tuning is None:
                assert rel.max().item() <= bound, (j, rel.max().item())
            else:
                assert rel.max().item() <= 1e-6, (j, rel.max().item())
    assert max(worst.values()) > 0
