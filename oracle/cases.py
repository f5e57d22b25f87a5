"""Synthetic inputs for golden / parity cases.  TEST INFRASTRUCTURE ONLY (part of ``oracle/``).

A *case* is a plain dict (JSON-serialisable, stored inside each fixture as ``case_json``):

    sizes      latent layer sizes n_1..n_L          acts   'identity'|'relu'|'tanh' per latent layer
    ecoef      energy coefficient per latent layer   n_in   width of the pseudo-input, n_out head width (0 = no head)
    loss       'none'|'zero'|'gaussian'|'bernoulli'|'gaussian_mask'|'bernoulli_mask'   var, perc
    B, seed    batch and generator seed              calls  list of train_on_batch descriptions

Weights follow torch's ``nn.Linear`` default U(-1/sqrt(in), 1/sqrt(in)) but are drawn from
``oracle.philox`` so that the same numbers exist here, in the fixture generator and on the
GPU box without shipping megabytes.
"""
import numpy as np

from oracle import philox


def make_params(seed, n_in, sizes, n_out, no_bias=()):
    dims = [n_in] + list(sizes) + ([n_out] if n_out else [])
    W, b = [], []
    for j in range(len(dims) - 1):
        k = 1.0 / np.sqrt(dims[j])
        W.append(philox.uniform_pm(seed, 2 * j, (dims[j + 1], dims[j]), -k, k))
        b.append(None if j in no_bias else philox.uniform_pm(seed, 2 * j + 1, (dims[j + 1],), -k, k))
    return W, b


def make_case_inputs(case):
    sizes, B, seed = case["sizes"], case["B"], case["seed"]
    n_in, n_out = case["n_in"], case["n_out"]
    W, b = make_params(seed, n_in, sizes, n_out, no_bias=tuple(case.get("no_bias", ())))
    if case.get("weight_scale"):
        W = [w * np.float32(case["weight_scale"]) for w in W]
    if "const_params" in case:      # figure_2/3 style hand-set toy parameters
        for j, (wv, bv) in enumerate(case["const_params"]):
            if wv is not None:
                W[j][...] = wv
            if bv is not None and b[j] is not None:
                b[j][...] = bv
    r = case.get("x0_range", 1.0)
    X0 = [philox.uniform_pm(seed, 100 + l, (B, n), -r, r) for l, n in enumerate(sizes)]
    if case.get("inputs_zero", True):
        inputs = np.zeros((B, n_in), dtype=np.float32)
    else:
        inputs = philox.uniform_pm(seed, 200, (B, n_in), -1.0, 1.0)
    target = None
    if n_out and case["loss"] not in ("none", "zero"):
        u = philox.uniform_pm(seed, 300, (B, n_out), 0.0, 1.0)
        if case["loss"].startswith("bernoulli"):
            target = (u < case.get("target_p", 0.3)).astype(np.float32)
        else:
            target = (2.0 * u - 0.5).astype(np.float32)
    return W, b, X0, inputs, target


def call_noise(case, call_index):
    """noise(t, l) callable for call ``call_index`` (step counter continues across calls)."""
    t_base = sum(c["T"] for c in case["calls"][:call_index])
    sizes, B, seed = case["sizes"], case["B"], case["seed"]

    def noise(t, l):
        return philox.layer_normals(seed, t_base + t, l, 0, B, sizes[l])
    return noise
