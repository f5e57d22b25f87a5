"""CPU checks of the distribution-level fixtures (g13: stationary statistics of the reference's own sampler, g14:
get_representations), so that what the GPU tests compare against is itself pinned:
  * the fixtures are well-formed (12 seeds, symmetric positive covariance, consistent labels);
  * the ORACLE driven by the NumPy twin of the device Philox (oracle/philox.py, bit-identical to the HIP generator) samples the
    same stationary distribution as the reference's `random_step` with torch's `normal_` (utils/model.py:35-44) -- a shorter
    window than the fixture's (500 steps instead of 2000), so the yardstick is the reference's seed spread scaled accordingly.
"""
import json
import os

import numpy as np

from oracle import mcpc_oracle as mo
from oracle import philox
from oracle.cases import make_case_inputs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sampling_fixtures_are_well_formed():
    for name in ("tanh_gaussian", "relu_bernoulli", "relu_zero"):
        g = np.load(os.path.join(GOLDEN, f"g13_sampling_moments_{name}.npz"))
        case = json.loads(str(g["case_json"]))
        if name == "relu_zero":
            # unclamped generation (figure_3.py:125-161): a zero loss, and the read-out's moments beside the latents'
            no = case["n_out"]
            assert case["loss"] == "zero" and np.all(g["energies"][:, 0] == 0)
            assert g["out_mean"].shape == (12, no) and g["out_cov"].shape == (12, no, no)
            for c in g["out_cov"]:
                assert np.allclose(c, c.T, atol=1e-12) and np.linalg.eigvalsh(c).min() > -1e-9
            assert (np.array([np.diag(c) for c in g["out_cov"]]).std(0, ddof=1) / np.array([np.diag(c) for c in g["out_cov"]]).mean(0) < 0.02).all()
        n = sum(case["sizes"])
        assert g["mean"].shape == (12, n) and g["cov"].shape == (12, n, n) and g["energies"].shape == (12, 3)
        assert (int(g["burn"]), int(g["T"]), case["B"]) == (500, 2500, 4096)
        for c in g["cov"]:
            assert np.allclose(c, c.T, atol=1e-12) and np.linalg.eigvalsh(c).min() > 0
        # Langevin at noise variance 2 (the correct temperature): unit-ish variances of the top layer (prior N(b0, 1) blurred by the likelihood)
        assert 0.9 < np.diag(g["cov"].mean(0))[:case["sizes"][0]].min() and np.diag(g["cov"].mean(0))[:case["sizes"][0]].max() < 1.1
        np.testing.assert_allclose(g["energies"][:, 0] + g["energies"][:, 1], g["energies"][:, 2], rtol=1e-6)      # fp32 `loss + energy` per step in the reference
        # the seeds agree with each other to a fraction of a percent: the statistic is sharp enough to be a yardstick
        assert (np.array([np.diag(c) for c in g["cov"]]).std(0, ddof=1) < 0.01).all()
    r = np.load(os.path.join(GOLDEN, "g14_representations.npz"))
    assert r["map_reps"].shape == (48, 8) and r["expectation_reps"].shape == (12, 48, 8) and r["full_reps"].shape == (480, 8)
    assert np.array_equal(r["map_labels"], r["labels"]) and np.array_equal(r["expectation_labels"], r["labels"])
    # "full", n = 10: per batch of 16 the labels are repeated 10 times (reference utils/model.py:158 `label.repeat(n)`)
    assert np.array_equal(r["full_labels"], np.concatenate([np.tile(r["labels"][k:k + 16], 10) for k in (0, 16, 32)]))


def test_oracle_with_the_philox_twin_samples_the_reference_distribution():
    g = np.load(os.path.join(GOLDEN, "g13_sampling_moments_relu_bernoulli.npz"))
    case = json.loads(str(g["case_json"]))
    W, b, X0, inputs, target = make_case_inputs(case)
    sizes, B = case["sizes"], case["B"]
    net = mo.NetSpec(sizes=sizes, acts=[mo.ACT_RELU] * 3, W=W, b=b)
    burn, T = 500, 1000
    ref = mo.run(net, inputs, X0, mo.LossSpec(mo.LOSS_BERNOULLI, target), mo.XOpt(mo.OPT_SGD, float(g["lr"])), T,
                 noise=lambda t, l: philox.layer_normals(4242, t, l, 0, B, sizes[l]), noise_var=float(g["noise_var"]),
                 record_at=range(burn, T))
    x = np.concatenate([np.concatenate(ref.rec_xs[t], axis=1) for t in range(burn, T)]).astype(np.float64)
    mean = x.mean(0)
    cov = x.T @ x / x.shape[0] - np.outer(mean, mean)
    # a window of 500 steps against the fixture's 2000: the time-average noise of this run is ~2x a reference seed's
    widen = np.sqrt(4.0 + 1.0 / 12.0)
    iu = np.triu_indices(cov.shape[0], k=1)
    for key, got, refs in (("means", mean, g["mean"]), ("variances", np.diag(cov), np.array([np.diag(c) for c in g["cov"]])),
                           ("covariances", cov[iu], np.array([c[iu] for c in g["cov"]]))):
        z = (got - refs.mean(0)) / (refs.std(0, ddof=1) * widen)
        assert np.abs(z).max() < 8.0, (key, float(np.abs(z).max()))
        assert (z * z).mean() < 2.5, (key, float((z * z).mean()))
    en = np.array([ref.loss[burn:].mean(), ref.energy[burn:].mean(), ref.overall[burn:].mean()])
    np.testing.assert_allclose(en, g["energies"].mean(0), rtol=3e-3)
