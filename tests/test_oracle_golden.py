"""Pin the CPU restatement (oracle/) against vectors produced by the imported reference."""
import numpy as np
import pytest

from oracle import mcpc_oracle as mo
from tests.golden_util import Golden, fixture_names


def run_oracle_case(g, dtype=np.float32):
    """Replays every call of a fixture on the oracle; yields (ci, call, result)."""
    xs = g.X0
    gW = gb = None
    W, b = g.W, g.b
    for ci, call in enumerate(g.case["calls"]):
        if not call.get("sample_x", True) and ci > 0:
            xs0 = xs
        else:
            xs0 = g.X0
        up, acc = g.schedules(call)
        res = mo.run(g.net(W, b), g.inputs, xs0, g.loss_spec(), g.xopt(call), call["T"],
                     noise=g.noise(ci), noise_var=call.get("noise_var", 2.0),
                     update_p_at=up, accumulate_p_at=acc, record_at=call.get("record_at", []),
                     dtype=dtype, gW_in=gW, gb_in=gb)
        yield ci, call, res
        xs, gW, gb = res.xs, res.gW, res.gb
        if up:      # a p-step changed the parameters: continue from the reference's post-step values
            W = [g.get(ci, f"W{j}_after") for j in range(len(W))]
            b = [g.get(ci, f"b{j}_after") if bb is not None else None for j, bb in enumerate(b)]


@pytest.mark.parametrize("name", fixture_names(exclude=("g9_",)))
def test_oracle_matches_reference(name):
    g = Golden(name)
    nc = g.case.get("rec_chains", None)
    for ci, call, res in run_oracle_case(g):
        e_ref = g.get(ci, "energy")
        # energies: fp32 sums over B*n elements, reduction order differs (SURVEY 7 'Deterministic energies')
        np.testing.assert_allclose(res.energy, e_ref, rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(res.loss, g.get(ci, "loss"), rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(res.overall, g.get(ci, "overall"), rtol=2e-5, atol=1e-5)
        for t in call.get("record_at", []):
            for l in range(len(g.case["sizes"])):
                np.testing.assert_allclose(res.rec_xs[t][l][:nc], g.get(ci, f"x_t{t}_l{l}"), rtol=0, atol=2e-4)
            np.testing.assert_allclose(res.rec_out[t][:nc], g.get(ci, f"out_t{t}"), rtol=0, atol=5e-4)
        for l in range(len(g.case["sizes"])):
            np.testing.assert_allclose(res.xs[l][:nc], g.get(ci, f"x_final_l{l}"), rtol=0, atol=2e-4)
        for j in range(len(g.W)):
            if g.has(ci, f"gW{j}"):
                ref = g.get(ci, f"gW{j}")
                np.testing.assert_allclose(res.gW[j], ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))
            if g.has(ci, f"gb{j}") and res.gb[j] is not None:
                ref = g.get(ci, f"gb{j}")
                np.testing.assert_allclose(res.gb[j], ref, rtol=1e-4, atol=1e-5 * max(1.0, np.abs(ref).max()))
            if g.has(ci, f"gW{j}_idx"):
                idx = g.get(ci, f"gW{j}_idx")
                ref = g.get(ci, f"gW{j}_val")
                scale = g.get(ci, f"gW{j}_abs") / res.gW[j].size
                np.testing.assert_allclose(res.gW[j].reshape(-1)[idx], ref, rtol=1e-3, atol=1e-3 * scale)
                np.testing.assert_allclose(np.abs(res.gW[j].astype(np.float64)).sum(), g.get(ci, f"gW{j}_abs"), rtol=1e-4)


def test_oracle_fp64_close_to_fp32_on_short_windows():
    """The restatement is precision-generic: fp64 vs fp32 stays tight over the 50-step tiny cases."""
    g = Golden("g1_tanh_gaussian_sgdnoise")
    r32 = [r for _, _, r in run_oracle_case(g, np.float32)][0]
    r64 = [r for _, _, r in run_oracle_case(g, np.float64)][0]
    for a, b in zip(r32.xs, r64.xs):
        assert np.abs(a - b).max() < 1e-4
