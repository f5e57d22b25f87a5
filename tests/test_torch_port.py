"""The cpu_baseline port (oracle/torch_port.py, torch autograd) agrees with the NumPy oracle and a golden."""
import numpy as np
import torch

from oracle import mcpc_oracle as mo
from oracle import torch_port
from tests.golden_util import Golden


def test_port_matches_oracle_and_reference_golden():
    g = Golden("g1_relu_bernoulli_sgdnoise")
    call = g.case["calls"][0]
    T = call["T"]
    torch.manual_seed(0)
    model, nodes, lins = torch_port.build(g.case["sizes"], g.case["acts"], g.case["n_in"], g.case["n_out"], g.W, g.b)
    loss_fn = torch_port.make_loss("bernoulli", g.target)
    en, lo = torch_port.run(model, nodes, lins, g.inputs, g.X0, loss_fn, T, call["lr"], noise_var=2.0,
                            noise=g.noise(0), acc_begin=0)
    np.testing.assert_allclose(en, g.get(0, "energy"), rtol=2e-5)
    np.testing.assert_allclose(lo, g.get(0, "loss"), rtol=2e-5)
    for l, node in enumerate(nodes):
        np.testing.assert_allclose(node.state.detach().numpy(), g.get(0, f"x_final_l{l}"), atol=2e-4)
    for j, lin in enumerate(lins):
        np.testing.assert_allclose(lin.weight.grad.numpy(), g.get(0, f"gW{j}"), rtol=1e-4, atol=1e-4)
    res = mo.run(g.net(), g.inputs, g.X0, g.loss_spec(), g.xopt(call), T, noise=g.noise(0))
    np.testing.assert_allclose(en, res.energy, rtol=2e-5)
