"""Time-faithful CPU port of the reference loop (torch autograd on host cores).  TEST INFRASTRUCTURE ONLY.

Part of ``oracle/``: used by ``bench.py``'s ``cpu_baseline`` leg (kind = "port") and by tests.
The reference's Python cannot travel to the GPU box, so this module re-creates -- from the
behaviour documented in SURVEY.md section 3, not from the reference's text -- the same *op mix*
the reference executes per Langevin step on the CPU:

  * an ``nn.Sequential`` forward in which each latent node returns its own state tensor
    (pc_layer.py:300) and stores ``0.5*(mu-x)^2 .sum()`` (pc_layer.py:266-295),
  * ``loss + sum(energies)`` followed by ONE autograd ``backward()`` that produces the gradients
    of every x AND every parameter (pc_trainer.py:821-862),
  * ``optim.SGD`` / ``optim.Adam`` ``zero_grad`` + ``step`` on the x tensors (pc_trainer.py:848-877),
  * the Langevin kick as a second optimizer step on freshly drawn ``normal_`` gradients
    (utils/model.py:35-44).

``tests/test_torch_port.py`` checks it against the NumPy oracle (and therefore, transitively,
against the reference's golden vectors); BASELINE.md section 3 asks for it to be within +-10 % of
the imported reference's speed, which ``oracle/compare_speed.py`` measures in this container.
"""
import math

import torch
import torch.nn as nn


class LatentNode(nn.Module):
    """Holds the latent state of one PC layer; forward returns the state, remembers the energy."""

    def __init__(self, coef=1.0):
        super().__init__()
        self.coef = coef
        self.state = None
        self.energy = None

    def forward(self, mu):
        d = mu - self.state
        e = 0.5 * d ** 2
        if self.coef != 1.0:
            e = self.coef * e
        self.energy = e.sum()
        return self.state


def build(sizes, acts, n_in, n_out, W, b, ecoef=None):
    """nn.Sequential[Linear, LatentNode, act, ...] with given numpy parameters."""
    act_mod = {0: None, 1: nn.ReLU, 2: nn.Tanh, "identity": None, "relu": nn.ReLU, "tanh": nn.Tanh}
    mods, nodes = [], []
    dims = [n_in] + list(sizes) + ([n_out] if n_out else [])
    for l in range(len(sizes)):
        mods.append(nn.Linear(dims[l], dims[l + 1], bias=b[l] is not None))
        node = LatentNode(1.0 if ecoef is None else float(ecoef[l]))
        nodes.append(node)
        mods.append(node)
        a = act_mod[acts[l]]
        if a is not None:
            mods.append(a())
    if n_out:
        mods.append(nn.Linear(dims[-2], dims[-1], bias=b[len(sizes)] is not None))
    model = nn.Sequential(*mods)
    lins = [m for m in model if isinstance(m, nn.Linear)]
    with torch.no_grad():
        for j, lin in enumerate(lins):
            lin.weight.copy_(torch.as_tensor(W[j]))
            if b[j] is not None:
                lin.bias.copy_(torch.as_tensor(b[j]))
    return model, nodes, lins


def make_loss(kind, target=None, var=1.0, mask_start=0):
    if kind in (None, "none", "zero", 0):
        return None
    y = torch.as_tensor(target)
    if kind in ("gaussian", 1):
        return lambda out: (1.0 / var) * 0.5 * (out[:, mask_start:] - y[:, mask_start:]).pow(2).sum()
    if kind in ("bernoulli", 2):
        bce = nn.BCEWithLogitsLoss(reduction="sum")
        return lambda out: bce(out[:, mask_start:], y[:, mask_start:])
    raise ValueError(kind)


def run(model, nodes, lins, inputs, xs0, loss_fn, T, lr, xopt="sgd", noise_var=None, noise=None,
        acc_begin=None, generator=None, record_energy=True):
    """T steps.  noise_var=None: PC.  noise: optional callable (t, l) -> numpy normals (else normal_).

    Returns (energies [T], losses [T]).  Parameter grads are left in ``lin.weight.grad`` exactly as
    autograd accumulates them (zeroed at t == acc_begin if given)."""
    inputs = torch.as_tensor(inputs)
    for node, x0 in zip(nodes, xs0):
        node.state = nn.Parameter(torch.as_tensor(x0).clone())
    xs = [n.state for n in nodes]
    opt = torch.optim.SGD(xs, lr=lr) if xopt == "sgd" else torch.optim.Adam(xs, lr=lr)
    params = [p for lin in lins for p in lin.parameters()]
    energies, losses = [], []
    for t in range(T):
        out = model(inputs)
        energy = sum(n.energy for n in nodes)
        loss = loss_fn(out) if loss_fn is not None else None
        overall = energy if loss is None else loss + energy
        if record_energy:
            energies.append(energy.item())
            losses.append(0.0 if loss is None else loss.item())
        opt.zero_grad()
        if acc_begin is not None and t == acc_begin:
            for p in params:
                p.grad = None
        overall.backward()
        opt.step()
        if noise_var is not None:
            std = math.sqrt(noise_var / lr)
            for l, x in enumerate(xs):
                if noise is None:
                    x.grad.normal_(0.0, std, generator=generator)
                else:
                    x.grad.copy_(torch.as_tensor(noise(t, l)) * (-std))
            opt.step()
    return energies, losses
