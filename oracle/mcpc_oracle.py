"""CPU restatement of the reference's Langevin / PC inference loop.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does.

Parity status: PINNED.  ``oracle/gen_golden.py`` drives the *imported* reference
(/root/reference/predictive_coding, this container only) through API-legal hooks
and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those vectors on every CPU test run.

What is restated (closed form, no autograd), with the reference lines it follows
-------------------------------------------------------------------------------
* network forward of ``nn.Sequential[Linear, PCLayer, act, ..., Linear(, PCLayer)]``
  where a training-mode PCLayer returns its own ``x`` and not ``mu`` (graph cut):
  /root/reference/predictive_coding/pc_layer.py:235-300, /root/reference/utils/model.py:54-65
* layer energy ``0.5*(mu-x)**2`` summed over batch and units, optionally scaled
  by a constant (figure_3.py:47-55 ``(1/var)*0.5*(mu-x)**2``): pc_layer.py:17-18,266-295
* losses (/root/reference/utils/model.py:17-33): Gaussian ``fe_fn``, Bernoulli
  ``bernoulli_fn`` (BCEWithLogits, sum), the two ``*_mask`` variants acting on the
  last ``round(n0*perc)`` columns, ``zero_fn``/``None``
* ``overall = loss + energy`` and its gradient w.r.t. every x and every parameter,
  i.e. what ``overall.backward()`` leaves in ``.grad``: pc_trainer.py:821-862
* ``optimizer_x.step()`` for ``optim.SGD(lr)`` and ``optim.Adam(lr, betas, eps)``:
  pc_trainer.py:871-877 (optimizer re-created per call: :742-752)
* the Langevin kick of ``random_step``: ``x.grad <- N(0, sqrt(var/lr)); optimizer.step()``
  i.e. ``x <- x - lr*sqrt(var/lr)*xi``: /root/reference/utils/model.py:35-44.  Here the
  noise is *injected* (``noise[t][l]``) and the sign convention is ``x <- x + sqrt(var*lr)*xi``.
* parameter-gradient bookkeeping (zero at ``t==accumulate_p_at[0]`` or at a non-accumulating
  update step, otherwise autograd keeps adding into ``.grad``; division by
  ``len(accumulate_p_at)*B`` or ``B`` at an update step): pc_trainer.py:853-859,904-914
* recording semantics: energies / loss / xs at step t are those of x_t *before* the
  update of step t: pc_trainer.py:768-797
* order inside one step: forward -> record -> backward -> x step -> (p step) -> noise:
  pc_trainer.py:733-918

The optimizer_p step itself (torch.optim on parameters) is NOT restated: the oracle
returns the normalised ``param.grad`` and the caller applies whatever torch optimizer
the reference would (it is stock PyTorch on both sides).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

ACT_IDENTITY, ACT_RELU, ACT_TANH = 0, 1, 2
LOSS_NONE, LOSS_GAUSSIAN, LOSS_BERNOULLI = 0, 1, 2
OPT_SGD, OPT_ADAM = 0, 1


@dataclass
class NetSpec:
    """Chain of L latent (PC) layers, optional read-out head.

    ``lin[j]`` (j = 0..L-1) predicts latent layer j+1 from ``inputs`` (j = 0) or from
    ``act[j-1](x_j)``; ``lin[L]`` is the head (present iff ``has_head``).
    Weights are torch ``nn.Linear`` layout ``[out, in]``; a bias may be ``None``.
    """
    sizes: List[int]                       # n_1 .. n_L
    acts: List[int]                        # activation applied to x_l (l = 1..L) before the next Linear
    W: List[np.ndarray]                    # len L (+1 with head)
    b: List[Optional[np.ndarray]]
    ecoef: List[float] = field(default_factory=list)   # energy coefficient per latent layer (default 1)
    has_head: bool = True

    def __post_init__(self):
        if not self.ecoef:
            self.ecoef = [1.0] * len(self.sizes)

    @property
    def L(self):
        return len(self.sizes)


@dataclass
class LossSpec:
    kind: int = LOSS_NONE
    target: Optional[np.ndarray] = None     # [B, n_out]
    var: float = 1.0                        # Gaussian variance
    mask_start: int = 0                     # first column that contributes (masked losses)


@dataclass
class XOpt:
    kind: int = OPT_SGD
    lr: float = 0.1
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8


def mask_start_from_perc(n_out, perc):
    """Column where ``output[:, -round(n_out*perc):]`` starts (utils/model.py:24-25,31-33)."""
    k = round(n_out * perc)
    return n_out - k if k > 0 else 0    # python: x[:, -0:] is the full slice


def _act(kind, x):
    if kind == ACT_IDENTITY:
        return x
    if kind == ACT_RELU:
        return np.maximum(x, 0)
    if kind == ACT_TANH:
        return np.tanh(x)
    raise ValueError(kind)


def _dact(kind, x):
    if kind == ACT_IDENTITY:
        return np.ones_like(x)
    if kind == ACT_RELU:
        return (x > 0).astype(x.dtype)          # torch threshold_backward: strict
    if kind == ACT_TANH:
        t = np.tanh(x)
        return 1 - t * t
    raise ValueError(kind)


def _linear(a, W, b):
    y = a @ W.T
    if b is not None:
        y = y + b
    return y


def forward(net: NetSpec, inputs, xs, loss: LossSpec):
    """One evaluation at the current x.  Returns a dict with everything a step needs."""
    dt = xs[0].dtype
    L = net.L
    a_prev = inputs.astype(dt)
    acts_in = []            # a_{l-1}: input of lin[l-1]
    errs = []
    energies = []
    for l in range(L):
        acts_in.append(a_prev)
        mu = _linear(a_prev, net.W[l].astype(dt), None if net.b[l] is None else net.b[l].astype(dt))
        d = xs[l] - mu
        c = dt.type(net.ecoef[l])
        errs.append(c * d)
        energies.append(float((dt.type(0.5) * c * d * d).sum(dtype=np.float64)))
        a_prev = _act(net.acts[l], xs[l])
    out = None
    e_out = None
    loss_val = 0.0
    if net.has_head:
        acts_in.append(a_prev)
        out = _linear(a_prev, net.W[L].astype(dt), None if net.b[L] is None else net.b[L].astype(dt))
        e_out = np.zeros_like(out)
        if loss.kind != LOSS_NONE:
            y = loss.target.astype(dt)
            m0 = loss.mask_start
            o = out[:, m0:]
            yy = y[:, m0:]
            if loss.kind == LOSS_GAUSSIAN:
                inv = dt.type(1.0 / loss.var)
                loss_val = float((inv * dt.type(0.5) * (o - yy) ** 2).sum(dtype=np.float64))
                e_out[:, m0:] = inv * (o - yy)
            elif loss.kind == LOSS_BERNOULLI:
                # BCEWithLogits: max(o,0) - o*y + log1p(exp(-|o|))
                lv = np.maximum(o, 0) - o * yy + np.log1p(np.exp(-np.abs(o)))
                loss_val = float(lv.sum(dtype=np.float64))
                e_out[:, m0:] = 1.0 / (1.0 + np.exp(-o)) - yy
            else:
                raise ValueError(loss.kind)
    else:
        # no head: the model output is the last PCLayer's x (pc_layer.py:300)
        out = xs[L - 1]
    return dict(acts_in=acts_in, errs=errs, energies=energies, out=out, e_out=e_out, loss=loss_val)


def x_grads(net: NetSpec, xs, fw):
    """dF/dx_l for every latent layer."""
    L = net.L
    dt = xs[0].dtype
    gs = []
    for l in range(L):
        g = fw["errs"][l].copy()
        if l + 1 < L:
            back = fw["errs"][l + 1] @ net.W[l + 1].astype(dt)
            g -= _dact(net.acts[l], xs[l]) * back
        elif net.has_head:
            back = fw["e_out"] @ net.W[L].astype(dt)
            g += _dact(net.acts[l], xs[l]) * back
        gs.append(g)
    return gs


def p_grads(net: NetSpec, fw):
    """dF/dW_j, dF/db_j (un-normalised sums over the batch) for every Linear."""
    L = net.L
    gW, gb = [], []
    for j in range(L):
        e = fw["errs"][j]
        gW.append(-(e.T @ fw["acts_in"][j]))
        gb.append(None if net.b[j] is None else -e.sum(axis=0))
    if net.has_head:
        e = fw["e_out"]
        gW.append(e.T @ fw["acts_in"][L])
        gb.append(None if net.b[L] is None else e.sum(axis=0))
    return gW, gb


@dataclass
class RunResult:
    xs: List[np.ndarray]                       # final x
    energy: np.ndarray                         # [T] sum of layer energies
    layer_energy: np.ndarray                   # [T, L]
    loss: np.ndarray                           # [T]
    overall: np.ndarray                        # [T]
    rec_xs: dict                               # t -> list of x copies (before the update of step t)
    rec_out: dict                              # t -> model output at x_t
    gW: Optional[List[np.ndarray]]             # param.grad as left by the call (normalised iff a p-step ran)
    gb: Optional[List[Optional[np.ndarray]]]
    p_step_done: bool


def run(net: NetSpec, inputs, xs0: Sequence[np.ndarray], loss: LossSpec, xopt: XOpt, T: int,
        noise=None, noise_var: float = 2.0,
        update_p_at: Sequence[int] = (), accumulate_p_at: Sequence[int] = (),
        record_at: Sequence[int] = (), dtype=np.float32,
        gW_in=None, gb_in=None) -> RunResult:
    """T steps of ``train_on_batch`` (pc_trainer.py:712-981) in closed form.

    ``noise`` : None (PC) or callable ``noise(t, l) -> xi[B, n_l]`` / nested list ``noise[t][l]``.
    ``gW_in/gb_in`` : the ``.grad`` contents carried in from earlier calls (None = empty grads).
    Weight *values* stay fixed: the p-step is applied by the caller from the returned grads
    (it can only occur at a step in ``update_p_at``; the fused engine supports 'last'/'never').
    """
    dt = np.dtype(dtype)
    xs = [np.array(x, dtype=dt, copy=True) for x in xs0]
    inputs = np.asarray(inputs, dtype=dt)
    B = inputs.shape[0]
    L = net.L
    update_p_at = list(update_p_at)
    accumulate_p_at = list(accumulate_p_at)
    record = set(record_at)
    lr = dt.type(xopt.lr)
    energy = np.zeros(T); loss_a = np.zeros(T); layer_energy = np.zeros((T, L))
    rec_xs, rec_out = {}, {}
    gW = None if gW_in is None else [g.astype(dt).copy() for g in gW_in]
    gb = None if gb_in is None else [None if g is None else g.astype(dt).copy() for g in gb_in]
    p_step_done = False
    if xopt.kind == OPT_ADAM:
        m = [np.zeros_like(x) for x in xs]
        v = [np.zeros_like(x) for x in xs]
    for t in range(T):
        fw = forward(net, inputs, xs, loss)
        layer_energy[t] = fw["energies"]
        energy[t] = sum(fw["energies"])
        loss_a[t] = fw["loss"]
        if t in record:
            rec_xs[t] = [x.copy() for x in xs]
            rec_out[t] = np.array(fw["out"], copy=True)
        # -- parameter-gradient bookkeeping (pc_trainer.py:853-862)
        zero_p = ((t in update_p_at) and (t not in accumulate_p_at)) or \
                 (len(accumulate_p_at) > 0 and t == accumulate_p_at[0])
        cW, cb = p_grads(net, fw)
        if zero_p or gW is None:
            gW, gb = cW, cb
        else:
            gW = [a + c for a, c in zip(gW, cW)]
            gb = [None if a is None else a + c for a, c in zip(gb, cb)]
        # -- x step (pc_trainer.py:871-877)
        gs = x_grads(net, xs, fw)
        if xopt.kind == OPT_SGD:
            for l in range(L):
                xs[l] = xs[l] - lr * gs[l]
        else:
            # torch passes python doubles (1-beta) as scalar arguments, rounded to fp32 once
            b2 = dt.type(xopt.beta2)
            omb1, omb2 = dt.type(1.0 - xopt.beta1), dt.type(1.0 - xopt.beta2)
            step = t + 1
            bc1 = 1.0 - xopt.beta1 ** step
            bc2s = np.sqrt(1.0 - xopt.beta2 ** step)
            step_size = dt.type(xopt.lr / bc1)
            for l in range(L):
                m[l] = m[l] + (gs[l] - m[l]) * omb1                    # exp_avg.lerp_(grad, 1-beta1)
                v[l] = v[l] * b2 + (omb2 * gs[l]) * gs[l]             # mul_(beta2).addcmul_(g, g, 1-beta2): (alpha * t1) * t2
                denom = np.sqrt(v[l]) / dt.type(bc2s) + dt.type(xopt.eps)
                # param.addcdiv_(exp_avg, denom, value=-step_size): ATen computes self + (alpha * t1) / t2 -- the scalar first
                xs[l] = xs[l] + (-step_size * m[l]) / denom
        # -- p step normalisation (pc_trainer.py:904-914); optimizer_p itself is the caller's
        if t in update_p_at:
            div = dt.type(len(accumulate_p_at) * B if len(accumulate_p_at) > 0 else B)
            gW = [g / div for g in gW]
            gb = [None if g is None else g / div for g in gb]
            p_step_done = True
        # -- Langevin kick (utils/model.py:35-44)
        if noise is not None:
            s = dt.type(np.sqrt(noise_var * xopt.lr))
            for l in range(L):
                xi = noise(t, l) if callable(noise) else noise[t][l]
                xs[l] = xs[l] + s * np.asarray(xi, dtype=dt)
    return RunResult(xs=xs, energy=energy, layer_energy=layer_energy, loss=loss_a,
                     overall=energy + loss_a, rec_xs=rec_xs, rec_out=rec_out,
                     gW=gW, gb=gb, p_step_done=p_step_done)
