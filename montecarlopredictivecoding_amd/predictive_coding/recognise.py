"""Structure recognisers: decide whether a (model, loss, optimizer, callback) combination maps onto
the HIP engine's C ABI, and if so produce the numbers the ABI needs.

The reference accepts arbitrary Python callables everywhere; its scripts use a
handful.  Recognition is by *behaviour* (probing a callable on test tensors) wherever that is cheap,
and by tag (``fn._mcpc``) for this package's own helpers, never by name.
"""
import typing
from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import _lib as L
from .pc_layer import PCLayer

_ACT = {nn.ReLU: L.ACT_RELU, nn.Tanh: L.ACT_TANH, nn.Identity: L.ACT_IDENTITY}


@dataclass
class NetDescription:
    linears: typing.List[nn.Linear]          # L (+1 with read-out)
    pc_layers: typing.List[PCLayer]          # L
    acts: typing.List[int]
    ecoef: typing.List[float]
    sizes: typing.List[int]
    n_in: int
    n_out: int                               # 0: the model ends with a PCLayer

    def key(self, batch, device):
        return (tuple(self.sizes), tuple(self.acts), tuple(self.ecoef), self.n_in, self.n_out, batch, str(device))


def describe_model(model) -> typing.Tuple[typing.Optional[NetDescription], str]:
    """Parse ``Sequential[Linear, PCLayer, (act), ..., Linear (, PCLayer)]`` (reference utils/model.py:54-65,
    figure_2.py:40-44, figure_3.py:50-55).  Returns (description, "") or (None, reason)."""
    if not isinstance(model, nn.Sequential):
        return None, "model is not an nn.Sequential"
    mods = list(model)
    linears, pcs, acts = [], [], []
    i, n = 0, len(mods)
    while i < n:
        if not isinstance(mods[i], nn.Linear):
            return None, f"module {i} ({type(mods[i]).__name__}) where an nn.Linear was expected"
        linears.append(mods[i])
        i += 1
        if i == n:
            break                                   # trailing Linear = read-out
        if not isinstance(mods[i], PCLayer):
            return None, f"module {i} ({type(mods[i]).__name__}) where a PCLayer was expected"
        pcs.append(mods[i])
        i += 1
        if i < n and type(mods[i]) in _ACT:
            acts.append(_ACT[type(mods[i])])
            i += 1
        else:
            acts.append(L.ACT_IDENTITY)
    if not pcs:
        return None, "model has no PCLayer"
    if len(pcs) > L.MAX_LATENT:
        return None, f"{len(pcs)} PCLayers > {L.MAX_LATENT}"
    has_head = len(linears) == len(pcs) + 1
    if not has_head and acts[-1] != L.ACT_IDENTITY:
        return None, "an activation after the final PCLayer has nothing to feed"
    ecoef = []
    for k, p in enumerate(pcs):
        if not p.is_plain():
            return None, f"PCLayer {k} uses S/M masks, error holding, per-datapoint energy or a non-quadratic energy_fn"
        ecoef.append(p.energy_coefficient())
    sizes = [lin.out_features for lin in linears[:len(pcs)]]
    for j in range(1, len(linears)):
        if linears[j].in_features != sizes[j - 1]:
            return None, f"Linear {j} in_features {linears[j].in_features} != previous width {sizes[j - 1]}"
    for lin in linears:
        if lin.weight.dtype != torch.float32:
            return None, "parameters are not float32"
    return NetDescription(linears, pcs, acts, ecoef, sizes, linears[0].in_features,
                          linears[-1].out_features if has_head else 0), ""


@dataclass
class LossDescription:
    kind: int = L.LOSS_NONE
    var: float = 1.0
    mask_start: int = 0
    target: typing.Optional[torch.Tensor] = None
    returns_value: bool = False          # the reference appends results["loss"] only if loss_fn is not None


def _mask_start(n_out, perc):
    k = round(n_out * perc)
    return n_out - k if k > 0 else 0


def describe_loss(loss_fn, loss_fn_kwargs, n_out, batch, device) -> typing.Tuple[typing.Optional[LossDescription], str]:
    """Map a loss callable to (kind, var, mask_start, target) or explain why not."""
    if loss_fn is None:
        return LossDescription(), ""
    if n_out == 0:
        return None, "a loss on the output of a model that ends with a PCLayer is not fused"
    tag = getattr(loss_fn, "_mcpc", None)
    kw = dict(loss_fn_kwargs)
    if isinstance(tag, dict) and "loss" in tag:
        if tag["loss"] == "none":
            return LossDescription(returns_value=True), ""
        target = kw.get("_target")
        if not isinstance(target, torch.Tensor):
            return None, "loss needs a tensor `_target`"
        extra = set(kw) - {"_target", "_var", "perc"}
        if extra:
            return None, f"unsupported loss kwargs {sorted(extra)}"
        ms = _mask_start(n_out, kw.get("perc", 0.5)) if tag["masked"] else 0
        if tag["loss"] == "gaussian":
            var = kw.get("_var")
            if var is None or not float(var) > 0:
                return None, "Gaussian loss needs a positive `_var`"
            return LossDescription(L.LOSS_GAUSSIAN, float(var), ms, target, True), ""
        return LossDescription(L.LOSS_BERNOULLI, 1.0, ms, target, True), ""
    return _probe_loss(loss_fn, kw, n_out, batch, device)


def _probe_loss(loss_fn, kw, n_out, batch, device):
    """Unknown callable: evaluate it and its gradient on a probe and match against the fused family."""
    g = torch.Generator(device="cpu").manual_seed(4321)
    o = (torch.randn(batch, n_out, generator=g) * 2.0).to(device).requires_grad_(True)
    try:
        val = loss_fn(o, **kw)
    except Exception as exc:
        return None, f"loss_fn raised on a probe: {exc!r}"
    if not isinstance(val, torch.Tensor) or val.numel() != 1:
        return None, "loss_fn does not return a scalar tensor"
    if not val.requires_grad:
        if float(val) == 0.0:
            return LossDescription(returns_value=True), ""      # zero_fn-like
        return None, "loss_fn is constant but non-zero"
    (grad,) = torch.autograd.grad(val, o)
    target = kw.get("_target")
    if not isinstance(target, torch.Tensor) or tuple(target.shape) != (batch, n_out):
        return None, "cannot find a `_target` of shape [batch, n_out] in loss_fn_kwargs"
    y = target.to(device=device, dtype=torch.float32)
    od = o.detach()
    nz = (grad != 0).any(dim=0)
    if not bool(nz.any()):
        return LossDescription(returns_value=True), ""
    ms = int(torch.nonzero(nz)[0])
    if bool((~nz[ms:]).any()):
        return None, "loss gradient has zero columns after its first active column"
    ga, oa, ya = grad[:, ms:], od[:, ms:], y[:, ms:]
    # Bernoulli with logits: sigmoid(o) - y
    if torch.allclose(ga, torch.sigmoid(oa) - ya, rtol=1e-4, atol=1e-5):
        ref = nn.functional.binary_cross_entropy_with_logits(oa, ya, reduction="sum")
        if torch.allclose(val.detach(), ref, rtol=1e-4):
            return LossDescription(L.LOSS_BERNOULLI, 1.0, ms, target, True), ""
    # Gaussian: (o - y)/var
    diff = oa - ya
    inv_var = float((ga * diff).sum() / (diff * diff).sum())
    if inv_var > 0 and torch.allclose(ga, inv_var * diff, rtol=1e-4, atol=1e-5):
        ref = 0.5 * inv_var * (diff * diff).sum()
        if torch.allclose(val.detach(), ref, rtol=1e-4):
            return LossDescription(L.LOSS_GAUSSIAN, 1.0 / inv_var, ms, target, True), ""
    return None, "loss_fn is neither Gaussian (fe_fn) nor Bernoulli-with-logits (bernoulli_fn), masked or not"


@dataclass
class XOptDescription:
    kind: int
    lr: float
    betas: typing.Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8


def describe_x_optimizer(fn, kwargs) -> typing.Tuple[typing.Optional[XOptDescription], str]:
    kw = dict(kwargs)
    if "lr" not in kw:
        return None, "optimizer_x_kwargs has no lr"
    lr = float(kw.pop("lr"))
    if not lr > 0:
        return None, "lr must be positive"
    if fn is torch.optim.SGD:
        plain = dict(momentum=0, dampening=0, weight_decay=0, nesterov=False, maximize=False)
        for k, v in kw.items():
            if k in ("foreach", "differentiable", "fused"):
                continue
            if k not in plain or v != plain[k]:
                return None, f"SGD option {k}={v!r} is not fused"
        return XOptDescription(L.XOPT_SGD, lr), ""
    if fn is torch.optim.Adam:
        betas = tuple(kw.pop("betas", (0.9, 0.999)))
        eps = float(kw.pop("eps", 1e-8))
        plain = dict(weight_decay=0, amsgrad=False, maximize=False, capturable=False)
        for k, v in kw.items():
            if k in ("foreach", "differentiable", "fused"):
                continue
            if k not in plain or v != plain[k]:
                return None, f"Adam option {k}={v!r} is not fused"
        return XOptDescription(L.XOPT_ADAM, lr, (float(betas[0]), float(betas[1])), eps), ""
    return None, f"optimizer_x_fn {getattr(fn, '__name__', fn)!r} is not SGD/Adam"


def describe_callback(callback, kwargs, trainer) -> typing.Tuple[typing.Optional[typing.Optional[float]], str]:
    """Returns (noise_var or None for 'no callback', "") if fusable, else (None, reason) with ok=False signalled by reason."""
    if callback is None:
        return None, ""
    tag = getattr(callback, "_mcpc", None)
    if isinstance(tag, dict) and tag.get("langevin"):
        kw = dict(kwargs)
        if kw.pop("_pc_trainer", trainer) is not trainer:
            return None, "random_step is bound to a different trainer"
        var = float(kw.pop("var", 2.0))
        if kw:
            return None, f"unexpected random_step kwargs {sorted(kw)}"
        if var < 0:
            return None, "negative noise variance"
        return var, ""
    return None, "callback_after_t is an arbitrary callable"
