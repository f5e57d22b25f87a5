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
    """Returns (noise_var or None for 'no callback', "") if fusable, else (None, reason) with ok=False signalled by reason.

    Fusable = the callback is a Langevin kick: this package's tagged ``random_step``, or a plain Python function whose CODE can be
    nothing but the reference's ``random_step`` (utils/model.py:35-44: straight-line code over ``get_model_xs()`` /
    ``get_optimizer_x()`` / ``x.grad.normal_`` / ``optimizer.step()``, no branch, no use of ``t``, no closure, no other name --
    ``_static_langevin_check``) AND which behaves like it when probed -- e.g. the user's own unmodified copy of those ten lines.
    Anything else (a kick that also logs, counts, clamps, skips steps ...) keeps the step-wise path, where it is really called."""
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
    why = _static_langevin_check(callback)
    if why:
        return None, f"callback_after_t is not provably a plain Langevin kick ({why})"
    verdict = _probe_langevin_callback(callback, kwargs, trainer)
    if verdict[0] is not None and callback not in _FUSED_ANNOUNCED:
        import warnings
        _FUSED_ANNOUNCED.add(callback)
        warnings.warn(f"callback_after_t {getattr(callback, '__qualname__', callback)!r} (defined in "
                      f"{getattr(callback, '__module__', '?')}) is the Langevin kick random_step (noise_var = {verdict[0]:g}): it is fused into the "
                      "HIP x update (Philox noise) and NOT invoked per step", RuntimeWarning, stacklevel=4)
    return verdict


import weakref                                                     # noqa: E402

_FUSED_ANNOUNCED = weakref.WeakSet()        # (callbacks already announced; weak: a script's closures are not kept alive by a warning)

# What the code of an untagged callback may mention if it is to be fused without ever being called again: the reference's random_step
# (utils/model.py:35-44) uses get_model_xs, get_optimizer_x, grad, normal_, np.sqrt, defaults and step -- nothing else.
_KICK_ATTRS = frozenset({"get_model_xs", "get_optimizer_x", "get_optimizer_x_lr", "grad", "normal_", "sqrt", "defaults", "param_groups",
                         "step", "zero_grad"})
_KICK_MODULES = frozenset({"numpy", "math", "torch"})
_KICK_FORBIDDEN_OPS = ("STORE_GLOBAL", "STORE_ATTR", "STORE_SUBSCR", "DELETE_", "IMPORT_", "MAKE_FUNCTION", "LOAD_CLOSURE", "LOAD_DEREF",
                       "STORE_DEREF", "LOAD_CLASSDEREF", "YIELD", "RAISE", "COMPARE_OP", "CONTAINS_OP", "IS_OP", "POP_JUMP", "JUMP_IF",
                       "SETUP_WITH", "BEFORE_WITH", "SETUP_FINALLY", "LOAD_BUILD_CLASS", "LOAD_NAME", "STORE_NAME", "MATCH_")
# ... and the ONLY opcodes such a function may consist of (a deny-list silently lets through whatever a newer interpreter
# adds -- 3.13's LOAD_FAST_LOAD_FAST, 3.14's LOAD_FAST_BORROW* read locals under other names): straight-line loads, calls, arithmetic,
# the for-loop over get_model_xs() and the return.  Anything else -- an opcode of an interpreter this list was not written for
# included -- fails the check, and the callback keeps the step-wise path, where it is really called.
_KICK_ALLOWED_OPS = frozenset({
    "RESUME", "NOP", "CACHE", "PRECALL", "PUSH_NULL", "COPY", "SWAP", "POP_TOP", "KW_NAMES", "EXTENDED_ARG", "COPY_FREE_VARS",
    "LOAD_FAST", "LOAD_FAST_CHECK", "LOAD_FAST_AND_CLEAR", "LOAD_FAST_LOAD_FAST", "LOAD_FAST_BORROW", "LOAD_FAST_BORROW_LOAD_FAST_BORROW",
    "STORE_FAST", "STORE_FAST_STORE_FAST", "STORE_FAST_LOAD_FAST", "LOAD_CONST", "LOAD_SMALL_INT", "LOAD_GLOBAL", "LOAD_ATTR", "LOAD_METHOD",
    "CALL", "CALL_FUNCTION", "CALL_METHOD", "CALL_FUNCTION_KW", "CALL_KW", "BINARY_OP", "BINARY_SUBSCR", "BINARY_TRUE_DIVIDE", "BINARY_MULTIPLY",
    "BINARY_ADD", "BINARY_SUBTRACT", "BINARY_POWER", "UNARY_NEGATIVE", "BUILD_TUPLE", "BUILD_LIST", "GET_ITER", "FOR_ITER", "END_FOR", "POP_ITER",
    "JUMP_ABSOLUTE", "JUMP_BACKWARD", "JUMP_BACKWARD_NO_INTERRUPT", "JUMP_FORWARD", "RETURN_VALUE", "RETURN_CONST", "NOT_TAKEN", "DUP_TOP",
    "ROT_TWO"})


def _static_langevin_check(callback) -> str:
    """"" if the function's bytecode can only be a plain Langevin kick, else the reason.  A static guarantee, so that fusing the
    callable (never calling it again) cannot drop anything it would have done: a plain function without closure whose code is free
    of conditional branches and comparisons (nothing can depend on t or on a counter), never reads its first parameter (t), loads
    no global but the numpy / math / torch modules, and touches no attribute outside the handful random_step needs."""
    import dis
    import types
    if not isinstance(callback, types.FunctionType):
        return "not a plain Python function"
    if callback.__closure__:
        return "it closes over variables"
    code = callback.__code__
    if code.co_argcount + code.co_kwonlyargcount < 1 or code.co_flags & 0x0C:       # *args / **kwargs
        return "its signature is not (t, _pc_trainer, ...)"
    t_name = code.co_varnames[0]
    for ins in dis.get_instructions(code):
        op = ins.opname
        if any(op.startswith(f) for f in _KICK_FORBIDDEN_OPS):
            return f"its code contains {op} (a branch, comparison, store or import)"
        if op not in _KICK_ALLOWED_OPS:
            return f"its code contains {op}, which a plain Langevin kick has no use for (or this interpreter's bytecode is newer than the check)"
        if op.startswith("LOAD_FAST") or op.startswith("STORE_FAST"):
            # (the fused forms of newer interpreters carry a tuple of names)
            names = ins.argval if isinstance(ins.argval, (tuple, list)) else (ins.argval,)
            if t_name in names:
                return f"it reads its step argument {t_name!r}"
        if op == "LOAD_GLOBAL":
            mod = callback.__globals__.get(ins.argval)
            if not isinstance(mod, types.ModuleType) or mod.__name__.split(".")[0] not in _KICK_MODULES:
                return f"it uses the global {ins.argval!r}"
        if op in ("LOAD_ATTR", "LOAD_METHOD") and ins.argval not in _KICK_ATTRS:
            return f"it touches .{ins.argval}"
    return ""


# ---- behavioural recognition of a Langevin callback -------------------------------------------------------------------------
class _ProbeRefused(Exception):
    pass


class _SpyGrad(torch.Tensor):
    """A gradient tensor that notes how it is filled: ``normal_(mean, std)`` calls are logged with their exact arguments."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if getattr(func, "__name__", "") == "normal_" and args and isinstance(args[0], _SpyGrad):
            mean = args[1] if len(args) > 1 else kwargs.get("mean", 0.0)
            std = args[2] if len(args) > 2 else kwargs.get("std", 1.0)
            log = getattr(args[0], "_spy_log", None)
            if log is not None:
                log.append((float(mean), float(std), kwargs.get("generator") is not None))
        return super().__torch_function__(func, types, args, kwargs)


class _StubOptimizer:
    """What random_step touches of an optimizer: ``defaults['lr']``, ``param_groups``, ``zero_grad()``, ``step()``."""

    def __init__(self, params, lr):
        self.defaults = {"lr": lr}
        self.param_groups = [{"params": params, "lr": lr}]
        self.state = {}
        self.steps = []                       # per step() call: the gradients it saw

    def zero_grad(self, set_to_none=False):
        for p in self.param_groups[0]["params"]:
            if p.grad is not None:
                p.grad.zero_()

    def step(self, closure=None):
        self.steps.append([None if p.grad is None else p.grad.detach().as_subclass(torch.Tensor).clone()
                           for p in self.param_groups[0]["params"]])

    def __getattr__(self, name):
        raise _ProbeRefused(f"the callback uses optimizer_x.{name}")


class _StubTrainer:
    """What random_step touches of a trainer: ``get_model_xs()`` and ``get_optimizer_x()``; anything else refuses the probe."""

    def __init__(self, params, optimizer):
        self._params, self._optimizer = params, optimizer

    def get_model_xs(self, is_warning_x_not_initialized=True):
        return iter(self._params)

    def get_optimizer_x(self):
        return self._optimizer

    def get_optimizer_x_lr(self):
        return self._optimizer.param_groups[0]["lr"]

    def __getattr__(self, name):
        raise _ProbeRefused(f"the callback uses trainer.{name}")


_PROBE_SHAPES = ((64, 48), (32, 40))          # 3072 + 1280 probe elements, two "layers"
_PROBE_CACHE = {}


def _probe_key(callback, kwargs, trainer, lr, T):
    items = []
    for k in sorted(kwargs):
        v = kwargs[k]
        items.append((k, "<trainer>" if v is trainer else (v if isinstance(v, (int, float, str, bool, type(None))) else id(v))))
    return (id(callback), getattr(callback, "__code__", None), tuple(items), lr, T)


def _probe_langevin_callback(callback, kwargs, trainer):
    """Run ``callback(t, **kwargs)`` on a scratch trainer (the kwarg that IS the trainer is replaced by a stub whose latent
    tensors are CPU probes) at t = 0, 1, T // 3, T // 2, T - 2, T - 1 (its code has already been shown not to read t:
    _static_langevin_check) and accept it as a Langevin kick iff, every time, it
      * fills every ``x.grad`` with ``normal_(0, std)`` -- one common std, no private generator -- and leaves it at that
        (the gradients ``optimizer.step()`` sees have the moments of N(0, std^2): nothing rescaled them afterwards),
      * calls ``optimizer.step()`` exactly once, and touches neither the x values nor anything else of the trainer.
    That is the reference's ``random_step`` (utils/model.py:35-44: ``x.grad.normal_(0., sqrt(var / lr)); optimizer.step()``,
    i.e. x <- x - lr * std * xi) whatever it is called and wherever it is defined; the fused kick is
    x <- x + sqrt(noise_var * lr) * xi with noise_var = lr * std^2.  The probe draws from torch's CPU generator; the
    generator states (torch, NumPy, random) are restored afterwards, so a seeded script is not disturbed."""
    import random as _random

    import numpy as np
    xkw = getattr(trainer, "_optimizer_x_kwargs", None)
    lr = xkw.get("lr") if isinstance(xkw, dict) else None
    if not isinstance(lr, (int, float)) or not lr > 0:
        return None, "callback_after_t with an x optimizer that has no positive lr"
    lr = float(lr)
    if not any(v is trainer for v in kwargs.values()):
        return None, "callback_after_t is an arbitrary callable (it does not take the trainer as a keyword argument, so it cannot be probed)"
    T = int(getattr(trainer, "_T", 1))
    key = _probe_key(callback, kwargs, trainer, lr, T)
    if key in _PROBE_CACHE:
        return _PROBE_CACHE[key]
    saved = (torch.get_rng_state(), np.random.get_state(), _random.getstate())
    verdict = None
    try:
        stds = []
        for t in sorted({0, min(1, T - 1), T // 3, T // 2, max(T - 2, 0), T - 1}):
            g = torch.Generator().manual_seed(1234 + t)
            params, logs = [], []
            for shape in _PROBE_SHAPES:
                p = torch.nn.Parameter(torch.randn(*shape, generator=g))
                log = []
                grad = torch.randn(*shape, generator=g).as_subclass(_SpyGrad)
                grad._spy_log = log
                p.grad = grad
                params.append(p); logs.append(log)
            before = [p.detach().clone() for p in params]
            opt = _StubOptimizer(params, lr)
            stub = _StubTrainer(params, opt)
            kw = {k: (stub if v is trainer else v) for k, v in kwargs.items()}
            callback(t, **kw)
            if len(opt.steps) != 1:
                raise _ProbeRefused(f"it calls optimizer_x.step() {len(opt.steps)} times, a Langevin kick calls it once")
            if any(not torch.equal(a, p.detach()) for a, p in zip(before, params)):
                raise _ProbeRefused("it writes to the x values directly")
            for log, seen in zip(logs, opt.steps[0]):
                if len(log) != 1:
                    raise _ProbeRefused("it does not fill every x.grad with exactly one normal_() call")
                mean, std, private_gen = log[0]
                if mean != 0.0 or not std > 0 or private_gen:
                    raise _ProbeRefused("its noise is not N(0, std) from the default generator")
                stds.append(std)
                n = seen.numel()
                m1, m2 = float(seen.mean()) / std, float((seen * seen).mean()) / std ** 2
                m4 = float((seen ** 4).mean()) / std ** 4
                if abs(m1) > 6.0 / n ** 0.5 or abs(m2 - 1.0) > 6.0 * (2.0 / n) ** 0.5 or abs(m4 - 3.0) > 6.0 * (96.0 / n) ** 0.5:
                    raise _ProbeRefused("the gradients its optimizer step sees are not the N(0, std) it drew")
        if max(stds) - min(stds) > 1e-12 * max(stds):
            raise _ProbeRefused("its noise scale differs between layers or steps")
        verdict = (float("%.12g" % (lr * stds[0] ** 2)), "")
    except _ProbeRefused as why:
        verdict = (None, f"callback_after_t is not a plain Langevin kick: {why}")
    except Exception as exc:                   # the callable needs more of a trainer than the stub has
        verdict = (None, f"callback_after_t could not be probed ({type(exc).__name__}: {exc})")
    finally:
        torch.set_rng_state(saved[0]); np.random.set_state(saved[1]); _random.setstate(saved[2])
    if len(_PROBE_CACHE) > 256:
        _PROBE_CACHE.clear()
    _PROBE_CACHE[key] = verdict
    return verdict
