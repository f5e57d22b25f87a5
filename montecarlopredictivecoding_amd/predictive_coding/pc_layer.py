"""PCLayer -- host-side mirror of the reference's latent-state holder.

API parity target: /root/reference/predictive_coding/pc_layer.py:8-304 (constructor arguments,
``get_x`` / ``set_is_sample_x`` / ``energy`` accessors, train-vs-eval forward behaviour, the
resample-with-RuntimeWarning rules of :185-218).

Role in this package: a PCLayer is mostly a *descriptor* (where a latent layer sits, its energy
coefficient, how x is initialised) plus the owner of the ``x`` tensor the HIP engine reads and
writes back.  Its torch ``forward`` is only executed (a) in eval mode (returns ``mu``), (b) once per
call to draw the initial x through the user's ``sample_x_fn`` and (c) by the opt-in generic path.
"""
import typing
import warnings

import torch
import torch.nn as nn


def quadratic_energy(inputs):
    """Default energy: 0.5 * (mu - x)^2 per element (reference pc_layer.py:17-18)."""
    return 0.5 * (inputs["mu"] - inputs["x"]) ** 2


def sample_x_from_mu(inputs):
    """Default initialisation: x starts at its prediction (reference pc_layer.py:19-20)."""
    return inputs["mu"].detach().clone()


def probe_energy_coefficient(energy_fn) -> typing.Optional[float]:
    """Return c if ``energy_fn`` computes c*0.5*(mu-x)^2 element-wise, else None.

    The reference accepts arbitrary callables; the scripts only ever pass the default or
    ``(1/var)*0.5*(mu-x)**2`` closures (figure_3.py:47-48, figure_6.py:80-81).  Rather than trusting
    names, the callable is evaluated on a small random probe and compared with the quadratic form.
    """
    if energy_fn is quadratic_energy:
        return 1.0
    g = torch.Generator().manual_seed(1234)
    mu = torch.randn(5, 7, generator=g, dtype=torch.float64)
    x = torch.randn(5, 7, generator=g, dtype=torch.float64)
    try:
        with torch.no_grad():
            e = energy_fn({"mu": mu, "x": x})
    except Exception:
        return None
    if not isinstance(e, torch.Tensor) or e.shape != mu.shape:
        return None
    base = 0.5 * (mu - x) ** 2
    c = (e / base).median().item()
    if not (c > 0) or not torch.allclose(e, c * base, rtol=1e-9, atol=1e-12):
        return None
    # must not depend on anything but (mu - x): second probe with a shifted pair
    with torch.no_grad():
        e2 = energy_fn({"mu": mu + 3.0, "x": x + 3.0})
    if not torch.allclose(e2, c * base, rtol=1e-9, atol=1e-12):
        return None
    return float(c)


class PCLayer(nn.Module):
    """Marks a latent layer of a predictive-coding network and owns its state ``x``.

    train mode : forward(mu) stores the layer energy and returns ``x`` (not ``mu``), which cuts the
                 autograd graph between layers; x is (re)drawn by ``sample_x_fn`` when flagged.
    eval mode  : forward(mu) returns ``mu`` (plain feed-forward network).
    """

    def __init__(
        self,
        energy_fn: typing.Callable = quadratic_energy,
        sample_x_fn: typing.Callable = sample_x_from_mu,
        S: torch.Tensor = None,
        M: torch.Tensor = None,
        is_holding_error: bool = False,
        is_keep_energy_per_datapoint: bool = False,
    ):
        super().__init__()
        assert callable(energy_fn)
        assert callable(sample_x_fn)
        self._energy_fn = energy_fn
        self._sample_x_fn = sample_x_fn
        self._energy = None
        self.set_S(S)
        self.set_M(M)
        assert isinstance(is_holding_error, bool)
        assert isinstance(is_keep_energy_per_datapoint, bool)
        self.is_holding_error = is_holding_error
        self.is_keep_energy_per_datapoint = is_keep_energy_per_datapoint
        self._energy_per_datapoint = None
        self._is_sample_x = False
        self._x = None
        self._ecoef_cache = "unprobed"
        self._mcpc_sampling_only = False
        self.eval()      # like the reference: a fresh layer is in eval mode (pc_layer.py:104)

    # ---- accessors (names follow the reference) -------------------------------------------------
    def set_M(self, M):
        if M is not None:
            assert isinstance(M, torch.Tensor)
        self._M = M

    def set_S(self, S):
        if S is not None:
            assert isinstance(S, torch.Tensor)
            assert S.dim() == 2
        self._S = S

    def get_is_sample_x(self) -> bool:
        return self._is_sample_x

    def set_is_sample_x(self, is_sample_x: bool) -> None:
        assert isinstance(is_sample_x, bool)
        self._is_sample_x = is_sample_x

    def get_x(self) -> nn.Parameter:
        return self._x

    def energy(self) -> torch.Tensor:
        return self._energy

    def clear_energy(self):
        self._energy = None

    def energy_per_datapoint(self) -> torch.Tensor:
        assert self.is_keep_energy_per_datapoint
        return self._energy_per_datapoint

    def clear_energy_per_datapoint(self):
        assert self.is_keep_energy_per_datapoint
        self._energy_per_datapoint = None

    # ---- engine-facing description ------------------------------------------------------------------
    def energy_coefficient(self) -> typing.Optional[float]:
        """c of c*0.5*(mu-x)^2, or None when the energy is not of that form (engine cannot fuse it)."""
        if self._ecoef_cache == "unprobed":
            self._ecoef_cache = probe_energy_coefficient(self._energy_fn)
        return self._ecoef_cache

    def is_plain(self) -> bool:
        """True when nothing but the quadratic energy is configured (no S/M masks, no extras)."""
        return (self._S is None and self._M is None and not self.is_holding_error
                and not self.is_keep_energy_per_datapoint and self.energy_coefficient() is not None)

    def needs_resample(self, mu: torch.Tensor) -> typing.Optional[str]:
        """The three situations in which the reference silently re-draws x (pc_layer.py:185-218)."""
        if self._x is None:
            return "The <self._x> has not been initialized yet, run with <pc_layer.set_is_sample_x(True)> first. We will do it for you."
        if mu.device != self._x.device:
            return "The device of <self._x> is not consistent with that of <mu>, run with <pc_layer.set_is_sample_x(True)> first. We will do it for you."
        if mu.size() != self._x.size():
            return ("You have changed the shape of this layer, you should do <pc_layer.set_is_sample_x(True) when changing "
                    "the shape of this layer. We will do it for you.")
        return None

    # ---- torch forward (initial sampling, eval mode, generic path) ---------------------------------
    def forward(self, mu: torch.Tensor, energy_fn_additional_inputs: dict = {}) -> torch.Tensor:
        assert isinstance(mu, torch.Tensor)
        assert isinstance(energy_fn_additional_inputs, dict)
        if not self.training:
            return mu
        if not self._is_sample_x:
            why = self.needs_resample(mu)
            if why is not None:
                warnings.warn(why, category=RuntimeWarning)
                self._is_sample_x = True
        if self._is_sample_x:
            drawn = self._sample_x_fn({"mu": mu, "x": self._x})
            self._x = nn.Parameter(drawn.to(mu.device), True)
            self._is_sample_x = False
        if self._mcpc_sampling_only:
            # the trainer's forward that only draws x for an engine call (pc_trainer._initial_state): the layer energy of this forward is
            # never read -- the engine computes the energies of every step itself -- so it is not evaluated (27 of 89 ms of that forward at
            # 6000 chains of cfg-M when the model lives on the CPU)
            return self._x
        x = self._x
        if self._S is not None:
            assert mu.dim() == 2 and x.dim() == 2
            assert self._S.size(0) == mu.size(1) and self._S.size(1) == x.size(1)
            mu = mu.unsqueeze(2).expand(-1, -1, x.size(1))
            x = x.unsqueeze(1).expand(-1, mu.size(1), -1)
        fn_inputs = {"mu": mu, "x": x}
        fn_inputs.update(energy_fn_additional_inputs)
        energy = self._energy_fn(fn_inputs)
        if self._S is not None:
            energy = energy * self._S.unsqueeze(0)
        elif self._M is not None:
            energy = energy * self._M.unsqueeze(0)
        if self.is_keep_energy_per_datapoint:
            self._energy_per_datapoint = energy.sum(dim=list(range(1, energy.dim())), keepdim=False).unsqueeze(1)
        self._energy = energy.sum()
        if self.is_holding_error:
            self.error = (self._x.data - mu).detach().clone()
        return self._x
