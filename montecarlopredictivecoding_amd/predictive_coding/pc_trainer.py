"""PCTrainer -- host-side mirror of the reference trainer, driving the HIP engine.

API parity target: /root/reference/predictive_coding/pc_trainer.py:22-1108 -- constructor keywords
(:27-49), getters (:268-461), ``train_on_batch`` keywords and the ``results`` dict (:500-524,
:682-694, :768-836), schedule strings (:1068-1108), assertion / warning behaviour (:144-264,
:609-655, :199-220).

How a call is executed
----------------------
``train_on_batch`` first *recognises* the call (``recognise.py``).  Three outcomes on the engine, one beside it:

fused     the whole T-step loop runs inside ``mcpc_run`` (libmcpc.so): network
          Sequential[Linear, PCLayer, act, ...], quadratic energies, Gaussian / Bernoulli / masked /
          no loss, SGD or Adam on x at every step, ``update_p_at`` in {'never','last'}, and either no
          callback or a Langevin kick -- this package's tagged ``random_step`` or any callable that behaves
          like the reference's (utils/model.py:35-44), e.g. a script's own unmodified copy -- fused as Philox noise.
          This is every call pattern of the reference's scripts.
stepwise  same network/loss, but arbitrary callbacks, ``update_p_at='all'``, custom x optimizers
          (any torch optimizer), dynamic x-lr ...: per step the HIP kernel produces dF/dx and
          dF/dtheta (``update_x=0``), the reference's control flow around them is replayed with the
          user's torch optimizers and callbacks.  A RuntimeWarning names the reason.
generic   anything the kernels do not express (S/M masks, non-quadratic ``energy_fn``, per-datapoint energies,
          ``loss_x_fn``, optimised / unwrapped inputs, ``backward_kwargs``, an ``early_stop_condition``, models
          that are not the Sequential chain ...: SURVEY section 8b, "must work, need not be fast") runs on the
          package's own torch-autograd restatement of the reference's step (``generic_loop.py``) and says so with
          a RuntimeWarning naming the reason -- never silently, and never for a call the engine can run.
staged    (round 5) a model built on the CPU -- how most of the reference's call sites are written (figure_2.py:29-75,
          figure_4.py:537, figure_5.py:25-27, figure_6.py:55-93: ``use_cuda = False`` or no ``.cuda()`` at all) -- is not refused:
          W, b, x, inputs and the target (a few MB) are copied to the MI355X, the SAME fused / step-wise HIP path runs there,
          and x, the results and ``param.grad`` come back to the CPU tensors the script holds (``plan["staged"]``; one
          RuntimeWarning per trainer).  The computation never runs on the CPU: without a visible HIP device the call raises.
rejected  any call when no HIP device is visible (there is no CPU path: ``MCPCLibraryError``) and ``plot_progress`` (out of scope).
"""
import collections
import threading
import os
import typing
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from .. import _lib as L
from .. import dist
from ..engine import Engine
from . import recognise
from .generic_loop import run_generic
from .pc_layer import PCLayer
from .utils import slow_down_warning


# Philox step counter shared by every trainer of the process: consecutive calls never reuse noise,
# and all ranks of a sharded job (same sequence of calls) stay in lock-step.
_PHILOX_STEPS = [0]
# engines are keyed by (network shape, batch, device, tuning) and shared between trainers -- the scripts build a PC and an
# MCPC trainer over the same nn.Sequential, and figure_4 / table_1 run several models of one architecture side by side.
# WHICH parameters an engine is bound to is therefore tracked on the engine (``_bound_sig``), never on a trainer.  The
# cache is a small LRU: every distinct (shape, batch) -- e.g. the ragged last batch of a data loader -- owns device
# memory, evicted engines are closed.
_ENGINES = collections.OrderedDict()
_ENGINE_CACHE_SIZE = 8


class _few_cpu_threads:
    """Scoped cap on torch's intra-op CPU threads while a STAGED call runs (a model on the CPU, round 5).

    The CPU-side work of such a call -- the model's forward that draws x, the copies, the script's optimizer_p.step() -- is small, but on a
    many-core host torch runs it on every core, and the OpenMP workers then SPIN-wait for the next parallel region (libgomp / libiomp default)
    on all of them: the HIP runtime's completion handling starves, and the very same kernels take 52 instead of 14.5 ms per recipe iteration
    on a 256-thread host (measured, scripts/staging_cost.py; with 8 threads 14.8).  Capped for the duration of the call, restored on exit."""
    CAP = 8
    # `torch.set_num_threads` is process-global: nested staged calls (a callback that trains another model) and trainers on several Python
    # threads share ONE cap -- the outermost entry saves the caller's value, the last exit restores it, and only if nobody else has changed
    # the setting meanwhile (a script that calls set_num_threads inside a callback keeps what it set).  A side effect the staging
    # warning states: the user's callbacks and optimizer_p.step() inside a staged call run under the cap too.
    _lock = threading.Lock()
    _depth = 0
    _saved = None

    def __init__(self, active):
        self.active, self.entered = bool(active), False

    def __enter__(self):
        if self.active:
            cls = _few_cpu_threads
            with cls._lock:
                if cls._depth == 0:
                    n = torch.get_num_threads()
                    if n > cls.CAP:
                        cls._saved = n
                        torch.set_num_threads(cls.CAP)
                    else:
                        cls._saved = None
                cls._depth += 1
                self.entered = True
        return self

    def __exit__(self, *exc):
        if self.entered:
            cls = _few_cpu_threads
            with cls._lock:
                cls._depth -= 1
                if cls._depth == 0 and cls._saved is not None:
                    if torch.get_num_threads() == cls.CAP:          # still the cap this context set: nobody chose another value meanwhile
                        torch.set_num_threads(cls._saved)
                    cls._saved = None
            self.entered = False
        return False


def _take_philox_steps(n):
    base = _PHILOX_STEPS[0]
    _PHILOX_STEPS[0] += int(n)
    return base


class PCTrainer(object):
    """Trainer for predictive-coding networks built from :class:`PCLayer`."""

    # tests/test_generic_loop.py sets this on an instance to check generic_loop.py against the reference's fixtures in the CPU suite;
    # nothing else does: a model on the CPU raises (there is no CPU path)
    _test_only_generic_on_cpu = False

    def __init__(
        self,
        model: nn.Module,
        optimizer_x_fn: typing.Callable = optim.SGD,
        optimizer_x_kwargs: dict = {"lr": 0.1},
        manual_optimizer_x_fn: typing.Callable = None,
        x_lr_amplifier: float = 1.0,
        x_lr_discount: float = 1.0,
        loss_x_fn: typing.Callable = None,
        loss_inputs_fn: typing.Callable = None,
        optimizer_p_fn: typing.Callable = optim.Adam,
        optimizer_p_kwargs: dict = {"lr": 0.001},
        manual_optimizer_p_fn: typing.Callable = None,
        T: int = 512,
        update_x_at: typing.Union[str, typing.List[int]] = "all",
        update_p_at: typing.Union[str, typing.List[int]] = "all",
        accumulate_p_at: typing.Union[str, typing.List[int]] = "never",
        energy_coefficient: float = 1.0,
        early_stop_condition: str = "False",
        update_p_at_early_stop: bool = True,
        plot_progress_at: typing.Union[str, typing.List[int]] = "all",
        is_disable_warning_energy_from_different_batch_sizes: bool = False,
    ):
        assert isinstance(model, nn.Module)
        self._model = model
        assert callable(optimizer_x_fn)
        assert isinstance(optimizer_x_kwargs, dict)
        self._optimizer_x_fn, self._optimizer_x_kwargs = optimizer_x_fn, optimizer_x_kwargs
        if manual_optimizer_x_fn is not None:
            assert callable(manual_optimizer_x_fn)
        self._manual_optimizer_x_fn = manual_optimizer_x_fn
        self._optimizer_x = None
        assert isinstance(x_lr_discount, float) and x_lr_discount <= 1.0
        assert isinstance(x_lr_amplifier, float) and x_lr_amplifier >= 1.0
        self._x_lr_discount, self._x_lr_amplifier = x_lr_discount, x_lr_amplifier
        for fn, what in ((loss_x_fn, "loss_x_fn"), (loss_inputs_fn, "loss_inputs_fn")):
            if fn is not None:
                assert callable(fn)
                assert self.get_is_model_has_pc_layers(), f"<{what}> should only work with models with <PCLayer>. "
        self._loss_x_fn, self._loss_inputs_fn = loss_x_fn, loss_inputs_fn
        assert callable(optimizer_p_fn)
        assert isinstance(optimizer_p_kwargs, dict)
        self._optimizer_p_fn, self._optimizer_p_kwargs = optimizer_p_fn, optimizer_p_kwargs
        if manual_optimizer_p_fn is not None:
            assert callable(manual_optimizer_p_fn)
        self._manual_optimizer_p_fn = manual_optimizer_p_fn
        self.recreate_optimize_p()
        assert isinstance(T, int) and T > 0
        self._T = T
        if self.get_is_model_has_pc_layers():
            if self._T < self.get_num_pc_layers() + 1:
                warnings.warn(
                    "You should always choose T such that T >= (<pc_trainer.get_num_pc_layers()> + 1), "
                    "as it ensures that the error can be PC-propagated through the network.", category=RuntimeWarning)
            min_t = self.get_least_T()
            if self._T < min_t:
                warnings.warn(
                    f"If you have one pc_layer per layer, T={self._T} is too small. Please use a minimum T of {min_t}, "
                    "which is just enough to PC-propagate the error through the network and have all weigths updated "
                    "based on these PC-propagated errors. In practice, you normally should have T much larger than this minimum T. ",
                    category=RuntimeWarning)
        self._update_x_at = self._preprocess_step_index_list(indices=update_x_at, T=self._T)
        self._update_p_at = self._preprocess_step_index_list(indices=update_p_at, T=self._T)
        self._accumulate_p_at = self._preprocess_step_index_list(indices=accumulate_p_at, T=self._T)
        assert isinstance(energy_coefficient, float)
        self._energy_coefficient = energy_coefficient
        assert isinstance(early_stop_condition, str)
        self._early_stop_condition = early_stop_condition
        assert isinstance(update_p_at_early_stop, bool)
        self._update_p_at_early_stop = update_p_at_early_stop
        if isinstance(plot_progress_at, str):
            assert plot_progress_at in ["all"]
        elif isinstance(plot_progress_at, list):
            for h in plot_progress_at:
                assert isinstance(h, int)
        else:
            raise NotImplementedError
        self._plot_progress_at = plot_progress_at
        self._is_plot_progress = not (isinstance(plot_progress_at, list) and len(plot_progress_at) == 0)
        assert isinstance(is_disable_warning_energy_from_different_batch_sizes, bool)
        self.is_disable_warning_energy_from_different_batch_sizes = is_disable_warning_energy_from_different_batch_sizes

        # ---- engine-side state (no counterpart in the reference) ------------------------------------
        self.mcpc_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF   # Philox key; follows torch.manual_seed
        self.mcpc_chain_base = 0              # global id of this shard's first chain
        self.mcpc_process_group = None        # torch.distributed group for the Hebbian all-reduce (or None)
        self.mcpc_world_batch = None          # global batch for the 1/(n*B) normalisation when sharded
        self.mcpc_sharded = False
        self.mcpc_reduce_results = False      # set_shard(reduce_results=True): results are those of the whole batch
        self.mcpc_materialize_unused_grads = False   # reference quirk: autograd fills .grad even if never used
        self.last_call_mode = None            # 'fused' | 'stepwise' | 'generic' (for tests / diagnostics)
        # trajectories larger than this (bytes of host-bound records per call) are recorded slice by slice into a
        # two-buffer device ring that is drained to pinned host memory while the next slice runs (figure_5 pulls
        # 10 000 steps x all latents to the host, pc_trainer.py:440-445,772-774)
        self.mcpc_record_chunk_bytes = 1 << 30
        self.last_record_slices = 0

    # ---- getters & setters (reference :268-461) -------------------------------------------------------
    def get_T(self) -> int:
        return self._T

    def get_model(self) -> nn.Module:
        return self._model

    def get_optimizer_x(self) -> optim.Optimizer:
        return self._optimizer_x

    def get_optimizer_x_lr(self):
        for group in self._optimizer_x.param_groups:
            return group["lr"]

    def set_optimizer_x(self, optimizer_x: optim.Optimizer) -> None:
        assert isinstance(optimizer_x, optim.Optimizer)
        self._optimizer_x = optimizer_x

    def set_optimizer_x_lr(self, lr: float) -> None:
        for group in self._optimizer_x.param_groups:
            group["lr"] = lr

    def get_optimizer_p(self) -> optim.Optimizer:
        return self._optimizer_p

    def set_optimizer_p(self, optimizer_p: optim.Optimizer) -> None:
        assert isinstance(optimizer_p, optim.Optimizer)
        self._optimizer_p = optimizer_p

    def get_model_pc_layers(self) -> typing.Generator[PCLayer, None, None]:
        for module in self._model.modules():
            if isinstance(module, PCLayer):
                yield module

    def get_named_model_pc_layers(self):
        for name, module in self._model.named_modules():
            if isinstance(module, PCLayer):
                yield name, module

    def get_is_model_has_pc_layers(self) -> bool:
        return any(True for _ in self.get_model_pc_layers())

    def get_model_pc_layers_training(self) -> list:
        return [layer.training for layer in self.get_model_pc_layers()]

    def get_is_model_training(self):
        flags = self.get_model_pc_layers_training()
        if self._model.training and all(flags):
            return True
        if (not self._model.training) and not any(flags):
            return False
        return None

    def get_energies(self, is_per_datapoint: bool = False, named_layers: bool = False):
        energies, batch_sizes = {}, []
        for name, layer in self.get_named_model_pc_layers():
            energy = layer.energy_per_datapoint() if is_per_datapoint else layer.energy()
            if energy is not None:
                energies[name] = energy
                batch_sizes.append(energy.size(0) if is_per_datapoint else energy.size())
        assert len(energies) > 0, "You don't have any pc_layers or none of them is holding energy. "
        if (not self.is_disable_warning_energy_from_different_batch_sizes) and batch_sizes.count(batch_sizes[0]) != len(batch_sizes):
            warnings.warn(f"You pc_layers hold energy of different batch_sizes: {batch_sizes}.", category=RuntimeWarning)
        return energies if named_layers else list(energies.values())

    def get_model_xs(self, is_warning_x_not_initialized=True) -> typing.Generator[nn.Parameter, None, None]:
        for layer in self.get_model_pc_layers():
            x = layer.get_x()
            if x is not None:
                yield x
            elif is_warning_x_not_initialized:
                warnings.warn(
                    "While you are getting x from all pc layers (calling <pc_trainer.get_model_xs()>), some pc layers has "
                    "not been initialized yet (i.e., has x being None). This potentially causes bugs. ", category=RuntimeWarning)

    def get_model_parameters(self) -> typing.Generator[nn.Parameter, None, None]:
        xs = list(self.get_model_xs(is_warning_x_not_initialized=False))
        for param in self._model.parameters():
            if not any(param is x for x in xs):
                yield param

    def get_numparameters(self, is_gen=True):
        params = list(self.get_model_parameters())
        return sum(p.numel() for i, p in enumerate(params) if not (is_gen and i == 0))

    def get_weights_norms(self):
        weights_abs, mu_abs = [], []
        for par in self.get_model_parameters():
            (mu_abs if par.dim() == 1 else weights_abs).append(par.abs().mean())
        return weights_abs, mu_abs

    def get_model_representations(self):
        return self._model[1].get_x()          # reference hard-codes model[1] (pc_trainer.py:437-438)

    def get_model_xs_copy(self):
        return [x.clone().detach().cpu() for x in self.get_model_xs()]

    def get_num_pc_layers(self) -> int:
        return sum(1 for _ in self.get_model_pc_layers())

    def get_least_T(self) -> int:
        return self.get_num_pc_layers() + 1

    def recreate_optimize_x(self) -> None:
        if self._manual_optimizer_x_fn is None:
            self._optimizer_x = self._optimizer_x_fn(self.get_model_xs(), **self._optimizer_x_kwargs)
        else:
            self._optimizer_x = self._manual_optimizer_x_fn()

    def recreate_optimize_p(self) -> None:
        if self._manual_optimizer_p_fn is None:
            self._optimizer_p = self._optimizer_p_fn(self.get_model_parameters(), **self._optimizer_p_kwargs)
        else:
            self._optimizer_p = self._manual_optimizer_p_fn()

    def reset_plot_progress(self):
        """API parity (reference pc_trainer.py:489-498): the plotting tallies; plotting itself is out of scope (plot_progress_at=[])."""
        self._h = 0
        self._plot_progress = {"key": [], "h": [], "t": [], "value": []}

    # ---- distributed sharding (no counterpart in the reference: it is single-device) -------------------
    def set_shard(self, process_group=None, chain_base: int = 0, world_batch: typing.Optional[int] = None,
                  reduce_results: bool = False):
        """Declare that this trainer holds one shard of a larger batch of chains.

        ``chain_base``  global index of the first local chain (keeps Philox noise independent of the sharding),
        ``world_batch`` total number of chains over all shards (the reference divides grads by ``len(inputs)``); ``None``:
                        the local batch travels with the gradient bucket as one extra float of its all-reduce (fused calls: no
                        further collective and no host synchronisation; never cached on a rank-local value -- shards may be uneven
                        and the last batch of a data loader ragged),
        ``process_group`` the group whose members' Hebbian sums are all-reduced once per learning call,
        ``reduce_results`` True: ``results["loss" / "energy" / "overall"]`` are those of the WHOLE batch, as in the reference
                        (pc_layer.py:295 sums the energy over the batch; pc_trainer.py:785-797): one more all-reduce per call, of the
                        [T, L + 2] fp64 table.  False (default): every rank gets its shard's partial sums.
        """
        self.mcpc_process_group = process_group
        self.mcpc_chain_base = int(chain_base)
        if world_batch is not None and int(world_batch) < 1:
            raise ValueError("world_batch must be positive")
        self.mcpc_world_batch = None if world_batch is None else int(world_batch)
        self.mcpc_sharded = True
        self.mcpc_reduce_results = bool(reduce_results)

    def _global_batch(self, local_batch: int) -> int:
        """Batch the reference divides by (``len(inputs)``, pc_trainer.py:905): the local one for an unsharded trainer,
        the job-wide one for a shard -- given to set_shard, or else the sum of the local batches over the group."""
        if not self.mcpc_sharded:
            return local_batch
        if self.mcpc_world_batch is not None:
            return self.mcpc_world_batch
        # a collective: every rank of the group reaches this line once per learning call (same trainer schedule on all ranks)
        return dist.sum_over_group(local_batch, self.mcpc_process_group)

    # ---- the call --------------------------------------------------------------------------------------
    def train_on_batch(
        self,
        inputs: typing.Any,
        loss_fn: typing.Callable = None,
        loss_fn_kwargs: dict = {},
        is_sample_x_at_batch_start: bool = True,
        is_reset_optimizer_x_at_batch_start: bool = True,
        is_reset_optimizer_p_at_batch_start: bool = False,
        is_unwrap_inputs: bool = False,
        is_optimize_inputs: bool = False,
        callback_after_backward: typing.Callable = None,
        callback_after_backward_kwargs: dict = {},
        callback_after_t: typing.Callable = None,
        callback_after_t_kwargs: dict = {},
        is_log_progress: bool = True,
        is_return_results_every_t: bool = True,
        is_checking_after_callback_after_t: bool = True,
        debug: dict = {},
        backward_kwargs: dict = {},
        is_clear_energy_after_use: bool = False,
        is_return_outputs: bool = False,
        is_return_representations: bool = False,
        is_return_xs: bool = False,
        is_return_batchelement_loss: bool = False,
    ):
        """Run T inference steps on one batch (reference pc_trainer.py:500-1064).  Returns the results dict."""
        self.inputs = inputs
        assert self.get_is_model_training() == True, (  # noqa: E712  (three-valued: True / False / None)
            "PCLayer behaves differently in train and eval modes, like Dropout or Batch Normalization. "
            "Make sure your model is in train mode before calling <train_on_batch()>. It can be done by calling <model.train()>. ")
        if loss_fn is not None:
            assert callable(loss_fn)
        assert isinstance(loss_fn_kwargs, dict)
        for flag in (is_sample_x_at_batch_start, is_reset_optimizer_x_at_batch_start, is_reset_optimizer_p_at_batch_start,
                     is_unwrap_inputs, is_optimize_inputs, is_log_progress, is_return_results_every_t,
                     is_return_outputs, is_return_representations, is_return_xs):
            assert isinstance(flag, bool)
        if is_unwrap_inputs:
            assert isinstance(inputs, (tuple, list, dict))
        if is_optimize_inputs:
            assert self.get_is_model_has_pc_layers(), "<is_optimize_inputs> should only work with models with <PCLayer>. "
            assert not is_unwrap_inputs
        for cb in (callback_after_backward, callback_after_t):
            if cb is not None:
                assert callable(cb)
        assert isinstance(callback_after_backward_kwargs, dict)
        assert isinstance(callback_after_t_kwargs, dict)
        assert isinstance(debug, dict)
        if is_log_progress:
            slow_down_warning("PCTrainer.train_on_batch", "is_log_progress", "False")
        if self._is_plot_progress:
            raise NotImplementedError(
                "plot_progress is a plotting feature of the reference (seaborn/pandas PNGs behind an input() prompt) and is out "
                "of scope of this engine: construct the trainer with plot_progress_at=[] as every reference script does.")
        if is_return_results_every_t:
            slow_down_warning("PCTrainer.train_on_batch", "is_return_results_every_t", "False")

        plan, why_not_fused = self._plan(inputs, loss_fn, loss_fn_kwargs, is_unwrap_inputs, is_optimize_inputs,
                                         callback_after_backward, callback_after_t, callback_after_t_kwargs,
                                         backward_kwargs, is_clear_energy_after_use, is_return_batchelement_loss)
        if plan is None:
            # outside what the kernels express: the package's generic torch loop (generic_loop.py), loudly, on the device the model
            # lives on ("must work, need not be fast").  It is no way around a missing GPU: without a visible HIP device
            # every call raises, whatever path it would have taken.
            if not torch.cuda.is_available() and not self._test_only_generic_on_cpu:
                dev = next((p.device for p in self._model.parameters()), None)
                raise L.MCPCLibraryError(
                    "no HIP device is visible (the model lives on %s): the MCPC engine runs on an MI355X only; "
                    "there is no CPU path" % dev)
            self.last_call_mode = "generic"
            return run_generic(
                self, why_not_fused, inputs=inputs, loss_fn=loss_fn, loss_fn_kwargs=loss_fn_kwargs,
                is_sample_x_at_batch_start=is_sample_x_at_batch_start,
                is_reset_optimizer_x_at_batch_start=is_reset_optimizer_x_at_batch_start,
                is_reset_optimizer_p_at_batch_start=is_reset_optimizer_p_at_batch_start, is_unwrap_inputs=is_unwrap_inputs,
                is_optimize_inputs=is_optimize_inputs, callback_after_backward=callback_after_backward,
                callback_after_backward_kwargs=callback_after_backward_kwargs, callback_after_t=callback_after_t,
                callback_after_t_kwargs=callback_after_t_kwargs, is_return_results_every_t=is_return_results_every_t,
                is_checking_after_callback_after_t=is_checking_after_callback_after_t, backward_kwargs=backward_kwargs,
                is_clear_energy_after_use=is_clear_energy_after_use, is_return_outputs=is_return_outputs,
                is_return_representations=is_return_representations, is_return_xs=is_return_xs,
                is_return_batchelement_loss=is_return_batchelement_loss)
        common = dict(inputs=inputs, loss_fn=loss_fn, is_sample_x_at_batch_start=is_sample_x_at_batch_start,
                      is_reset_optimizer_x_at_batch_start=is_reset_optimizer_x_at_batch_start,
                      is_reset_optimizer_p_at_batch_start=is_reset_optimizer_p_at_batch_start,
                      is_return_results_every_t=is_return_results_every_t, is_return_outputs=is_return_outputs,
                      is_return_representations=is_return_representations, is_return_xs=is_return_xs)
        if (plan["mode"] == "fused" and plan["xopt"].kind == L.XOPT_ADAM and self._optimizer_x is not None
                and not is_sample_x_at_batch_start and not is_reset_optimizer_x_at_batch_start):
            # the reference keeps the Adam moments and step count of optimizer_x across calls in this case
            # (pc_trainer.py:742-752 recreates it only behind one of the two flags); the fused kernel restarts them, so the
            # call is replayed step-wise with the trainer's own persistent torch optimizer
            plan["mode"] = "stepwise"
            plan["why_stepwise"] = "Adam state of optimizer_x carried over from the previous call"
        if plan["mode"] == "fused":
            self.last_call_mode = "fused"
            with _few_cpu_threads(plan["staged"]):
                return self._run_fused(plan, **common)
        self.last_call_mode = "stepwise"
        # the reference warns about everything that slows a call down (utils.py:8-16); leaving the fused loop is the one that matters here
        warnings.warn(
            "In PCTrainer.train_on_batch, this call leaves the fused HIP loop and runs step by step (T kernel launches with the "
            "reference's control flow, optimizers and callbacks replayed on the host between them), this will slow down training. "
            "Reason: {}. ".format(plan["why_stepwise"]), category=RuntimeWarning)
        with _few_cpu_threads(plan["staged"]):
            return self._run_stepwise(plan, callback_after_backward=callback_after_backward,
                                      callback_after_backward_kwargs=callback_after_backward_kwargs,
                                      callback_after_t=callback_after_t, callback_after_t_kwargs=callback_after_t_kwargs,
                                      is_checking_after_callback_after_t=is_checking_after_callback_after_t, **common)

    # ---- recognition ------------------------------------------------------------------------------------
    def _plan(self, inputs, loss_fn, loss_fn_kwargs, is_unwrap_inputs, is_optimize_inputs, callback_after_backward,
              callback_after_t, callback_after_t_kwargs, backward_kwargs, is_clear_energy_after_use,
              is_return_batchelement_loss):
        """Returns (plan dict, "") or (None, reason: the call runs on the generic loop).  plan['mode'] is 'fused' or 'stepwise'."""
        net, why = recognise.describe_model(self._model)
        if net is None:
            return None, why
        if is_unwrap_inputs or is_optimize_inputs:
            return None, "is_unwrap_inputs / is_optimize_inputs"
        if self._loss_x_fn is not None or self._loss_inputs_fn is not None:
            return None, "loss_x_fn / loss_inputs_fn"
        if not self._energy_coefficient > 0.0:
            return None, "energy_coefficient <= 0"
        if self._early_stop_condition.strip() != "False":
            return None, "an early_stop_condition other than 'False'"
        if backward_kwargs or is_clear_energy_after_use or is_return_batchelement_loss:
            return None, "backward_kwargs / is_clear_energy_after_use / is_return_batchelement_loss"
        if self._energy_coefficient != 1.0:
            # overall = loss + energy * c (pc_trainer.py:821-836): every layer's c_l scaled by c -- errors, x gradients and Hebbian
            # sums follow -- and the energy the results report divided by c again (_collect_results)
            import dataclasses
            net = dataclasses.replace(net, ecoef=[float(np.float32(c_l) * np.float32(self._energy_coefficient)) for c_l in net.ecoef])
        if not isinstance(inputs, torch.Tensor) or inputs.dim() != 2 or inputs.shape[1] != net.n_in:
            return None, f"inputs must be a [batch, {net.n_in}] tensor"
        if inputs.dtype != torch.float32:
            return None, "inputs must be float32"
        model_device = net.linears[0].weight.device
        if any(p.device != model_device for lin in net.linears for p in lin.parameters()):
            return None, "the model's parameters live on several devices"
        staged = model_device.type != "cuda"
        if staged:
            # a model built on the CPU (figure_2.py:29-75, figure_4.py:537, figure_5.py:25-27, figure_6.py:55-93): its tensors are
            # staged onto the MI355X for the call and the results written back -- the computation itself has no CPU form
            if not torch.cuda.is_available():
                raise L.MCPCLibraryError(
                    "the model lives on %s and no HIP device is visible: the MCPC engine runs on an MI355X only; "
                    "there is no CPU path" % model_device)
            device = torch.device("cuda", torch.cuda.current_device())
        else:
            device = model_device
        if inputs.device != model_device:
            return None, f"inputs on {inputs.device}, model on {model_device}"
        B = inputs.shape[0]
        loss, why = recognise.describe_loss(loss_fn, loss_fn_kwargs, net.n_out, B, model_device)
        if loss is None:
            return None, why
        if loss.target is not None:
            if tuple(loss.target.shape) != (B, net.n_out):
                return None, f"_target must have shape {(B, net.n_out)}"
        # plan["device"]: where the engine runs; plan["model_device"]: where the script's tensors live (the same unless staged)
        plan = dict(net=net, loss=loss, B=B, device=device, model_device=model_device, staged=staged)
        # ---- can the whole loop be fused? otherwise fall to the step-wise HIP path
        reasons = []
        xopt, why = (None, "manual_optimizer_x_fn is set") if self._manual_optimizer_x_fn is not None else \
            recognise.describe_x_optimizer(self._optimizer_x_fn, self._optimizer_x_kwargs)
        if xopt is None:
            reasons.append(why)
        if self._x_lr_discount < 1.0 or self._x_lr_amplifier > 1.0:
            reasons.append("dynamic x learning rate")
        if self._update_x_at != list(range(self._T)):
            reasons.append("update_x_at is not 'all'")
        if self._update_p_at not in ([], [self._T - 1]):
            reasons.append("update_p_at is neither 'never' nor 'last'")
        if callback_after_backward is not None:
            reasons.append("callback_after_backward is set")
        noise_var, why = recognise.describe_callback(callback_after_t, callback_after_t_kwargs, self)
        if why:
            reasons.append(why)
        if noise_var is not None and xopt is not None and xopt.kind != L.XOPT_SGD:
            reasons.append("Langevin noise through a non-SGD x optimizer")
        plan["xopt"], plan["noise_var"] = xopt, noise_var
        plan["mode"] = "stepwise" if reasons else "fused"
        plan["why_stepwise"] = "; ".join(reasons)
        return plan, ""

    # ---- engine plumbing ----------------------------------------------------------------------------------
    def _engine_for(self, plan) -> Engine:
        net, B, device = plan["net"], plan["B"], plan["device"]
        key = (net.key(B, device), os.environ.get("MCPC_TUNING"))
        eng = _ENGINES.get(key)
        if eng is None:
            eng = Engine(net.sizes, net.acts, net.n_in, net.n_out, B, device=device, ecoef=net.ecoef)
            eng._bound_sig = None
            _ENGINES[key] = eng
            while len(_ENGINES) > _ENGINE_CACHE_SIZE:
                _, old = _ENGINES.popitem(last=False)
                old.close()
        else:
            _ENGINES.move_to_end(key)
        return eng

    def _announce_staging(self, plan):
        if plan["staged"] and not getattr(self, "_staging_announced", False):
            self._staging_announced = True
            warnings.warn(
                "In PCTrainer.train_on_batch, the model lives on {}: its parameters, latent states, inputs and targets are staged onto {} "
                "for every call, the MCPC HIP engine runs there, and x, the results and param.grad are written back to the {} tensors "
                "(a few MB over PCIe per call; move the model to 'cuda' to avoid the copies).  torch's CPU work inside the call runs on at most "
                "{} threads (all {}: their OpenMP spin-wait starves the HIP runtime); CPU work of the script between calls on many threads has the "
                "same effect -- consider OMP_NUM_THREADS / torch.set_num_threads. ".format(
                    plan["model_device"], plan["device"], plan["model_device"], _few_cpu_threads.CAP, torch.get_num_threads()),
                category=RuntimeWarning)

    @staticmethod
    def _on_engine(plan, t):
        """The tensor the engine is handed for a tensor of the script: itself, or its copy on the engine's device when staged."""
        if t is None:
            return None
        return t.to(plan["device"]).contiguous() if plan["staged"] else t

    def _sync_params(self, eng: Engine, net, force=False, plan=None):
        """(Re)bind + re-pack the Linear parameters when their storage or contents changed."""
        sig = tuple((lin.weight.data_ptr(), lin.weight._version,
                     None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version)) for lin in net.linears)
        # the signature lives on the ENGINE: two models of one architecture share it, and a trainer whose own weights did
        # not change must still rebind after another trainer ran on the engine
        if force or sig != eng._bound_sig:
            for lin in net.linears:
                if not lin.weight.is_contiguous():
                    lin.weight.data = lin.weight.data.contiguous()
            if plan is not None and plan["staged"]:
                # staged model: the engine is bound to device copies (kept alive by the engine), refreshed whenever the CPU parameters
                # change (an optimizer_p step bumps their version)
                eng.bind_params([self._on_engine(plan, lin.weight.data) for lin in net.linears],
                                [None if lin.bias is None else self._on_engine(plan, lin.bias.data) for lin in net.linears])
            else:
                eng.bind_params([lin.weight.data for lin in net.linears],
                                [None if lin.bias is None else lin.bias.data for lin in net.linears])
            # (contiguous() above may have replaced storage: take the signature after binding)
            eng._bound_sig = tuple((lin.weight.data_ptr(), lin.weight._version,
                                    None if lin.bias is None else (lin.bias.data_ptr(), lin.bias._version))
                                   for lin in net.linears)

    def _initial_state(self, plan, inputs, is_sample_x_at_batch_start):
        """Draw / validate x exactly as the first forward of the reference does (pc_trainer.py:717-733,
        pc_layer.py:185-233): one torch forward through the Sequential with the sampling flags set,
        which honours arbitrary ``sample_x_fn`` callables and their RNG draw order."""
        net = plan["net"]
        layers = net.pc_layers
        need = is_sample_x_at_batch_start
        if not need:
            # shapes are only known after Linear j; check against the expected [B, n_l]
            for l, layer in enumerate(layers):
                x = layer.get_x()
                if x is None or x.device != plan["model_device"] or tuple(x.shape) != (plan["B"], net.sizes[l]):
                    need = True     # the layer's own forward emits the reference's RuntimeWarning
        if need:
            if is_sample_x_at_batch_start:
                for layer in layers:
                    layer.set_is_sample_x(True)
            for layer in layers:
                layer._mcpc_sampling_only = True
            try:
                with torch.no_grad():
                    self._model(inputs)
            finally:
                for layer in layers:
                    layer._mcpc_sampling_only = False
        xs = []
        for layer in layers:
            x = layer.get_x()
            if x.dtype != torch.float32:
                raise NotImplementedError("latent state must be float32")
            if not x.is_contiguous():
                x.data = x.data.contiguous()
            xs.append(x)
        return xs

    def _grad_window(self):
        """First step whose parameter gradients survive to the end of the call (pc_trainer.py:853-862):
        ``.grad`` is zeroed at accumulate_p_at[0] and at an update step outside accumulate_p_at, and
        autograd adds every step after that.  None = never zeroed in this call (carries over)."""
        T, up, acc = self._T, self._update_p_at, self._accumulate_p_at
        start = None
        if acc:
            start = acc[0]
        for t in up:
            if t not in acc:
                start = t if start is None else max(start, t)
        return start

    def _collect_results(self, plan, res, T, is_return_results_every_t, is_return_outputs, is_return_representations,
                         is_return_xs, loss_fn):
        net = plan["net"]
        results = {"loss": [], "energy": [], "overall": []}
        self._engine_for(plan).sync_check()        # the one host sync of the call (also surfaces device-side faults)
        if self.mcpc_sharded and self.mcpc_reduce_results:
            dist.allreduce_flat(res.energies, self.mcpc_process_group)      # [T or 1, L + 2] fp64: loss, E_l, overall of the whole batch
        en = res.energies.cpu().numpy()
        rows = range(T) if is_return_results_every_t else [0]
        nl = len(net.sizes)
        for r in rows:
            if loss_fn is not None:
                results["loss"].append(float(en[r, 0]))
            results["energy"].append(float(en[r, 1:1 + nl].sum()) / self._energy_coefficient)
            results["overall"].append(float(en[r, -1]))
        n_rec = T if is_return_results_every_t else 1
        if is_return_outputs:
            src = res.rec_out if net.n_out > 0 else res.rec_x[-1]
            if src.device != plan["model_device"]:
                # sliced recording drains latent records to pinned host memory; `outputs` are live tensors on the model's device in the
                # reference (pc_trainer.py:733,770) whatever mcpc_record_chunk_bytes is: a model without a read-out gets them back
                # there; so does a staged (CPU) model
                src = src.to(plan["model_device"])
            results["outputs"] = [src[k] for k in range(n_rec)]
        if is_return_representations:
            host = res.rec_x[0].cpu()
            results["representations"] = [host[k] for k in range(n_rec)]
        if is_return_xs:
            hosts = [r_.cpu() for r_ in res.rec_x]
            results["xs"] = [[h[k] for h in hosts] for k in range(n_rec)]
        return results

    def _apply_p_step(self, plan, eng, net, n_acc):
        """Normalise (pc_trainer.py:905-913), all-reduce across shards, hand to the user's optimizer_p."""
        if self.mcpc_sharded and self.mcpc_world_batch is None:
            # the job-wide batch is not known on this rank: the local one rides in the bucket's last float through the ONE all-reduce,
            # and the division by the group's sum happens on the device afterwards (no second collective, no host sync)
            flat = eng.read_param_grads_flat(scale=dist.grad_scale(n_acc, 1), tail=1)
            flat[-1] = float(plan["B"])
            dist.allreduce_flat(flat, self.mcpc_process_group)
            flat = flat[:-1] / flat[-1]
        else:
            flat = eng.read_param_grads_flat(scale=dist.grad_scale(n_acc, self._global_batch(plan["B"])))
            if self.mcpc_sharded:
                dist.allreduce_flat(flat, self.mcpc_process_group)     # RCCL: one bucket per call
        if plan["staged"]:
            flat = flat.to(plan["model_device"])      # param.grad lives where the script's parameters (and its optimizer_p) live
        dist.assign_flat_grads(net.linears, flat)
        self._optimizer_p.step()

    def _add_param_grads(self, plan, eng, j, lin, accumulate):
        """dF/dtheta of Linear j (the engine's un-normalised sums) into ``.grad``: overwritten, or added like autograd does."""
        if lin.weight.grad is None:
            lin.weight.grad = torch.zeros_like(lin.weight)
        if lin.bias is not None and lin.bias.grad is None:
            lin.bias.grad = torch.zeros_like(lin.bias)
        if not plan["staged"]:
            eng.read_param_grads(j, lin.weight.grad, None if lin.bias is None else lin.bias.grad, 1.0, accumulate=accumulate)
            return
        dW = torch.empty(lin.weight.shape, dtype=torch.float32, device=plan["device"])
        db = None if lin.bias is None else torch.empty(lin.bias.shape, dtype=torch.float32, device=plan["device"])
        eng.read_param_grads(j, dW, db, 1.0, accumulate=False)
        for g, d in ((lin.weight.grad, dW), (None if lin.bias is None else lin.bias.grad, db)):
            if g is not None:
                g.add_(d.to(g.device)) if accumulate else g.copy_(d)

    # ---- fused path ---------------------------------------------------------------------------------------
    def _run_fused(self, plan, inputs, loss_fn, is_sample_x_at_batch_start, is_reset_optimizer_x_at_batch_start,
                   is_reset_optimizer_p_at_batch_start, is_return_results_every_t, is_return_outputs,
                   is_return_representations, is_return_xs):
        net, loss, xopt, T = plan["net"], plan["loss"], plan["xopt"], self._T
        eng = self._engine_for(plan)
        xs = self._initial_state(plan, inputs, is_sample_x_at_batch_start)
        if is_sample_x_at_batch_start or is_reset_optimizer_x_at_batch_start or self._optimizer_x is None:
            self.recreate_optimize_x()          # kept for API compatibility (get_optimizer_x); never stepped here
        if is_reset_optimizer_p_at_batch_start:
            self.recreate_optimize_p()
        self._announce_staging(plan)
        self._sync_params(eng, net, plan=plan)
        eng.bind_inputs(None if not bool(inputs.any()) else self._on_engine(plan, inputs.contiguous()))
        if loss.target is not None:
            tgt = loss.target
            if tgt.dtype != torch.float32 or not tgt.is_contiguous() or tgt.device != plan["device"]:
                tgt = tgt.to(device=plan["device"], dtype=torch.float32).contiguous()
            eng.bind_target(tgt)
        xs_eng = [self._on_engine(plan, x.data) for x in xs]          # the script's own tensors, or their device copies when staged
        eng.load_state(xs_eng)

        do_update = self._T - 1 in self._update_p_at
        start = self._grad_window()
        want_grads = do_update or self.mcpc_materialize_unused_grads
        acc_begin, acc_end, acc_reset = 0, 0, True
        if want_grads:
            acc_begin, acc_end = (0 if start is None else start), T
            # the engine's sums always restart; a carry from earlier calls (start is None) lives in
            # param.grad, as in the reference, and is added when the sums are read out below
        n_rec = T if is_return_results_every_t else 1
        rec_begin = 0 if is_return_results_every_t else T - 1
        rec_layers = [False] * len(net.sizes)
        if is_return_xs:
            rec_layers = [True] * len(net.sizes)
        if is_return_representations:
            rec_layers[0] = True
        if is_return_outputs and net.n_out == 0:
            rec_layers[-1] = True
        any_rec = any(rec_layers) or (is_return_outputs and net.n_out > 0)
        run_kw = dict(
            loss_kind=loss.kind, loss_var=loss.var, mask_start=loss.mask_start,
            xopt=xopt.kind, lr=xopt.lr, betas=xopt.betas, eps=xopt.eps,
            noise_mode=L.NOISE_PHILOX if plan["noise_var"] is not None else L.NOISE_NONE,
            noise_var=0.0 if plan["noise_var"] is None else plan["noise_var"],
            seed=self.mcpc_seed, step_base=_take_philox_steps(T), chain_base=self.mcpc_chain_base,
            acc_begin=acc_begin, acc_end=acc_end,
            energy_mode=L.ENERGY_ALL if is_return_results_every_t else L.ENERGY_LAST)
        host_step_bytes = 4 * plan["B"] * sum(n for n, on in zip(net.sizes, rec_layers) if on)
        self.last_record_slices = 0
        if any(rec_layers) and n_rec == T and T * host_step_bytes > self.mcpc_record_chunk_bytes:
            res = self._run_fused_sliced(eng, net, plan, T, run_kw, acc_reset, rec_layers, host_step_bytes,
                                         is_return_outputs and net.n_out > 0)
        else:
            res = eng.run(T, acc_reset=acc_reset, rec_begin=rec_begin, rec_stride=1, rec_count=n_rec if any_rec else 0,
                          rec_x=rec_layers, rec_out=is_return_outputs and net.n_out > 0, **run_kw)
        eng.store_state(xs_eng)
        if plan["staged"]:
            for x, d in zip(xs, xs_eng):
                x.data.copy_(d)                 # x back into the CPU nn.Parameters the script holds (PCLayer.get_x())
        if xopt.kind == L.XOPT_ADAM and isinstance(self._optimizer_x, optim.Adam):
            # the reference's optimizer_x object outlives the call (pc_trainer.py:742-752 recreates it only behind a flag):
            # leave it in the state T fused steps produce, so that a later call that keeps it continues correctly
            ms = [torch.empty_like(d) for d in xs_eng]
            vs = [torch.empty_like(d) for d in xs_eng]
            eng.store_adam_state(ms, vs)
            for x, m_, v_ in zip(xs, ms, vs):
                self._optimizer_x.state[x] = {"step": torch.tensor(float(T)), "exp_avg": m_.to(x.device), "exp_avg_sq": v_.to(x.device)}
        if do_update:
            self._apply_p_step(plan, eng, net, len(self._accumulate_p_at))
            self._sync_params(eng, net, force=True, plan=plan)
        elif self.mcpc_materialize_unused_grads:
            for j, lin in enumerate(net.linears):
                self._add_param_grads(plan, eng, j, lin, accumulate=start is None)
        return self._collect_results(plan, res, T, is_return_results_every_t, is_return_outputs,
                                     is_return_representations, is_return_xs, loss_fn)

    def _run_fused_sliced(self, eng, net, plan, T, run_kw, acc_reset, rec_layers, host_step_bytes, rec_out):
        """A call whose every-step trajectory would not fit the record budget on the device: the same T steps as slices of
        one `mcpc_run` each (slicing does not change a bit of the trajectories, tests/test_gpu_fullsize.py), the latent
        records of a slice go to one half of a two-buffer device ring and are copied to pinned host memory on a side stream
        while the next slice computes.  Outputs stay on the device, as in the reference (live tensors, pc_trainer.py:733,770)."""
        from ..engine import RunResult
        dev, B = plan["device"], plan["B"]
        S = max(1, min(T, self.mcpc_record_chunk_bytes // max(2 * host_step_bytes, 1)))
        host = [torch.empty(T, B, n, dtype=torch.float32, pin_memory=True) if on else None for n, on in zip(net.sizes, rec_layers)]
        ring = [[torch.empty(S, B, n, dtype=torch.float32, device=dev) if on else None for n, on in zip(net.sizes, rec_layers)]
                for _ in range(2)]
        out_full = torch.empty(T, B, net.n_out, dtype=torch.float32, device=dev) if rec_out else None
        energies = torch.zeros(T, L.ENERGY_COLS, dtype=torch.float64, device=dev)
        main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
        drained = [None, None]                              # event: the copy out of this half of the ring has finished
        acc_b = run_kw["acc_begin"]
        run_kw = dict(run_kw, energy_mode=L.ENERGY_ALL)
        n_slices = 0
        for t0 in range(0, T, S):
            n = min(S, T - t0)
            half = n_slices & 1
            if drained[half] is not None:
                main.wait_event(drained[half])
            eng.run(T, t_begin=t0, n_steps=n, adam_step0=t0, energies_out=energies,
                    acc_reset=acc_reset and (t0 <= acc_b < t0 + n),
                    rec_begin=t0, rec_stride=1, rec_count=n, rec_x=rec_layers, rec_x_bufs=ring[half],
                    rec_out=rec_out, rec_out_buf=None if out_full is None else out_full[t0:t0 + n], **run_kw)
            filled = torch.cuda.Event()
            filled.record(main)
            side.wait_event(filled)
            with torch.cuda.stream(side):
                for h, d in zip(host, ring[half]):
                    if h is not None:
                        h[t0:t0 + n].copy_(d[:n], non_blocking=True)
                drained[half] = torch.cuda.Event()
                drained[half].record(side)
            n_slices += 1
        side.synchronize()
        self.last_record_slices = n_slices
        return RunResult(energies=energies, rec_x=host, rec_out=out_full)

    # ---- step-wise path -------------------------------------------------------------------------------------
    def _run_stepwise(self, plan, inputs, loss_fn, is_sample_x_at_batch_start, is_reset_optimizer_x_at_batch_start,
                      is_reset_optimizer_p_at_batch_start, is_return_results_every_t, is_return_outputs,
                      is_return_representations, is_return_xs, callback_after_backward, callback_after_backward_kwargs,
                      callback_after_t, callback_after_t_kwargs, is_checking_after_callback_after_t):
        """Reference control flow (pc_trainer.py:712-981) with the HIP kernel as the gradient engine."""
        net, loss, T = plan["net"], plan["loss"], self._T
        eng = self._engine_for(plan)
        xs = self._initial_state(plan, inputs, is_sample_x_at_batch_start)
        if is_sample_x_at_batch_start or is_reset_optimizer_x_at_batch_start or self._optimizer_x is None:
            self.recreate_optimize_x()
        if is_reset_optimizer_p_at_batch_start:
            self.recreate_optimize_p()
        self._announce_staging(plan)
        eng.bind_inputs(None if not bool(inputs.any()) else self._on_engine(plan, inputs.contiguous()))
        if loss.target is not None:
            eng.bind_target(loss.target.to(device=plan["device"], dtype=torch.float32).contiguous())
        results = {"loss": [], "energy": [], "overall": []}
        if is_return_outputs:
            results["outputs"] = []
        if is_return_representations:
            results["representations"] = []
        if is_return_xs:
            results["xs"] = []
        dynamic_lr = self._x_lr_discount < 1.0 or self._x_lr_amplifier > 1.0
        overalls = []
        energies = torch.zeros(T, L.ENERGY_COLS, dtype=torch.float64, device=plan["device"])
        params = [p for lin in net.linears for p in ([lin.weight] if lin.bias is None else [lin.weight, lin.bias])]
        nl = len(net.sizes)
        B_global = self._global_batch(plan["B"]) if self._update_p_at else None     # (a collective when sharded: only if used)
        for t in range(T):
            self._sync_params(eng, net, plan=plan)
            eng.load_state([self._on_engine(plan, x.data) for x in xs])
            keep = is_return_results_every_t or t == T - 1
            res = eng.run(T, t_begin=t, n_steps=1, loss_kind=loss.kind, loss_var=loss.var, mask_start=loss.mask_start,
                          xopt=L.XOPT_SGD, lr=1.0, update_x=False, acc_begin=t, acc_end=t + 1, acc_reset=True,
                          energy_mode=L.ENERGY_ALL, energies_out=energies,
                          rec_begin=t, rec_stride=1, rec_count=1 if (keep and is_return_outputs and net.n_out > 0) else 0,
                          rec_x=False, rec_out=True)
            if self.mcpc_sharded and self.mcpc_reduce_results:
                dist.allreduce_flat(energies[t], self.mcpc_process_group)         # the whole batch's loss / energies of this step
            if keep:
                if is_return_outputs:
                    results["outputs"].append(res.rec_out[0].to(plan["model_device"]) if net.n_out > 0 else xs[-1].detach().clone())
                if is_return_representations:
                    results["representations"].append(self.get_model_representations().clone().detach().cpu())
                if is_return_xs:
                    results["xs"].append(self.get_model_xs_copy())
                row = energies[t].cpu().numpy()
                if loss_fn is not None:
                    results["loss"].append(float(row[0]))
                results["energy"].append(float(row[1:1 + nl].sum()) / self._energy_coefficient)
                results["overall"].append(float(row[-1]))
            if dynamic_lr:
                overalls.append(float(energies[t, -1]))
            # zero_grad schedule (pc_trainer.py:848-859)
            if t in self._update_x_at:
                self._optimizer_x.zero_grad()
            if (t in self._update_p_at and t not in self._accumulate_p_at) or \
                    (self._accumulate_p_at and t == self._accumulate_p_at[0]):
                self._optimizer_p.zero_grad()
            # "backward": dF/dx from the kernel, dF/dtheta added into .grad like autograd does
            for x, g in zip(xs, res.xgrad):
                g = g.to(x.device)
                x.grad = g if x.grad is None else x.grad.add_(g)
            for j, lin in enumerate(net.linears):
                self._add_param_grads(plan, eng, j, lin, accumulate=True)
            if callback_after_backward is not None:
                callback_after_backward(t, **callback_after_backward_kwargs)
            if t in self._update_x_at:
                self._optimizer_x.step()
                if dynamic_lr and len(overalls) >= 2:
                    factor = self._x_lr_discount if not (overalls[-1] < overalls[-2]) else self._x_lr_amplifier
                    if factor != 1.0:
                        for group in self._optimizer_x.param_groups:
                            group["lr"] = group["lr"] * factor
            if t in self._update_p_at:
                div = len(self._accumulate_p_at) * B_global if self._accumulate_p_at else B_global
                for p in params:
                    p.grad = p.grad / div
                self._optimizer_p.step()
            if callback_after_t is not None:
                callback_after_t(t, **callback_after_t_kwargs)
                if is_checking_after_callback_after_t:
                    slow_down_warning("PCTrainer.train_on_batch", "is_checking_after_callback_after_t", "False")
                    if not (self.get_is_model_training() == True):  # noqa: E712
                        raise RuntimeError(
                            "If you do <model.eval()> in <callback_after_t()>, you need to put model back to train mode "
                            "when leaving <callback_after_t()>. ")
        eng.sync_check()           # surfaces a device-side fault word of any of the T single-step launches
        return results

    # ---- schedules (reference :1068-1108) ---------------------------------------------------------------------
    def _preprocess_step_index_list(self, indices: typing.Union[str, typing.List[int]], T: int) -> typing.List[int]:
        assert isinstance(indices, (str, list))
        assert isinstance(T, int) and T > 0
        if isinstance(indices, str):
            table = {"all": lambda: list(range(T)), "last": lambda: [T - 1],
                     "last_half": lambda: list(range(T // 2, T)), "never": lambda: []}
            if indices not in table:
                raise NotImplementedError
            return table[indices]()
        for t in indices:
            assert isinstance(t, (int, np.integer))
            assert 0 <= t < T
        return indices
