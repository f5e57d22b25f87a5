"""The generic loop: ``train_on_batch`` for what the HIP kernels do not express.

SURVEY.md section 8(b): the reference's keyword surface that no script uses -- S / M masks, non-quadratic ``energy_fn``,
``is_holding_error``, per-datapoint energies (``is_keep_energy_per_datapoint`` / ``is_return_batchelement_loss``), ``loss_x_fn``,
``loss_inputs_fn`` + ``is_optimize_inputs``, ``is_unwrap_inputs``, ``backward_kwargs``, ``is_clear_energy_after_use``, an
``early_stop_condition`` other than "False", models that are not the Sequential chain the engine is built for -- "must work, need not
be fast".  Such a call runs HERE: this package's own restatement of the reference's step semantics
(/root/reference/predictive_coding/pc_trainer.py:712-928) on torch autograd, on whatever device the model lives on the GPU included,
and it says so every time (``RuntimeWarning`` naming the reason; ``trainer.last_call_mode == "generic"``).  Nothing the benchmark or the
parity tests of the hot path measure goes through this module: the fused and the step-wise paths (pc_trainer.py) run on libmcpc.so
and never fall back to it silently -- a call is routed here only by ``PCTrainer._plan`` returning a reason.

Parity: tests/golden/g15_*.npz (oracle/gen_golden_generic.py drives the imported reference and this module with the same script).

One step t of the reference, as restated below (names of the quantities follow the reference so that ``early_stop_condition``
strings written against it -- they are ``eval``-ed in its local scope, pc_trainer.py:845 -- find what they mention):
    outputs = model(inputs)                      x is (re)drawn at t == 0 when flagged; optimizers are re-created behind it
    loss, energy = sum of layer energies, loss_x = sum_l loss_x_fn(x_l), loss_inputs = loss_inputs_fn(inputs)
    overall = loss + energy * energy_coefficient + loss_x + loss_inputs
    zero_grad of optimizer_x (t in update_x_at) and of optimizer_p (update step outside accumulate_p_at, or accumulate_p_at[0])
    overall.backward(**backward_kwargs); callback_after_backward
    optimizer_x.step() (+ dynamic x learning rate); parameter step with grad / (len(accumulate_p_at) * batch) or / batch
    callback_after_t; the loop ends behind a step whose early_stop_condition held
"""
import copy
import warnings

import torch


def _forward(model, inputs, unwrap):
    if unwrap == "**":
        return model(**inputs)
    if unwrap == "*":
        return model(*inputs)
    return model(inputs)


def run_generic(tr, reason, *, inputs, loss_fn, loss_fn_kwargs, is_sample_x_at_batch_start, is_reset_optimizer_x_at_batch_start,
                is_reset_optimizer_p_at_batch_start, is_unwrap_inputs, is_optimize_inputs, callback_after_backward,
                callback_after_backward_kwargs, callback_after_t, callback_after_t_kwargs, is_return_results_every_t,
                is_checking_after_callback_after_t, backward_kwargs, is_clear_energy_after_use, is_return_outputs,
                is_return_representations, is_return_xs, is_return_batchelement_loss):
    warnings.warn(
        "In PCTrainer.train_on_batch, this call is outside what the MCPC HIP engine expresses and runs on the package's generic torch "
        "loop (reference semantics on torch autograd, one Python iteration per step: orders of magnitude slower than the fused "
        "kernels), this will slow down training. Reason: {}. ".format(reason), category=RuntimeWarning)
    T = tr._T
    sharded = bool(getattr(tr, "mcpc_sharded", False))
    if sharded and tr._early_stop_condition.strip() != "False":
        # a condition evaluated on one shard's values could end the call on one rank and not on another: the ranks' collectives
        # (one per parameter step) would no longer match
        raise NotImplementedError("an early_stop_condition on a set_shard() trainer is not supported: the ranks could leave the loop at "
                                  "different steps")
    layers = list(tr.get_model_pc_layers())
    has_layers = len(layers) > 0
    unwrap = ""
    if is_unwrap_inputs:
        unwrap = "**" if isinstance(inputs, dict) else "*"
    results = {"loss": [], "energy": [], "overall": []}
    for key, on in (("outputs", is_return_outputs), ("representations", is_return_representations), ("xs", is_return_xs)):
        if on:
            results[key] = []
    dynamic_lr = tr._x_lr_discount < 1.0 or tr._x_lr_amplifier > 1.0
    history = []                    # overall of every step, for the dynamic x learning rate
    model_xs = []
    n_batch = len(inputs)
    for t in range(T):
        first = t == 0
        if first and has_layers:
            if is_sample_x_at_batch_start:
                for layer in layers:
                    layer.set_is_sample_x(True)
            if is_optimize_inputs:
                tr.inputs = torch.nn.Parameter(tr.inputs, True)
        outputs = _forward(tr._model, tr.inputs, unwrap).clone()
        if first and has_layers:
            if is_sample_x_at_batch_start or is_reset_optimizer_x_at_batch_start or tr._optimizer_x is None:
                tr.recreate_optimize_x()
            if is_optimize_inputs:
                assert len(tr._optimizer_x.param_groups) == 1
                tr._optimizer_x.param_groups[0]["params"].append(tr.inputs)
            model_xs = list(tr.get_model_xs())
            if is_reset_optimizer_p_at_batch_start:
                tr.recreate_optimize_p()
        keep = is_return_results_every_t or t == T - 1
        if keep:
            if is_return_outputs:
                results["outputs"].append(outputs)
            if is_return_representations:
                results["representations"].append(tr.get_model_representations().clone().detach().cpu())
            if is_return_xs:
                results["xs"].append(tr.get_model_xs_copy())
        # ---- the terms of the objective
        loss = loss_fn(outputs, **loss_fn_kwargs) if loss_fn is not None else None
        if loss is not None and keep:
            results["loss"].append(loss.item())
        energy = None
        if has_layers:
            energy = sum(tr.get_energies(is_per_datapoint=False))
            if is_clear_energy_after_use:
                for layer in layers:
                    layer.clear_energy()
            if keep:
                results["energy"].append(energy.item())
        loss_x = None
        if tr._loss_x_fn is not None:
            per_layer = [tr._loss_x_fn(x) for x in model_xs]
            if per_layer:
                loss_x = sum(per_layer).sum()
        loss_inputs = tr._loss_inputs_fn(tr.inputs) if (tr._loss_inputs_fn is not None and is_optimize_inputs) else None
        terms = [term for term in (loss, None if energy is None else energy * tr._energy_coefficient, loss_x, loss_inputs)
                 if term is not None]
        overall = sum(terms)
        if dynamic_lr:
            history.append(overall)
        if keep:
            results["overall"].append(overall.item())
            if is_return_batchelement_loss:
                per_chain = tr.get_energies(is_per_datapoint=True)
                kw = copy.deepcopy(loss_fn_kwargs)
                kw["_reduction"] = "none"
                results["overall_elementwise"] = sum(per_chain).squeeze() + loss_fn(outputs, **kw).sum(-1)
        # ---- the reference evaluates the condition string among its locals (pc_trainer.py:845)
        early_stop = eval(tr._early_stop_condition, {"torch": torch}, dict(
            self=tr, t=t, outputs=outputs, loss=loss, energy=energy, loss_x=loss_x, loss_inputs=loss_inputs, overall=overall,
            results=results, inputs=inputs, loss_fn_kwargs=loss_fn_kwargs, model_xs=model_xs))
        p_step = (t in tr._update_p_at) or bool(early_stop and tr._update_p_at_early_stop)
        # ---- gradients
        if has_layers and t in tr._update_x_at:
            tr._optimizer_x.zero_grad()
        if (p_step and t not in tr._accumulate_p_at) or (tr._accumulate_p_at and t == tr._accumulate_p_at[0]):
            tr._optimizer_p.zero_grad()
        overall.backward(**backward_kwargs)
        if callback_after_backward is not None:
            callback_after_backward(t, **callback_after_backward_kwargs)
        # ---- updates
        if has_layers and t in tr._update_x_at:
            tr._optimizer_x.step()
            if dynamic_lr and len(history) >= 2:
                improved = bool(history[-1] < history[-2])
                factor = tr._x_lr_amplifier if improved else tr._x_lr_discount
                if factor != 1.0:
                    for group in tr._optimizer_x.param_groups:
                        group["lr"] = group["lr"] * factor
        if p_step:
            params = list(tr.get_model_parameters())
            if sharded:
                # one shard of a larger batch (set_shard): the parameter gradients are sums over the chains, so the shards' sums are
                # all-reduced as ONE flat bucket and divided by the job-wide batch -- what the fused path does (pc_trainer._apply_p_step);
                # without this every rank would train its own replica
                from .. import dist
                flat = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in params])
                dist.allreduce_flat(flat, tr.mcpc_process_group)
                n_global = tr._global_batch(n_batch)
                off = 0
                for p in params:
                    p.grad = flat[off:off + p.numel()].view_as(p).clone()
                    off += p.numel()
            else:
                n_global = n_batch
            div = len(tr._accumulate_p_at) * n_global if tr._accumulate_p_at else n_global
            for p in params:
                p.grad = p.grad / div
            tr._optimizer_p.step()
        if callback_after_t is not None:
            callback_after_t(t, **callback_after_t_kwargs)
            if is_checking_after_callback_after_t:
                from .pc_trainer import slow_down_warning
                slow_down_warning("PCTrainer.train_on_batch", "is_checking_after_callback_after_t", "False")
                if not (tr.get_is_model_training() == True):  # noqa: E712
                    raise RuntimeError(
                        "If you do <model.eval()> in <callback_after_t()>, you need to put model back to train mode "
                        "when leaving <callback_after_t()>. ")
        if early_stop:          # the step that met the condition is completed (its parameter step included), then the call ends
            break               # (pc_trainer.py:979-981)
    if sharded and tr.mcpc_reduce_results:
        # set_shard(reduce_results=True): loss / energy / overall are sums over the chains -- those of the WHOLE batch, as in the
        # reference (pc_layer.py:295), after one all-reduce of the call's table
        from .. import dist
        keys = [k for k in ("loss", "energy", "overall") if results[k]]
        if keys:
            dev = next(tr._model.parameters()).device
            table = torch.tensor([results[k] for k in keys], dtype=torch.float64, device=dev)
            dist.allreduce_flat(table, tr.mcpc_process_group)
            for k, row in zip(keys, table.cpu().tolist()):
                results[k] = row
    return results
