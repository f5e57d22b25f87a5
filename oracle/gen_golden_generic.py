"""g15_*: the keyword surface of the reference that no script uses ("must work, need not be fast").
TEST INFRASTRUCTURE ONLY.

    python -m oracle.gen_golden_generic          # build container only: imports /root/reference, writes tests/golden/g15_*.npz

Every scenario below is ONE function of a ``predictive_coding``-shaped module ``pc``: it builds a small network and a trainer from
seeded NumPy data, makes one or two ``train_on_batch`` calls with the keyword under test and returns plain arrays (the results
lists, final latent states, parameters after the call, gradients ...).  The generator calls it with the IMPORTED REFERENCE on the CPU
and stores what comes back; tests/test_gpu_generic.py calls the SAME function with this package's ``predictive_coding`` on the GPU and
compares.  Nothing of the reference's source is stored: the fixtures are inputs' seeds and output arrays.

Keywords covered (reference file:line): PCLayer S / M masks, is_holding_error, is_keep_energy_per_datapoint, non-quadratic energy_fn
(pc_layer.py:15-25,236-300); PCTrainer loss_x_fn, loss_inputs_fn + is_optimize_inputs, is_unwrap_inputs, energy_coefficient,
early_stop_condition + update_p_at_early_stop, backward_kwargs, is_clear_energy_after_use, is_return_batchelement_loss
(pc_trainer.py:27-49,500-524,776-845,853-914).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _net(pc, rs, sizes, n_in, n_out, device, layer_kw=None, act=True):
    """Sequential[Linear, PCLayer, (Tanh), ..., Linear] with seeded weights; x0 of layer l pinned through sample_x_fn."""
    import torch.nn as nn
    dims = [n_in] + list(sizes) + [n_out]
    built = []
    for l in range(len(sizes)):
        built.append(nn.Linear(dims[l], dims[l + 1]))
        built.append((l, dict((layer_kw or {}).get(l, {}))))          # a PCLayer, made once the batch size is known (_assemble)
        if act:
            built.append(nn.Tanh())
    built.append(nn.Linear(dims[-2], dims[-1]))
    return built, dims


def _assemble(pc, rs, built, dims, B, device):
    import torch
    import torch.nn as nn
    mods = []
    for m in built:
        if isinstance(m, tuple):
            l, kw = m
            x0 = torch.from_numpy(rs.uniform(-1.0, 1.0, size=(B, dims[l + 1])).astype(np.float32)).to(device)
            kw = dict(kw)
            for key in ("S", "M"):
                if key in kw:
                    kw[key] = torch.from_numpy(kw[key]).to(device)
            mods.append(pc.PCLayer(sample_x_fn=(lambda inp, x0=x0: x0.clone()), **kw))
        else:
            mods.append(m)
    model = nn.Sequential(*mods)
    with torch.no_grad():
        for m in model:
            if isinstance(m, nn.Linear):
                m.weight.copy_(torch.from_numpy(rs.uniform(-0.6, 0.6, size=tuple(m.weight.shape)).astype(np.float32)))
                m.bias.copy_(torch.from_numpy(rs.uniform(-0.3, 0.3, size=tuple(m.bias.shape)).astype(np.float32)))
    model.train()
    return model.to(device)


def _sq_loss(o, _target, _reduction="sum"):
    d = 0.5 * (o - _target) ** 2
    return d.sum() if _reduction == "sum" else d


def _pack(model, trainer, res, extra=None):
    import torch
    import torch.nn as nn
    out = {}
    for key in ("loss", "energy", "overall"):
        out[key] = np.asarray(res[key], dtype=np.float64)
    for i, x in enumerate(trainer.get_model_xs()):
        out[f"x{i}"] = x.detach().cpu().numpy()
    j = 0
    for m in model.modules():
        if isinstance(m, nn.Linear):
            out[f"W{j}"] = m.weight.detach().cpu().numpy()
            out[f"b{j}"] = m.bias.detach().cpu().numpy()
            if m.weight.grad is not None:
                out[f"gW{j}"] = m.weight.grad.detach().cpu().numpy()
                out[f"gb{j}"] = m.bias.grad.detach().cpu().numpy()
            j += 1
    if "outputs" in res:
        out["outputs_last"] = res["outputs"][-1].detach().cpu().numpy()
    if "overall_elementwise" in res:
        out["overall_elementwise"] = res["overall_elementwise"].detach().cpu().numpy()
    out.update(extra or {})
    return out


def _standard(pc, device, seed, *, layer_kw=None, act=True, sizes=(5, 7, 6), n_in=5, n_out=4, B=6, T=7, trainer_kw=None, call_kw=None,
              loss=True, p_lr=0.05, after=None):
    import torch
    import torch.optim as optim
    rs = np.random.RandomState(seed)
    built, dims = _net(pc, rs, sizes, n_in, n_out, device, layer_kw, act)
    model = _assemble(pc, rs, built, dims, B, device)
    y = torch.from_numpy(rs.uniform(-1, 1, size=(B, n_out)).astype(np.float32)).to(device)
    tkw = dict(T=T, optimizer_x_fn=optim.SGD, optimizer_x_kwargs={"lr": 0.1}, update_p_at="last", accumulate_p_at="never",
               optimizer_p_fn=optim.SGD, optimizer_p_kwargs={"lr": p_lr}, plot_progress_at=[])
    tkw.update(trainer_kw or {})
    trainer = pc.PCTrainer(model, **tkw)
    ckw = dict(inputs=torch.zeros(B, n_in, device=device), is_log_progress=False, is_return_results_every_t=True,
               is_checking_after_callback_after_t=False, is_return_outputs=True)
    if loss:
        ckw.update(loss_fn=_sq_loss, loss_fn_kwargs={"_target": y})
    ckw.update(call_kw or {})
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = trainer.train_on_batch(**ckw)
    extra = after(model, trainer, res) if after else None
    return _pack(model, trainer, res, extra), trainer


# ---- scenarios ---------------------------------------------------------------------------------------------------------------------
def s_m_mask(pc, device):
    M = np.array([1.0, 0.0, 2.0, 1.0, 0.5, 0.0, 1.0], dtype=np.float32)
    return _standard(pc, device, 101, layer_kw={1: dict(M=M)})


def s_s_mask(pc, device):
    # S works on linear chains: energy[b, i, j] = S[i, j] * 0.5 (mu_i - x_j)^2
    rs = np.random.RandomState(7)
    S = (rs.rand(4, 4) < 0.6).astype(np.float32) + np.eye(4, dtype=np.float32)
    return _standard(pc, device, 102, layer_kw={0: dict(S=S)}, act=False, sizes=(4,), n_in=3, n_out=2)


def s_loss_x(pc, device):
    return _standard(pc, device, 103, trainer_kw=dict(loss_x_fn=lambda x: 0.05 * (x ** 2).sum()))      # (the reference sums the per-layer values: equal shapes or scalars, pc_trainer.py:800-805)


def s_optimize_inputs(pc, device):
    import torch

    def after(model, trainer, res):
        return {"inputs_final": trainer.inputs.detach().cpu().numpy()}

    rs = np.random.RandomState(5)
    inp = torch.from_numpy(rs.uniform(-1, 1, size=(6, 5)).astype(np.float32)).to(device)
    return _standard(pc, device, 104, trainer_kw=dict(loss_inputs_fn=lambda i: 0.1 * (i ** 2).sum()),
                     call_kw=dict(inputs=inp, is_optimize_inputs=True), after=after)


def s_energy_coefficient(pc, device):
    # a kernel path in this package (every layer's c_l scaled): the only g15 scenario that must NOT take the generic loop
    return _standard(pc, device, 105, trainer_kw=dict(energy_coefficient=0.5))


def s_early_stop(pc, device):
    # a step that meets the condition takes a parameter step (update_p_at_early_stop) and ends the call (pc_trainer.py:845-859,904-914,979-981)
    return _standard(pc, device, 106, trainer_kw=dict(update_p_at="never", early_stop_condition="t > 1 and overall.item() < 9.0",
                                                       update_p_at_early_stop=True))


def s_batchelement(pc, device):
    kw = {l: dict(is_keep_energy_per_datapoint=True) for l in range(3)}

    def after(model, trainer, res):
        return {f"epd{i}": layer.energy_per_datapoint().detach().cpu().numpy() for i, layer in enumerate(trainer.get_model_pc_layers())}

    return _standard(pc, device, 107, layer_kw=kw, call_kw=dict(is_return_batchelement_loss=True), after=after)


def s_quartic_energy(pc, device):
    return _standard(pc, device, 108, layer_kw={1: dict(energy_fn=lambda i: 0.25 * (i["mu"] - i["x"]) ** 4)})


def s_holding_error_clear_energy_backward_kwargs(pc, device):
    def after(model, trainer, res):
        layers = list(trainer.get_model_pc_layers())
        return {"error1": layers[1].error.detach().cpu().numpy(), "energy_cleared": np.asarray([float(layers[0].energy() is None)])}

    return _standard(pc, device, 109, layer_kw={1: dict(is_holding_error=True)},
                     call_kw=dict(is_clear_energy_after_use=True, backward_kwargs={"retain_graph": True}), after=after)


def s_unwrap_inputs(pc, device):
    """A model that is not the Sequential chain and takes two inputs (is_unwrap_inputs, pc_trainer.py:702-740)."""
    import torch
    import torch.nn as nn
    import torch.optim as optim
    rs = np.random.RandomState(110)
    B = 5
    x0 = torch.from_numpy(rs.uniform(-1, 1, size=(B, 6)).astype(np.float32)).to(device)

    class TwoIn(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Linear(3, 6), nn.Linear(2, 6)
            self.pcl = pc.PCLayer(sample_x_fn=lambda inp: x0.clone())
            self.out = nn.Linear(6, 4)

        def forward(self, u, v):
            return self.out(torch.tanh(self.pcl(self.a(u) + self.b(v))))

    model = TwoIn()
    with torch.no_grad():
        for m in (model.a, model.b, model.out):
            m.weight.copy_(torch.from_numpy(rs.uniform(-0.6, 0.6, size=tuple(m.weight.shape)).astype(np.float32)))
            m.bias.copy_(torch.from_numpy(rs.uniform(-0.3, 0.3, size=tuple(m.bias.shape)).astype(np.float32)))
    model.train()
    model.to(device)
    u = torch.from_numpy(rs.uniform(-1, 1, size=(B, 3)).astype(np.float32)).to(device)
    v = torch.from_numpy(rs.uniform(-1, 1, size=(B, 2)).astype(np.float32)).to(device)
    y = torch.from_numpy(rs.uniform(-1, 1, size=(B, 4)).astype(np.float32)).to(device)
    trainer = pc.PCTrainer(model, T=6, optimizer_x_fn=optim.SGD, optimizer_x_kwargs={"lr": 0.1}, update_p_at="last",
                           optimizer_p_fn=optim.SGD, optimizer_p_kwargs={"lr": 0.05}, plot_progress_at=[])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = trainer.train_on_batch(inputs=(u, v), is_unwrap_inputs=True, loss_fn=_sq_loss, loss_fn_kwargs={"_target": y},
                                     is_log_progress=False, is_checking_after_callback_after_t=False)
    return _pack(model, trainer, res), trainer


SCENARIOS = {
    "m_mask": s_m_mask, "s_mask": s_s_mask, "loss_x": s_loss_x, "optimize_inputs": s_optimize_inputs,
    "energy_coefficient": s_energy_coefficient, "early_stop": s_early_stop, "batchelement": s_batchelement,
    "quartic_energy": s_quartic_energy, "holding_error_clear_energy_backward_kwargs": s_holding_error_clear_energy_backward_kwargs,
    "unwrap_inputs": s_unwrap_inputs,
}


def main():
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle.gen_golden import import_reference
    pc, _ = import_reference()
    for name, fn in SCENARIOS.items():
        out, _ = fn(pc, "cpu")
        path = os.path.join(GOLDEN, f"g15_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{path}: {sorted(out)}  overall[0]={out['overall'][0]:.6f} overall[-1]={out['overall'][-1]:.6f}")


if __name__ == "__main__":
    main()
