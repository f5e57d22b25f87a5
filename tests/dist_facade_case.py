"""One sharded learning call through the facade on real engines -- shared by tests/test_gpu_dist2.py (the unsharded run, in the
pytest process) and by the two rank processes it starts (`python tests/dist_facade_case.py RANK WORLD PORT OUT_DIR`).

cfg-M's widths (30-256-256-784 ReLU, Bernoulli read-out), 6000 chains in two UNEVEN shards (3200 + 2800: the job-wide batch the
reference divides by, `len(inputs)` at pc_trainer.py:905, has to come from the group, not from the local shard), gloo process
group (two ranks cannot share one GPU under RCCL), both ranks on cuda:0.  Two consecutive learning calls, the second continuing
from the first one's state with the weights the first one's optimizer_p.step() left.
"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOTAL = 6000
BOUNDS = {1: [(0, 6000)], 2: [(0, 3200), (3200, 6000)]}
CFG = dict(input_size=30, hidden_size=256, hidden2_size=256, output_size=784, activation_fn="relu", mixing=40, sampling=88,
           optimizer_x_kwargs_mcpc={"lr": 0.03}, optimizer_p_fn_mcpc=torch.optim.SGD, optimizer_p_kwargs_mcpc={"lr": 0.5})


def run_case(rank, world, process_group=None, world_batch=None, dev="cuda:0"):
    import montecarlopredictivecoding_amd.utils.model as um
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer
    lo, hi = BOUNDS[world][rank]
    n = hi - lo
    torch.manual_seed(3)                                  # identical initial weights and Philox key on every rank
    g = torch.Generator().manual_seed(77)
    x0 = [((torch.rand(TOTAL, w, generator=g) * 2 - 1) * 2.0)[lo:hi].to(dev) for w in (30, 256, 256)]
    data = (torch.rand(TOTAL, 784, generator=g) < 0.13).float()[lo:hi].to(dev)
    model = um.get_model(CFG, True, sample_x_fn=lambda inp: None)
    layers = [m for m in model if hasattr(m, "get_x")]
    for layer, x in zip(layers, x0):
        layer._sample_x_fn = lambda inp, _x=x: _x.clone()
    tr = get_mcpc_trainer(model, CFG, training=True)
    if world > 1:
        tr.set_shard(process_group=process_group, chain_base=lo, world_batch=world_batch, reduce_results=True)
    out = {"lo": lo, "hi": hi, "modes": [], "loss": [], "energy": [], "overall": []}
    for call in range(2):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = tr.train_on_batch(inputs=torch.zeros(n, 30, device=dev), loss_fn=um.bernoulli_fn,
                                    loss_fn_kwargs={"_target": data, "_var": None}, callback_after_t=um.random_step,
                                    callback_after_t_kwargs={"_pc_trainer": tr}, is_sample_x_at_batch_start=(call == 0),
                                    is_log_progress=False, is_return_results_every_t=True,
                                    is_checking_after_callback_after_t=False)
        out["modes"].append(tr.last_call_mode)
        out["loss"].append(torch.tensor(res["loss"], dtype=torch.float64))
        out["energy"].append(torch.tensor(res["energy"], dtype=torch.float64))
        out["overall"].append(torch.tensor(res["overall"], dtype=torch.float64))
        out[f"weights{call}"] = [p.detach().cpu().clone() for lin in model if isinstance(lin, torch.nn.Linear) for p in (lin.weight, lin.bias)]
        out[f"grads{call}"] = [lin.weight.grad.detach().cpu().clone() for lin in model if isinstance(lin, torch.nn.Linear)]
        out[f"xs{call}"] = [layer.get_x().detach().cpu().clone() for layer in layers]
    return out


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as tdist
    torch.cuda.set_device(0)
    tdist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        out = run_case(rank, world, process_group=tdist.group.WORLD, world_batch=None)
    finally:
        tdist.destroy_process_group()
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))


if __name__ == "__main__":
    main()
