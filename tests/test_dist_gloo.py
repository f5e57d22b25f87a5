"""N > 1 path on CPU: two gloo ranks shard the chains, all-reduce the flat Hebbian bucket once per call
(montecarlopredictivecoding_amd/dist.py) and must reproduce the unsharded result.

No GPU here, so each rank's *compute* is the NumPy oracle; what is under test is the distributed
design itself: shard bounds, global chain ids feeding the Philox noise (trajectories independent of
the shard count), sum-all-reduce before the 1/(n_acc*B_global) normalisation, flat bucket layout,
and that a torch optimizer stepping on the reduced grads leaves every rank with identical weights.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import mcpc_oracle as mo
from oracle import philox
from oracle.cases import make_case_inputs

CASE = dict(sizes=[6, 16, 16], acts=["relu", "tanh", "relu"], ecoef=[1.0, 1.0, 1.0], n_in=6, n_out=24,
            loss="bernoulli", var=1.0, perc=0.5, B=22, seed=909, x0_range=2.0, calls=[dict(T=9)])
T, LR, MIX = 9, 0.03, 3
ACT = {"identity": 0, "relu": 1, "tanh": 2}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_shard(W, b, X0, inputs, target, begin, end):
    net = mo.NetSpec(sizes=CASE["sizes"], acts=[ACT[a] for a in CASE["acts"]], W=W, b=b)
    noise = lambda t, l: philox.layer_normals(CASE["seed"], t, l, begin, end - begin, CASE["sizes"][l])   # noqa: E731
    return mo.run(net, inputs[begin:end], [x[begin:end] for x in X0], mo.LossSpec(mo.LOSS_BERNOULLI, target[begin:end]),
                  mo.XOpt(mo.OPT_SGD, LR), T, noise=noise, accumulate_p_at=list(range(MIX, T)))


def _linears(W, b):
    lins = []
    for w, bb in zip(W, b):
        lin = nn.Linear(w.shape[1], w.shape[0])
        with torch.no_grad():
            lin.weight.copy_(torch.from_numpy(w)); lin.bias.copy_(torch.from_numpy(bb))
        lins.append(lin)
    return lins


def _worker(rank, world, port, out_dir):
    import torch.distributed as tdist
    from montecarlopredictivecoding_amd import dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    W, b, X0, inputs, target = make_case_inputs(CASE)
    begin, end = dist.shard_bounds(CASE["B"], rank, world)
    res = _run_shard(W, b, X0, inputs, target, begin, end)
    lins = _linears(W, b)
    # local un-normalised sums -> flat bucket (layout of mcpc_read_param_grads_flat) scaled by 1/(n_acc*B_global)
    scale = dist.grad_scale(T - MIX, CASE["B"])
    flat = torch.cat([torch.from_numpy(np.concatenate([gw.reshape(-1), gb.reshape(-1)])).float()
                      for gw, gb in zip(res.gW, res.gb)]) * scale
    assert flat.numel() == dist.flat_param_count(lins)
    dist.allreduce_flat(flat)
    dist.assign_flat_grads(lins, flat)
    opt = torch.optim.Adam([p for lin in lins for p in lin.parameters()], lr=0.01)
    opt.step()
    energy = torch.from_numpy(res.overall.copy())
    tdist.all_reduce(energy)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=flat.numpy(), energy=energy.numpy(),
             W0=lins[0].weight.detach().numpy(), W3=lins[3].weight.detach().numpy(),
             x_last=res.xs[2], begin=begin, end=end)
    tdist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_reproduces_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    W, b, X0, inputs, target = make_case_inputs(CASE)
    full = _run_shard(W, b, X0, inputs, target, 0, CASE["B"])
    scale = 1.0 / ((T - MIX) * CASE["B"])
    flat_ref = np.concatenate([np.concatenate([gw.reshape(-1), gb.reshape(-1)]) for gw, gb in zip(full.gW, full.gb)]) * scale
    lins = _linears(W, b)
    from montecarlopredictivecoding_amd import dist
    dist.assign_flat_grads(lins, torch.from_numpy(flat_ref).float())
    torch.optim.Adam([p for lin in lins for p in lin.parameters()], lr=0.01).step()
    r = [np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world)]
    assert (int(r[0]["begin"]), int(r[0]["end"]), int(r[1]["begin"]), int(r[1]["end"])) == (0, 11, 11, 22)
    for k in range(world):
        np.testing.assert_allclose(r[k]["flat"], flat_ref, rtol=2e-5, atol=1e-6)       # only the summation order differs
        np.testing.assert_allclose(r[k]["energy"], full.overall, rtol=1e-6)
        np.testing.assert_allclose(r[k]["W3"], lins[3].weight.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert np.array_equal(r[0]["flat"], r[1]["flat"]) and np.array_equal(r[0]["W0"], r[1]["W0"])
    # trajectories are independent of the sharding: bitwise equal to the unsharded run
    x_cat = np.concatenate([r[0]["x_last"], r[1]["x_last"]])
    assert np.array_equal(x_cat, full.xs[2])


def test_shard_bounds_cover_everything():
    from montecarlopredictivecoding_amd import dist
    for total, world in ((48000, 8), (6000, 7), (5, 8)):
        spans = [dist.shard_bounds(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    with pytest.raises(ValueError):
        dist.shard_bounds(10, 3, 3)


def _ragged_worker(rank, world, port, out_dir):
    import torch.distributed as tdist
    import montecarlopredictivecoding_amd.predictive_coding as pc
    from montecarlopredictivecoding_amd import dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    model = nn.Sequential(nn.Linear(3, 3), pc.PCLayer(), nn.Linear(3, 2))
    tr = pc.PCTrainer(model, T=4, update_p_at="last", plot_progress_at=[])
    tr.set_shard(process_group=tdist.group.WORLD, chain_base=64 * rank, world_batch=None)
    # batches of a data loader whose last one is ragged on one rank only: (64, 64), (64, 36), (64, 64), (64, 36)
    seen = []
    for local in ((64, 64), (64, 36), (64, 64), (64, 36)):
        seen.append(tr._global_batch(local[rank]))
        flat = torch.full((5,), float(rank + 1))
        dist.allreduce_flat(flat)                      # the gradient bucket's collective must pair with the other rank's bucket
        assert flat.tolist() == [3.0] * 5
    tr.set_shard(process_group=tdist.group.WORLD, chain_base=0, world_batch=777)
    seen.append(tr._global_batch(5))                   # given: no collective
    np.save(os.path.join(out_dir, f"ragged{rank}.npy"), np.array(seen))
    tdist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_batch_is_a_collective_on_every_call_not_cached_per_local_size(tmp_path):
    """the job-wide batch of a sharded learning call used to be cached per LOCAL batch size; with a ragged
    last batch on one rank only that rank re-issued the all-reduce, the other skipped it, and the next gradient-bucket
    all-reduce paired with the wrong collective.  Now every rank issues it on every learning call."""
    mp.spawn(_ragged_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert np.load(os.path.join(tmp_path, f"ragged{rank}.npy")).tolist() == [128, 100, 128, 100, 777]


# ---- the generic loop on a set_shard() trainer ---------------------------------------------------------------
def _generic_model():
    import montecarlopredictivecoding_amd.predictive_coding as pc
    torch.manual_seed(77)
    M = (torch.rand(5, 5) > 0.3).float()             # an M mask: outside what the kernels express -> generic loop
    model = nn.Sequential(nn.Linear(4, 5), pc.PCLayer(M=M[0]), nn.Tanh(), nn.Linear(5, 3))
    model.train()
    return model


def _generic_call(model, inputs, target, group=None, chain_base=0, world_batch=None, sharded=False):
    import warnings
    import montecarlopredictivecoding_amd.predictive_coding as pc
    from montecarlopredictivecoding_amd.utils.model import fe_fn
    tr = pc.PCTrainer(model, T=6, update_x_at="all", optimizer_x_fn=torch.optim.SGD, optimizer_x_kwargs={"lr": 0.05},
                      update_p_at="last", accumulate_p_at=[2, 3, 4, 5], optimizer_p_fn=torch.optim.SGD,
                      optimizer_p_kwargs={"lr": 0.5}, plot_progress_at=[])
    tr._test_only_generic_on_cpu = True              # (no GPU in the CPU suite: the loop's arithmetic is what is under test)
    if sharded:
        tr.set_shard(process_group=group, chain_base=chain_base, world_batch=world_batch, reduce_results=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = tr.train_on_batch(inputs=inputs, loss_fn=fe_fn, loss_fn_kwargs={"_target": target, "_var": 0.7}, is_log_progress=False)
    assert tr.last_call_mode == "generic"
    return res


def _linear_params(model):
    return [p for m in model if isinstance(m, nn.Linear) for p in m.parameters()]      # (a PCLayer's x is a Parameter too)


def _generic_data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(10, 4, generator=g), torch.randn(10, 3, generator=g)


def _generic_worker(rank, world, port, out_dir):
    import torch.distributed as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    inputs, target = _generic_data()
    begin, end = (0, 6) if rank == 0 else (6, 10)    # uneven shards; the job-wide batch rides on a collective (world_batch=None)
    model = _generic_model()
    res = _generic_call(model, inputs[begin:end], target[begin:end], group=tdist.group.WORLD, chain_base=begin, sharded=True)
    np.savez(os.path.join(out_dir, f"generic{rank}.npz"), overall=np.array(res["overall"]), loss=np.array(res["loss"]),
             energy=np.array(res["energy"]), **{f"p{i}": p.detach().numpy() for i, p in enumerate(_linear_params(model))})
    tdist.destroy_process_group()


@pytest.mark.timeout(300)
def test_generic_loop_on_a_sharded_trainer_all_reduces_grads_and_results(tmp_path):
    """A set_shard() trainer whose call is outside the kernels (an M mask here) used to train diverging replicas: the generic loop
    divided by the LOCAL batch and never all-reduced.  Two uneven gloo shards must end with identical parameters, equal to the
    unsharded call's up to the summation order, and report the whole batch's loss / energy / overall (reduce_results=True)."""
    mp.spawn(_generic_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    inputs, target = _generic_data()
    model = _generic_model()
    ref = _generic_call(model, inputs, target)
    r = [np.load(os.path.join(tmp_path, f"generic{k}.npz")) for k in range(2)]
    for i, p in enumerate(_linear_params(model)):
        assert np.array_equal(r[0][f"p{i}"], r[1][f"p{i}"])
        np.testing.assert_allclose(r[0][f"p{i}"], p.detach().numpy(), rtol=2e-5, atol=1e-6)
    for key in ("loss", "energy", "overall"):
        assert np.array_equal(r[0][key], r[1][key])
        np.testing.assert_allclose(r[0][key], np.array(ref[key]), rtol=2e-6)
