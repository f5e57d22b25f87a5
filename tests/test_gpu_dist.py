"""The sharded learning path on the GPU with the real collective backend: one process, a 1-rank "nccl" (= RCCL) group.
A learning call through PCTrainer._apply_p_step (flat Hebbian bucket written by mcpc_read_param_grads_flat, all-reduced ON
DEVICE by torch.distributed, normalised by the job-wide batch, handed to the user's optimizer_p) must leave exactly the weights
of the unsharded run.  World sizes > 1 are covered on the CPU by tests/test_dist_gloo.py (gloo, world_size 2); the driver
measures N = 2, 4, 8 on a whole node."""
import os
import socket
import warnings

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _learning_call(sharded, world_batch):
    import montecarlopredictivecoding_amd.utils.model as um
    from montecarlopredictivecoding_amd.utils.training_evaluation import get_mcpc_trainer
    torch.manual_seed(3)
    cfg = dict(input_size=8, hidden_size=32, hidden2_size=32, output_size=64, activation_fn="relu", mixing=6, sampling=10,
               optimizer_x_kwargs_mcpc={"lr": 0.03}, optimizer_p_fn_mcpc=torch.optim.SGD, optimizer_p_kwargs_mcpc={"lr": 0.5})
    x0 = [torch.rand(24, n, generator=torch.Generator().manual_seed(40 + i)).to(DEV) for i, n in enumerate((8, 32, 32))]
    model = um.get_model(cfg, True, sample_x_fn=lambda inp: None)
    for layer, x in zip([m for m in model if hasattr(m, "get_x")], x0):
        layer._sample_x_fn = lambda inp, _x=x: _x.clone()
    tr = get_mcpc_trainer(model, cfg, training=True)
    if sharded:
        tr.set_shard(process_group=dist.group.WORLD, chain_base=0, world_batch=world_batch)
    data = (torch.rand(24, 64, generator=torch.Generator().manual_seed(1)) < 0.3).float().to(DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr.train_on_batch(inputs=torch.zeros(24, 8, device=DEV), loss_fn=um.bernoulli_fn,
                          loss_fn_kwargs={"_target": data, "_var": None}, callback_after_t=um.random_step,
                          callback_after_t_kwargs={"_pc_trainer": tr}, is_log_progress=False, is_checking_after_callback_after_t=False)
    assert tr.last_call_mode == "fused"
    return [p.detach().clone() for p in model.parameters() if p.dim() <= 2 and p.shape[0] != 24], \
        [m.weight.grad.clone() for m in model if isinstance(m, torch.nn.Linear)]


@pytest.mark.parametrize("world_batch", [24, None])
def test_one_rank_rccl_group_matches_unsharded_run(world_batch):
    import montecarlopredictivecoding_amd.predictive_coding.pc_trainer as pt
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base_steps = pt._PHILOX_STEPS[0]
    plain_w, plain_g = _learning_call(False, None)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        pt._PHILOX_STEPS[0] = base_steps                  # same Philox step counter as the unsharded run
        shard_w, shard_g = _learning_call(True, world_batch)   # world_batch=None: the local batch travels in the bucket
    finally:
        dist.destroy_process_group()
    for a, c in zip(plain_g, shard_g):
        assert a.is_cuda
        if world_batch is not None:
            assert torch.equal(a, c)
        else:
            # the local batch rides in the bucket's last float and the division by the group's sum follows the all-reduce:
            # (G / n_acc) / B where the unsharded run multiplies by 1 / (n_acc B) -- one rounding apart
            torch.testing.assert_close(a, c, rtol=3e-7, atol=0)
    for a, c in zip(plain_w, shard_w):
        if world_batch is not None:
            assert torch.equal(a, c)
        else:
            torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-7)          # (one SGD step of lr 0.5 on those gradients)
    assert any(float(g.abs().max()) > 0 for g in plain_g)


def test_native_rccl_allreduce_of_the_gradient_bucket_one_rank():
    """include/mcpc.h "multi-GPU": the library's own RCCL path (mcpc_comm_unique_id / mcpc_comm_init / mcpc_allreduce_grads) for
    hosts that are not on torch.distributed.  With a communicator of one rank the sum over the shards is the identity: the
    bucket read from the engine must come back bit for bit, on the caller's stream, and the error paths must be clean."""
    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    g = torch.Generator().manual_seed(5)
    sizes, n_in, n_out, B = [8, 32, 32], 8, 64, 48
    dims = [n_in] + sizes + [n_out]
    W = [((torch.rand(dims[j + 1], dims[j], generator=g) - 0.5) * 0.4).to(DEV) for j in range(4)]
    b = [((torch.rand(dims[j + 1], generator=g) - 0.5) * 0.4).to(DEV) for j in range(4)]
    eng = Engine(sizes, [L.ACT_RELU] * 3, n_in, n_out, B, device=DEV)
    eng.bind_params(W, b); eng.bind_inputs(None)
    eng.bind_target((torch.rand(B, n_out, generator=g) < 0.3).float().to(DEV))
    eng.load_state([torch.rand(B, n, generator=g).to(DEV) for n in sizes])
    eng.run(12, loss_kind=L.LOSS_BERNOULLI, lr=0.03, noise_mode=L.NOISE_PHILOX, seed=1, step_base=0, acc_begin=2, acc_end=12)
    flat = eng.read_param_grads_flat(scale=1.0 / (10 * B))
    want = flat.clone()
    with pytest.raises(L.MCPCError, match="before mcpc_comm_init"):
        eng.allreduce_grads(flat)
    uid = Engine.comm_unique_id()
    assert len(uid) == L.COMM_ID_BYTES and any(uid)
    eng.comm_init(1, 0, uid)
    with pytest.raises(L.MCPCError, match="already has a communicator"):
        eng.comm_init(1, 0, uid)
    with pytest.raises(L.MCPCError, match="bucket of"):
        eng.allreduce_grads(flat[:-1].contiguous())
    eng.allreduce_grads(flat)
    eng.sync_check()
    assert torch.equal(flat, want) and float(want.abs().max()) > 0
    eng.comm_destroy()
    eng.comm_destroy()                     # idempotent
    eng.close()


def test_bench_multi_rank_code_path_with_one_rank():
    """bench.py's N > 1 path (RCCL group, barrier + synchronize brackets, all-reduce of the gradient bucket, MAX over ranks of the
    wall time) launched the way the driver launches it (torch.distributed.run), with the one rank this box has: one JSON line on
    stdout, learning and inference-only calls timed, the collective named in the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--T", "400",
           "--force-dist", "--no-cpu-baseline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, run.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["config"]["T"] == 400 and out["config"]["finite"]
    assert out["value"] > 0 and out["value_mode"] == "learning_call"
    assert out["config"]["inference_only"]["steps_per_s"] > out["value"]          # relative: no Hebbian sums, same box, same process
    assert out["roofline"]["bound"] == "mfma" and 0.0 < out["roofline"]["frac"] < 1.0
    sc = out["self_check"]                                    # the timed call replayed on the serial / plain schedule
    assert sc["ok"] and sc["bitwise_state"] and sc["bitwise_records"] and sc["bucket_max_rel"] <= 1e-5 and sc["ranks_checked"] == 1
    assert out["build_info"]["exp"] == "0" and out["build_info"]["built_from_this_tree"] and out["build_info"]["abi"] == "4"
    # --T 400 is not the call the tracked PMC passes profiled: the line must say null and why, not print another run's bytes
    assert out["roofline"]["traffic"] is None and "traffic_is_null_because" in out["roofline"]["traffic_detail"]


def test_bench_two_ranks_rehearsed_on_gloo():
    """bench.py's N > 1 branch with TWO ranks, launched the way the driver launches it
    (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both ranks on device 0 over gloo (two ranks cannot
    share a device under RCCL): ONE JSON line, from rank 0 only; n_gpus 2; every rank's self-check counted; the reduced gradient bucket
    identical on both ranks; value = the steps BOTH ranks did / the slower rank's wall time.  With this, RCCL itself is the only piece of
    the driver's first N > 1 run that has not executed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    T, K = 300, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", str(K), "--warmup", "1", "--T", str(T),
           "--dist-backend", "gloo", "--no-cpu-baseline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, run.stdout                                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == K and out["config"]["T"] == T and out["config"]["chains_total"] == 12000
    assert out["scaling"] == "weak" and "gloo rehearsal" in out["rehearsal"]
    assert "all-reduce of 276146 floats" in out["config"]["timed_mode"]
    # value is whole-job: n_gpus * K * T / wall (two ranks time-share one device here, so it is about ONE GPU's rate, not two)
    assert out["value"] == pytest.approx(2 * K * T / (out["ms_per_step"] * 1e-3 * K), rel=1e-6)
    sc = out["self_check"]
    assert sc["ok"] and sc["ranks_checked"] == 2 and sc["bitwise_state"] and sc["bitwise_records"]
    assert sc["bucket_identical_on_all_ranks"] is True
    assert "cpu_baseline" not in out                                       # rank 0 at N = 1 only
    assert out["build_info"]["exp"] == "0" and out["build_info"]["built_from_this_tree"]
