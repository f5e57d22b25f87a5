#!/usr/bin/env python3
"""bench.py -- Langevin steps/s of the MCPC hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5000 --warmup 500
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (cfg-M of BASELINE.md section 4, synthetic data, random-init weights):
    net 30-256-256-784 ReLU (get_model shape, reference utils/model.py:47-69), Bernoulli read-out,
    6000 chains PER GPU (weak scaling; global chain ids keep the Philox noise shard-invariant),
    SGD-x lr 0.03 + Langevin noise var 2, one call of K steps = K/5 mixing + 4K/5 sampling with the
    Hebbian sums accumulated over the sampling steps (the reference's training=True call,
    utils/training_evaluation.py:43-56), loss + layer energies recorded every step, x recorded
    every 100 steps, followed by the normalised parameter-gradient read-out and -- for N > 1 -- the
    single all-reduce of the 276 146-float gradient bucket.  One "step" = one Langevin step of all
    6000 chains of a GPU; value = N*K / wall time.

The mixing steps of the call (no Hebbian sums) run the library's mixed 32-/16-chain schedule on all
256 CUs, the sampling steps the plain schedule with the Hebbian flush on the idle CUs (DESIGN.md
section 4).  Everything in the timed region goes through the C ABI (libmcpc.so); inputs are resident
in HBM before the clock starts.  `cpu_baseline` (rank 0, N = 1 only) times oracle/torch_port.py -- a
torch-autograd port with the reference's op mix -- on the host cores for a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZES = [30, 256, 256]
N_OUT = 784
S_MACS = 30 * 256 + 256 * 256 + 256 * 784          # 273 920 MACs per chain per GEMM sweep
PEAK_FP32_TFLOPS = 157.3                           # MI355X_MICROARCH.md: fp32 MFMA = vector peak
PEAK_HBM_GBS = 8000.0


def make_problem(batch, seed, device):
    g = torch.Generator().manual_seed(seed)
    dims = [30] + SIZES + [N_OUT]
    W, b = [], []
    for j in range(len(dims) - 1):
        k = 1.0 / dims[j] ** 0.5
        W.append(((torch.rand(dims[j + 1], dims[j], generator=g) * 2 - 1) * k).to(device))
        b.append(((torch.rand(dims[j + 1], generator=g) * 2 - 1) * k).to(device))
    y = (torch.rand(batch, N_OUT, generator=g) < 0.13).float().to(device)     # binarised-MNIST density
    xs = [((torch.rand(batch, n, generator=g) * 2 - 1) * 10.0).to(device) for n in SIZES]
    return W, b, y, xs


def cpu_baseline(batch, budget_s=20.0):
    """Bounded sample of the same workload on the host cores (torch autograd port of the reference loop)."""
    from oracle import torch_port
    W, b, y, xs = make_problem(batch, 30, torch.device("cpu"))
    model, nodes, lins = torch_port.build(SIZES, [1, 1, 1], 30, N_OUT, [w.numpy() for w in W], [x.numpy() for x in b])
    loss_fn = torch_port.make_loss("bernoulli", y.numpy())
    inputs = torch.zeros(batch, 30)
    xs0 = [x.numpy() for x in xs]
    torch_port.run(model, nodes, lins, inputs, xs0, loss_fn, 2, 0.03, noise_var=2.0, acc_begin=0)      # warm-up
    # be fair to the CPU: torch's default (one thread per host cpu) oversubscribes these small GEMMs, so the
    # thread count is calibrated on 3-step probes and the bounded sample runs at the fastest setting
    default_threads = torch.get_num_threads()
    best = (float("inf"), default_threads)
    for nthreads in sorted({8, 16, 32, 64, default_threads}):
        if nthreads > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nthreads)
        t0 = time.perf_counter()
        torch_port.run(model, nodes, lins, inputs, xs0, loss_fn, 3, 0.03, noise_var=2.0, acc_begin=0)
        best = min(best, ((time.perf_counter() - t0) / 3, nthreads))
    per, nthreads = best
    torch.set_num_threads(nthreads)
    n = int(max(10, min(400, budget_s / per)))
    t0 = time.perf_counter()
    torch_port.run(model, nodes, lins, inputs, xs0, loss_fn, n, 0.03, noise_var=2.0, acc_begin=n // 5)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} Langevin steps of the same 6000-chain cfg-M call (mixing {n // 5} + sampling {n - n // 5}, "
                      f"Hebbian grads by autograd every step as the reference does), oracle/torch_port.py, "
                      f"{torch.get_num_threads()} torch threads (fastest of 8/16/32/64/default on 3-step probes) of {os.cpu_count()} host cpus"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--batch", type=int, default=6000, help="chains per GPU")
    ap.add_argument("--mode", choices=["learning", "inference"], default="learning")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed second call (other mode); used for clean PMC passes")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine

    B, K, Wm = args.batch, args.steps, args.warmup
    W, b, y, xs = make_problem(B, 30 + rank, device)
    # parameters are replicated: every rank uses rank 0's draw
    W0, b0, _, _ = make_problem(8, 30, device)
    W, b = W0, b0
    eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=device)
    eng.bind_params(W, b)
    eng.bind_inputs(None)
    eng.bind_target(y)
    eng.load_state(xs)
    n_params = eng.param_count()

    def one_call(T, learning, profile=False):
        mixing = T // 5
        eng.set_profiling(profile)
        res = eng.run(T, loss_kind=L.LOSS_BERNOULLI, xopt=L.XOPT_SGD, lr=0.03,
                      noise_mode=L.NOISE_PHILOX, noise_var=2.0, seed=30, chain_base=rank * B,
                      acc_begin=mixing if learning else 0, acc_end=T if learning else 0,
                      energy_mode=L.ENERGY_ALL, rec_begin=0, rec_stride=100, rec_count=(T + 99) // 100, rec_x=True)
        flat = None
        if learning:
            flat = eng.read_param_grads_flat(scale=1.0 / ((T - mixing) * B * world))
            if dist is not None:
                dist.all_reduce(flat)           # RCCL over xGMI: the path's only exchange step
        return res, flat

    learning = args.mode == "learning"
    if Wm > 0:
        # warm-up: W steps in the timed call's shape, and W steps in the other mode (the untimed secondary figure below; an
        # inference-only call also runs the mixed 32-/16-chain schedule, which a short mixing phase does not reach)
        one_call(Wm, learning)
        one_call(Wm, not learning)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, flat = one_call(K, learning, profile=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.sync_check()          # raises if a kernel reported a device-side fault during the timed call
    kernel_ms, n_launch, n_ksteps = eng.last_step_kernel_ms()
    eng.set_profiling(False)
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # secondary figure: the same K steps without Hebbian accumulation (training=False call)
    other = None
    if world == 1 and not args.no_secondary:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_call(K, not learning)
        torch.cuda.synchronize()
        other = K / (time.perf_counter() - t1)

    en = res.energies[-1].tolist()
    finite = all(abs(v) < 1e30 for v in en)
    if rank == 0:
        q = eng.query()
        avg_launch_s = kernel_ms * 1e-3 / max(n_launch, 1)
        steps_per_launch = n_ksteps / max(n_launch, 1)
        flops_per_step = 4.0 * S_MACS * B    # algorithmic FLOPs: forward 2*S + back-projection 2*S per chain-step
        achieved_tf = flops_per_step * steps_per_launch / avg_launch_s / 1e12
        bytes_per_step = 7472.0 * B
        # HBM traffic of K1 per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled as the
        # gfx950 guide prescribes, + WRITE_SIZE), stored by scripts/collect_traffic.py; None if never collected
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and B == 6000 and learning:
            with open(tpath) as fh:
                tj = json.load(fh)
            traffic = tj.get("bytes_per_step", 0.0) * steps_per_launch if tj.get("bytes_per_step") else None
        out = {
            "metric": "Langevin inference steps/sec (whole node), MNIST MCPC 784-256-256-30, batch 6000",
            "value": world * K / dt,
            "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg-M: 30-256-256-784 ReLU MCPC, Bernoulli read-out, %d chains/GPU, SGD-x lr 0.03 + Langevin noise var 2, "
                            "one call of K steps (K/5 mixing + 4K/5 sampling), energies every step, x every 100 steps" % B,
                "mode": args.mode + (" (Hebbian sums over the sampling steps + normalised grad read-out"
                                     + (" + 1 RCCL all-reduce of %d floats" % n_params if world > 1 else "") + ")" if learning else ""),
                "chains_total": B * world,
                "chain_steps_per_s": world * K * B / dt,
                "final_overall_energy": en[-1], "finite": finite,
                ("inference_only_steps_per_s" if learning else "learning_steps_per_s"): other,
                "lds_bytes_per_wg": q["lds_bytes"], "chains_per_wg": q["chains_per_wg"],
                "workgroups": q["n_workgroups"], "spill_slots": q["spill_slots"],
            },
            "roofline": {
                "kernel": q["step_kernel"],
                "bound": "mfma",
                "achieved": achieved_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_FP32_TFLOPS,
                "traffic": traffic,
                "flop_per_chain_step": 4 * S_MACS, "launches": n_launch, "steps_per_launch": steps_per_launch,
                "avg_launch_ms": avg_launch_s * 1e3,
                "hbm_side": {"achieved": bytes_per_step * steps_per_launch / avg_launch_s / 1e9, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "bytes_per_chain_step": 7472},
                # HIP events bracket the launches of the plain schedule only (all Hebbian stretches); inference stretches
                # run the mixed 32-/16-chain schedule, two concurrent launches per segment that an event would serialise.
                # Their rate is given from the wall clock of the untimed inference-only call of the same K steps:
                "inference_call_wall": (None if other is None or learning is False else
                                        {"us_per_step": 1e6 / other, "achieved": flops_per_step * other / 1e12,
                                         "unit": "TFLOP/s", "frac": flops_per_step * other / 1e12 / PEAK_FP32_TFLOPS}),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, args.cpu_budget)
            out["config"]["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
