#!/usr/bin/env python3
"""bench.py -- Langevin steps/s of the MCPC hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One bench "step" = ONE CALL of the hot path in BASELINE.json's configuration (cfg-M of BASELINE.md section 4):
    net 30-256-256-784 ReLU (get_model shape, reference utils/model.py:47-69), Bernoulli read-out,
    6000 chains PER GPU (weak scaling; global chain ids keep the Philox noise shard-invariant),
    SGD-x lr 0.03 + Langevin noise var 2, T = 5000 Langevin steps = 1000 mixing + 4000 sampling (--T overrides it),
    loss + layer energies recorded every step, x recorded every 100 steps; synthetic data, random-init weights.
--warmup W untimed calls, then --steps K timed calls (barrier + synchronize on both sides, max over ranks):

    learning call (the timed `value`): the reference's training=True call (utils/training_evaluation.py:43-56) -- Hebbian
        sums accumulated over the 4000 sampling steps, the normalised parameter-gradient read-out and, for N > 1, the
        single RCCL all-reduce of the 276 146-float gradient bucket.   value = N * K * T / wall   [Langevin steps/s]
    inference-only call (training=False, no Hebbian sums): K calls timed the same way right after, reported beside it.

Everything in the timed region goes through the C ABI (libmcpc.so); inputs are resident in HBM before the clock starts.
`roofline` is computed from HIP events the library records on its launch stream (mcpc_set_profiling) during the timed
calls; its `peak` is the ceiling of the pipe the kernel computes on (fp32 products as three fp16 MFMA products: dense fp16 peak / 3),
the fp32 MFMA peak SURVEY 8(d) names is carried beside it; `traffic` comes from the tracked PMC summary of the same command
(profiles/, separate --pmc passes: counters cannot be read inside this run) and is printed only when those passes ran THIS program --
same kernel sources (mcpc_build_info's csrc hash), batch, T, kernel and launch count -- else null with the reason.  `cpu_baseline` (rank 0, N = 1 only) times oracle/torch_port.py -- a torch-autograd port with the reference's op
mix -- on the host cores for a bounded sample.
"""
import argparse
import json
import os
import re
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
PEAK_L2_GBS = 34500.0                              # MI355X_MICROARCH.md, L2 (per XCD): ~34.5 TB/s aggregate
# packed Wf + Wb of the three GEMM Linears (padded), two fp16 planes per fp32 weight (csrc/mcpc_gemm_f16.h): 2.19 MB
FRAG_BYTES_PER_WG_STEP = 2 * 4 * (32 * 256 + 256 * 256 + 256 * 784)
# The step kernel computes every fp32 product as THREE fp16 MFMA products with fp32 accumulation (two fp16 pieces per operand, 22
# significant bits, scaled by powers of two; four products in contractions with K <= 64): the ceiling of THAT pipe for fp32-class work is
# the dense fp16 peak / 3 (MI355X_MICROARCH.md: 2516 TFLOP/s dense fp16 / bf16).  The fp32 MFMA peak SURVEY 8(d) names is carried beside it.
PEAK_F16X3_TFLOPS = 2516.0 / 3.0
# What binds a 16-chain workgroup of the step kernel with this core: the packed weights it streams out of L2 once per step through its
# CU's vector-memory return path, 64 B per clock (MI355X_MICROARCH.md) -- at the shader clock the chip holds under THIS load, which the
# library measures during the timed launches (mcpc_last_shader_clock_ghz: 1.8-2.0 GHz; the 2.4 GHz peak only as a fallback).
L1_FILL_BYTES_PER_CLK = 64
PEAK_SHADER_GHZ = 2.4
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_summary.json")


def pmc_traffic(mode, kernel_name, csrc, batch, T, launches_per_call):
    """(HBM bytes per launch of the step kernel, None) from the tracked PMC summary -- scripts/pmc_round.sh + scripts/reduce_pmc.py:
    separate --pmc passes of this command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide reads -- or (None, reason).

    Counters cannot be read inside this run, so the figure is a MEASUREMENT OF ANOTHER RUN and is only printed when that run was this
    program: the summary's `meta` (written by the passes themselves from their own bench lines) must name the kernel sources this
    library was built from (`csrc`: identical csrc/ = identical kernels, whatever the commit), this batch and T, the exact kernel, and
    the launch count of one call of this schedule.  Anything else -> `traffic: null` with the reason."""
    try:
        with open(PMC_SUMMARY) as f:
            summ = json.load(f)
    except (OSError, ValueError) as exc:
        return None, "no PMC summary (%s: %s)" % (os.path.relpath(PMC_SUMMARY, ROOT), type(exc).__name__)
    meta = summ.get("meta") or {}
    key = re.sub(r"^mcpc::", "", kernel_name.split(" (")[0])
    for what, got, want in (("csrc", meta.get("csrc"), csrc), ("batch", meta.get("batch"), batch), ("T", meta.get("T"), T)):
        if got != want:
            return None, "the PMC summary was collected for %s=%r, this run has %r (re-collect: scripts/pmc_round.sh)" % (what, got, want)
    ent = (summ.get(mode) or {}).get(key)
    if ent is None:
        return None, "the PMC summary has no entry for kernel %r in section %r" % (key, mode)
    try:
        n, d, c = ent["dispatches"], ent["derived"], ent["counters"]
        if launches_per_call and n != launches_per_call:
            return None, "the PMC passes saw %d launches of %s per call, this schedule issues %d" % (n, key, launches_per_call)
        out = {"hbm_read_bytes_per_launch": d["hbm_read_bytes"] / n, "hbm_write_bytes_per_launch": d["hbm_write_bytes"] / n,
               "launches": n, "kernel": key,
               "source": "%s, section '%s' (rocprofv3 --pmc passes of `bench.py %s`, builder-run, same kernel sources as this "
                         "library: csrc=%s; FETCH_SIZE x 2, WRITE_SIZE as read)" % (
                             os.path.relpath(PMC_SUMMARY, ROOT), mode, "--no-secondary" if mode == "learning" else "--only-inference", csrc),
               "meta": meta}
        if "TCC_REQ_sum" in c:
            out["l2_request_bytes_per_launch"] = c["TCC_REQ_sum"] * 128.0 / n       # the fragment stream out of L2: requests of 128 B
        return out, None
    except (KeyError, ZeroDivisionError) as exc:
        return None, "malformed PMC summary entry (%s)" % exc


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
    ap.add_argument("--steps", type=int, default=10, help="timed CALLS of T Langevin steps each")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up calls (of each kind)")
    ap.add_argument("--T", type=int, default=5000, help="Langevin steps per call: T/5 mixing + 4T/5 sampling (BASELINE: 5000)")
    ap.add_argument("--batch", type=int, default=6000, help="chains per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the inference-only calls (clean PMC passes of the learning call)")
    ap.add_argument("--only-inference", action="store_true", help="time inference-only calls only (clean PMC passes of the inference call)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-self-check", action="store_true", help="skip the replay of one call on the serial schedule of the barrier kernel (PMC passes)")
    ap.add_argument("--force-dist", action="store_true",
                    help="developer check: take the multi-rank code path (RCCL group, barriers, all-reduces) with whatever world size the "
                         "environment gives, 1 included -- the 1-GPU rehearsal of what the driver launches with torch.distributed.run")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL over xGMI: what the driver's N > 1 runs use).  gloo: REHEARSAL of the multi-rank branch on ONE GPU "
                         "(two ranks cannot share a device under RCCL): every rank runs on --rehearsal-device, the collectives are staged "
                         "through the host -- the line says so and is not a measurement of N GPUs")
    ap.add_argument("--rehearsal-device", type=int, default=0, help="with --dist-backend gloo: the one device every rank uses")
    args = ap.parse_args()

    # stdout carries ONE JSON line and nothing else: libraries that chat on fd 1 (RCCL prints a version banner there when a
    # communicator is created) are pointed at stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "gloo":
            local_rank = args.rehearsal_device
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    rehearsal = dist is not None and args.dist_backend == "gloo"

    def all_reduce(t, op=None):
        """The job's collective: RCCL on the device tensor; in a gloo rehearsal the tensor is staged through the host."""
        op = dist.ReduceOp.SUM if op is None else op
        if rehearsal:
            host = t.cpu()
            dist.all_reduce(host, op=op)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=op)
        return t

    from montecarlopredictivecoding_amd import _lib as L
    from montecarlopredictivecoding_amd.engine import Engine

    build, tree_csrc = L.build_info(), L.csrc_sha()
    B, K, Wm, T = args.batch, max(args.steps, 1), max(args.warmup, 0), args.T
    mixing = T // 5
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

    def one_call(learning):
        res = eng.run(T, loss_kind=L.LOSS_BERNOULLI, xopt=L.XOPT_SGD, lr=0.03,
                      noise_mode=L.NOISE_PHILOX, noise_var=2.0, seed=30, chain_base=rank * B,
                      acc_begin=mixing if learning else 0, acc_end=T if learning else 0,
                      energy_mode=L.ENERGY_ALL, rec_begin=0, rec_stride=100, rec_count=(T + 99) // 100, rec_x=True)
        flat = None
        if learning:
            flat = eng.read_param_grads_flat(scale=1.0 / ((T - mixing) * B * world))
            if dist is not None:
                all_reduce(flat)                # RCCL over xGMI: the path's only exchange step
        return res, flat

    def timed(learning, n_calls):
        """n_calls back-to-back calls between barrier + synchronize on both sides; max over ranks."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(n_calls):
            res, flat = one_call(learning)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        eng.sync_check()          # raises if a kernel reported a device-side fault during the timed calls
        plain = eng.last_step_kernel_ms() + (eng.last_shader_clock_ghz(),)
        eng.set_profiling(False)
        if dist is not None:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, res, plain

    for _ in range(Wm):
        if not args.only_inference:
            one_call(True)
        if not args.no_secondary:
            one_call(False)
    dt_learn = res = plain_l = None
    if not args.only_inference:
        dt_learn, res, plain_l = timed(True, K)
    dt_inf = plain_i = None
    if not args.no_secondary or args.only_inference:
        dt_inf, res_i, plain_i = timed(False, K)
        res = res if res is not None else res_i
    primary_learning = dt_learn is not None
    dt = dt_learn if primary_learning else dt_inf

    en = res.energies[-1].tolist()
    finite = all(abs(v) < 1e30 for v in en)

    # ---- self-check (outside the timed region): ONE call of the timed kind from the initial state, on the tuning that was
    # timed and on the serial one (the BARRIER kernel -- another program: four waves per workgroup, s_barrier hand-overs, generic
    # epilogues -- one launch per segment, one spill-ring part flushed on the caller's stream: nothing overlaps, nothing can race).  Per chain both run the same arithmetic in the same order, so the final state and the
    # records must agree BITWISE, the gradient bucket up to nothing (same 128-step Hebbian segments) and the energies up to
    # the grouping of fp32 partial sums.  What the reference defines for these: pc_trainer.py:853-862 (dF/dtheta summed over
    # accumulate_p_at), :904-914 (normalisation).
    self_check = None
    if not args.no_self_check:
        def replay(engine):
            engine.load_state(xs)
            r_ = engine.run(T, loss_kind=L.LOSS_BERNOULLI, xopt=L.XOPT_SGD, lr=0.03, noise_mode=L.NOISE_PHILOX, noise_var=2.0,
                            seed=30, step_base=0, chain_base=rank * B, acc_begin=mixing if primary_learning else 0,
                            acc_end=T if primary_learning else 0, energy_mode=L.ENERGY_ALL, rec_begin=0, rec_stride=100,
                            rec_count=(T + 99) // 100, rec_x=True)
            st = [torch.empty_like(x) for x in xs]
            engine.store_state(st)
            fl = engine.read_param_grads_flat(scale=1.0) if primary_learning else None
            engine.sync_check()
            return r_, st, fl
        ra, sa, fa = replay(eng)
        serial = "ws=0,no_overlap=1,slot_cap=128"
        eng_s = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=device, tuning=serial)
        eng_s.bind_params(W, b)
        eng_s.bind_inputs(None)
        eng_s.bind_target(y)
        eng_s_name = eng_s.query()["step_kernel"]
        rb, sb, fb = replay(eng_s)
        eng_s.close()
        ea, eb = ra.energies, rb.energies
        self_check = {
            "reference": "the same call (initial state, seed, Philox step base 0) on tuning '%s': %s" % (serial, eng_s_name),
            "mode": "learning call" if primary_learning else "inference-only call",
            "bitwise_state": all(torch.equal(p_, q_) for p_, q_ in zip(sa, sb)),
            "bitwise_records": all(torch.equal(p_, q_) for p_, q_ in zip(ra.rec_x, rb.rec_x)),
            "bucket_bitwise": None if fa is None else bool(torch.equal(fa, fb)),
            "bucket_max_rel": None if fa is None else float((fa - fb).abs().max() / fb.abs().max()),
            "energies_max_rel": float(((ea - eb).abs() / eb.abs().clamp_min(1e-300)).max()),
            "energies_finite": bool(torch.isfinite(ea).all()),
        }
        ok = (self_check["bitwise_state"] and self_check["bitwise_records"] and self_check["energies_finite"]
              and self_check["energies_max_rel"] <= 2e-6 and (fa is None or self_check["bucket_max_rel"] <= 1e-5))
        if dist is not None:
            flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
            all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
            if fa is not None:
                # the reduced bucket of the timed calls must be the SAME on every rank (one all-reduce, same summation tree)
                _, bucket = one_call(True)
                lo, hi = bucket.clone(), bucket.clone()
                all_reduce(lo, op=dist.ReduceOp.MIN)
                all_reduce(hi, op=dist.ReduceOp.MAX)
                # (reported, not part of `ok`: a collective whose ranks end with different bits would show here; the tests assert it)
                self_check["bucket_identical_on_all_ranks"] = bool(torch.equal(lo, hi))
        self_check["ok"] = ok
        self_check["ranks_checked"] = world
    if rank == 0:
        q = eng.query()
        flops_inf = 4.0 * S_MACS * B     # algorithmic FLOPs per step: forward 2*S + back-projection 2*S per chain
        flops_heb = 2.0 * S_MACS * B     # + the Hebbian sums e^T f(x) on accumulating steps
        bytes_per_step = 7472.0 * B

        def kernel_line(kernel, prof, flops_per_step, note, n_wg, wg_per_launch, mode):
            ms, n, steps, ghz = prof
            if not n or not steps:
                return None
            avg_s = ms * 1e-3 / n
            spl = steps / n
            tf = flops_per_step * spl / avg_s / 1e12
            clk = ghz if ghz and ghz > 0.5 else PEAK_SHADER_GHZ
            l1_peak = L1_FILL_BYTES_PER_CLK * clk
            l1_ach = FRAG_BYTES_PER_WG_STEP * spl * n_wg / wg_per_launch / avg_s / 1e9
            pmc, pmc_why = pmc_traffic(mode, kernel, build["csrc"], B, T, round(n / K) if K else 0)
            line = {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": PEAK_F16X3_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / PEAK_F16X3_TFLOPS,
                    "peak_is": "the ceiling of the pipe this kernel computes on: every fp32 product is three v_mfma_f32_16x16x32_f16 products "
                               "(two fp16 pieces per operand, fp32 accumulate), so dense fp16 peak 2516 / 3; the kernel issues no fp32 MFMA",
                    "fp32_mfma": {"peak": PEAK_FP32_TFLOPS, "frac": tf / PEAK_FP32_TFLOPS,
                                  "note": "the fp32 MFMA peak SURVEY 8(d) prescribes for dtype f32; not a ceiling of this kernel"},
                    # HBM bytes per launch from the PMC counters (separate passes of the same command, tracked summary)
                    "traffic": None if pmc is None else pmc["hbm_read_bytes_per_launch"] + pmc["hbm_write_bytes_per_launch"],
                    "traffic_detail": pmc if pmc is not None else {"traffic_is_null_because": pmc_why},
                    "algorithmic_flop_per_launch": flops_per_step * spl,
                    "brackets": n, "steps_per_bracket": spl,
                    "avg_bracket_ms": avg_s * 1e3, "us_per_step": avg_s / spl * 1e6,
                    "flop_per_chain_step": 4 * S_MACS,
                    "shader_clock_ghz": ghz,
                    # the same launches against the HBM roofline (north_star asks for both): algorithmic streaming bytes, SURVEY 8(d)
                    "hbm_side": {"achieved": bytes_per_step * spl / avg_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": bytes_per_step * spl / avg_s / 1e9 / PEAK_HBM_GBS, "bytes_per_chain_step": 7472},
                    # every workgroup streams the packed weights (Wf + Wb as two fp16 planes, 2.19 MB) out of its XCD's L2 once per step
                    "l2_fragment_stream": {"achieved": n_wg * FRAG_BYTES_PER_WG_STEP * spl / avg_s / 1e9, "peak": PEAK_L2_GBS,
                                           "unit": "GB/s", "frac": n_wg * FRAG_BYTES_PER_WG_STEP * spl / avg_s / 1e9 / PEAK_L2_GBS,
                                           "workgroups": n_wg, "bytes_per_workgroup_step": FRAG_BYTES_PER_WG_STEP},
                    # the same stream per CU: a workgroup of a launch does spl * n_wg / wg_per_launch steps in it
                    "l1_fill_per_cu": {"achieved": l1_ach, "peak": l1_peak, "unit": "GB/s", "frac": l1_ach / l1_peak,
                                       "peak_is": "64 B per clock x the shader clock measured during these launches (%s)"
                                                  % ("%.3f GHz" % ghz if ghz and ghz > 0.5 else "not measured: 2.4 GHz peak"),
                                       "workgroups_per_launch": wg_per_launch,
                                       "note": "the fragment stream of a workgroup on its CU's vector-memory return path: one of the serial terms "
                                               "of a 16-chain step (MFMAs 5.6 us, fragment requests 4.2, operand split ~1.5, table skeleton ~4 of "
                                               "~27; timing builds in profiles/r05_k1_bounds.txt, DESIGN section 4)"},
                    "note": note}
            return line

        # the dominant kernel of the timed call: every launch of the step kernel; algorithmic FLOPs of the STEP kernel = 4 S per
        # chain-step (the Hebbian GEMMs are a different kernel)
        km = re.search(r"round schedule: k=(\d+) .* m=(\d+)", q["step_kernel"])
        wg_per_launch = q["n_workgroups"] if km is None else round(q["n_workgroups"] * int(km.group(2)) / int(km.group(1)))
        roof = kernel_line(q["step_kernel"], plain_l if primary_learning else plain_i, flops_inf,
                           "HIP events around every launch of the step kernel during the timed "
                           + ("learning calls (mixing and Hebbian stretches alike; the Hebbian GEMMs of a segment run between and beside "
                              "the launches of the next); a launch of the round schedule advances its workgroups' share of the shard, "
                              "steps_per_bracket counts whole-shard steps" if primary_learning else "inference calls"),
                           q["n_workgroups"], wg_per_launch, "learning" if primary_learning else "inference")
        if roof is not None and primary_learning and plain_i is not None:
            inf_line = kernel_line(q["step_kernel"], plain_i, flops_inf, "the same kernel's launches during the timed inference-only calls",
                                   q["n_workgroups"], wg_per_launch, "inference")
            if inf_line is not None:
                roof["inference_only_launches"] = {k_: inf_line[k_] for k_ in ("achieved", "frac", "fp32_mfma", "traffic", "avg_bracket_ms",
                                                                               "us_per_step", "shader_clock_ghz", "l1_fill_per_cu")}
        value = world * K * T / dt
        out = {
            "metric": "Langevin inference steps/sec (whole node), MNIST MCPC 784-256-256-30, batch 6000",
            "value": value,
            "value_mode": "learning_call" if primary_learning else "inference_only_call",
            "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "cfg-M: 30-256-256-784 ReLU MCPC, Bernoulli read-out, %d chains/GPU, SGD-x lr 0.03 + Langevin noise var 2; "
                            "bench step = ONE CALL of T = %d Langevin steps (%d mixing + %d sampling), energies every step, "
                            "x every 100 steps; value = n_gpus * steps * T / wall; ms_per_step = ms per call" % (B, T, mixing, T - mixing),
                "T": T, "batch": B, "mixing": mixing, "sampling": T - mixing, "steps_are": "calls",
                # `dtype` f32: state, weights, energies and sums are fp32; every contraction multiplies its fp32 operands as two fp16 pieces
                # each (22 significant bits; rows scaled by powers of two into fp16's range) in three MFMA terms with fp32 accumulation --
                # error of the dot products against fp64 that of an fp32 MFMA chain (profiles/r05_f16x4_study.txt, DESIGN section 4: what
                # limits both is the fp32 accumulation), parity tolerances unchanged
                "arithmetic": "fp32 operands as two fp16 pieces (22 bits, power-of-two row scaling), products as 3 fp16 MFMA terms, fp32 "
                              "accumulate: on independent operands the accuracy of an fp32 MFMA chain (0.5-1.6e-7 of sum|terms|, "
                              "profiles/r05_f16x4_study.txt); WORST CASE per term 3 x 2^-22 = 7.2e-7 of |a b| when every operand rounds "
                              "the same way (one repeated value against same-sign weights), beside the fp32 accumulation both share -- "
                              "measured there 0.58 / 1.26 / 3.65e-6 of sum|terms| at K = 96 / 256 / 784 against 1.43 / 3.80 / 11.7e-6 for "
                              "torch's fp32 GEMM on the same GPU (tests/test_gpu_accuracy.py)",
                "timed_mode": ("learning call: Hebbian sums over the sampling steps + normalised grad read-out"
                               + (" + 1 RCCL all-reduce of %d floats" % n_params if world > 1 else "")) if primary_learning
                              else "inference-only call (no Hebbian sums)",
                "us_per_langevin_step": dt / (K * T) * 1e6,
                "chains_total": B * world,
                "chain_steps_per_s": value * B,
                "final_overall_energy": en[-1], "finite": finite,
                "inference_only": None if (dt_inf is None or not primary_learning) else {
                    "steps_per_s": world * K * T / dt_inf, "us_per_langevin_step": dt_inf / (K * T) * 1e6, "calls": K,
                    "achieved_tflops": flops_inf * K * T / dt_inf / 1e12,
                    "frac_of_fp32_peak": flops_inf * K * T / dt_inf / 1e12 / PEAK_FP32_TFLOPS,
                    "frac_of_f16x3_pipe": flops_inf * K * T / dt_inf / 1e12 / PEAK_F16X3_TFLOPS,
                    # the packed weights every workgroup streams out of L2 once per step (wall clock of the calls)
                    "l2_fragment_stream_gbs": q["n_workgroups"] * FRAG_BYTES_PER_WG_STEP * K * T / dt_inf / 1e9,
                    "l2_fragment_stream_frac_of_peak": q["n_workgroups"] * FRAG_BYTES_PER_WG_STEP * K * T / dt_inf / 1e9 / PEAK_L2_GBS},
                "learning_call_flops": None if not primary_learning else {
                    "achieved_tflops": (flops_inf * T + flops_heb * (T - mixing)) * K / dt / 1e12,
                    "frac_of_fp32_peak": (flops_inf * T + flops_heb * (T - mixing)) * K / dt / 1e12 / PEAK_FP32_TFLOPS,
                    "frac_of_f16x3_pipe": (flops_inf * T + flops_heb * (T - mixing)) * K / dt / 1e12 / PEAK_F16X3_TFLOPS,
                    "note": "whole call, wall clock: 4S per chain-step + 2S on the accumulating steps"},
                "lds_bytes_per_wg": q["lds_bytes"], "chains_per_wg": q["chains_per_wg"],
                "workgroups": q["n_workgroups"], "spill_slots": q["spill_slots"],
            },
            "roofline": roof,
            # what was measured: the library says what it is (include/mcpc.h: mcpc_build_info) and whether it is a build of the
            # kernel sources beside it; self_check fails on a timing-experiment build or on a stale binary
            "build_info": dict(build, tree_csrc=tree_csrc, built_from_this_tree=build["csrc"] == tree_csrc),
        }
        if rehearsal:
            out["rehearsal"] = ("gloo rehearsal of the multi-rank branch: %d ranks share cuda:%d, collectives staged through the host -- NOT "
                                "a measurement of %d GPUs" % (world, local_rank, world))
        if self_check is not None:
            out["self_check"] = self_check
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, args.cpu_budget)
            out["config"]["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    if build["exp"] != "0" or build["csrc"] != tree_csrc:
        sys.stderr.write("bench.py: the loaded library is not a clean build of this tree's kernel sources: %s (tree csrc %s)\n" % (build, tree_csrc))
        sys.exit(1)
    if self_check is not None and not self_check["ok"]:
        sys.stderr.write("bench.py: SELF-CHECK FAILED: %s\n" % json.dumps(self_check))
        sys.exit(1)


if __name__ == "__main__":
    main()
