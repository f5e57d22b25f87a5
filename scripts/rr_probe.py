#!/usr/bin/env python3
"""Developer probe: the round schedule (tuning rr=1) against the default plan on cfg-M: same trajectories and Hebbian sums, us per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_problem, SIZES, N_OUT  # noqa: E402
from montecarlopredictivecoding_amd import _lib as L  # noqa: E402
from montecarlopredictivecoding_amd.engine import Engine  # noqa: E402
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
tunings = sys.argv[3:] or ["", "rr=1"]
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(B, 30, dev)
base = dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, lr=0.03, seed=1, noise_var=2.0)
ref = None
for tn in tunings:
    eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=dev, tuning=tn or None)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
    q = eng.query()
    # correctness: a short learning call
    eng.load_state(xs)
    res = eng.run(331, acc_begin=97, acc_end=331, rec_begin=0, rec_stride=50, rec_count=7, rec_x=True, **base)
    out = [torch.empty_like(x) for x in xs]; eng.store_state(out)
    flat = eng.read_param_grads_flat(scale=1.0); eng.sync_check()
    cur = (out, flat, res.energies.clone(), [r_.clone() for r_ in res.rec_x])
    msg = ""
    if ref is None: ref = cur
    else:
        msg = " state_bitwise=%s bucket_bitwise=%s bucket_rel=%.2e energies_rel=%.2e rec_bitwise=%s" % (
            all(torch.equal(a, c) for a, c in zip(ref[0], cur[0])), torch.equal(ref[1], cur[1]),
            float((ref[1] - cur[1]).abs().max() / ref[1].abs().max()),
            float(((ref[2] - cur[2]).abs() / ref[2].abs().clamp_min(1e-30)).max()),
            all(torch.equal(a, c) for a, c in zip(ref[3], cur[3])))
    ts = []
    for kw in ({}, dict(acc_begin=K // 5, acc_end=K)):
        best = 1e9
        for rep in range(3):
            eng.load_state(xs); eng.run(50, **base)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.run(K, **base, **kw)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K * 1e6)
        ts.append(best)
    print(os.path.basename(os.environ.get("MCPC_LIB", "libmcpc.so")), f"B={B} tuning='{tn}' ct={q['chains_per_wg']} wgs={q['n_workgroups']}  inference {ts[0]:6.1f} | learning {ts[1]:6.1f} us/step{msg}", flush=True)
    eng.close()
