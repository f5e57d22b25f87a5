import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bench import make_problem, SIZES, N_OUT
from montecarlopredictivecoding_amd import _lib as L
from montecarlopredictivecoding_amd.engine import Engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(6000, 30, dev)
eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, 6000, device=dev)
eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
kw = dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, acc_begin=0)
eng.load_state(xs); eng.run(100, lr=0.03, seed=1, acc_end=100, **kw); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.run(K, lr=0.03, seed=1, acc_end=K, **kw); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"learning {os.environ.get('TAG','')}: {dt / K * 1e6:8.1f} us/step  slots={eng.query()['spill_slots']}", flush=True)
