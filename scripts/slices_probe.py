import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import make_problem, SIZES, N_OUT
from montecarlopredictivecoding_amd import _lib as L
from montecarlopredictivecoding_amd.engine import Engine
dev = torch.device("cuda", 0)
W, b, y, xs = make_problem(6000, 30, dev)
eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, 6000, device=dev, tuning="no_mix=1")
eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y)
kw = dict(noise_mode=L.NOISE_PHILOX, loss_kind=L.LOSS_BERNOULLI, energy_mode=L.ENERGY_ALL, lr=0.03, seed=1)
T = 1280
for sl in (1280, 256, 64, 16):
    best = 1e9
    for rep in range(3):
        eng.load_state(xs); torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(0, T, sl):
            eng.run(T, t_begin=t, n_steps=sl, **kw)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / T * 1e6)
    print(f"plain schedule, inference, launches of {sl} steps: {best:.2f} us/step", flush=True)
