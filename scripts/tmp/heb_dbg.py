import os, sys
import torch
sys.path.insert(0, os.getcwd())
from bench import make_problem, SIZES, N_OUT
from montecarlopredictivecoding_amd import _lib as L
from montecarlopredictivecoding_amd.engine import Engine
dev = torch.device("cuda", 0)
for B, T, acc0, tuning in ((64, 100, 20, None), (64, 100, 20, "no_lean=1"), (6000, 100, 20, None), (256, 40, 0, None), (64, 10, 0, None), (48, 10, 0, None)):
    W, b, y, xs = make_problem(B, 30, dev)
    eng = Engine(SIZES, [L.ACT_RELU] * 3, 30, N_OUT, B, device=dev, tuning=tuning)
    eng.bind_params(W, b); eng.bind_inputs(None); eng.bind_target(y); eng.load_state(xs)
    eng.run(T, loss_kind=L.LOSS_BERNOULLI, lr=0.03, noise_mode=L.NOISE_PHILOX, seed=1, energy_mode=L.ENERGY_ALL, acc_begin=acc0, acc_end=T)
    eng.sync_check()
    out = []
    for j in range(4):
        no, ni = eng.lin_shape(j)
        dW = torch.empty(no, ni, device=dev); db = torch.empty(no, device=dev)
        eng.read_param_grads(j, dW, db, 1.0)
        out.append((j, bool(torch.isfinite(dW).all()), float(dW.abs().max()), bool(torch.isfinite(db).all())))
    print(B, T, acc0, tuning, eng.query()["step_kernel"][:40], out, flush=True)
    eng.close()
