import sys, warnings, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import gen_golden, philox, mcpc_oracle as mo
from tests.golden_util import Golden
import montecarlopredictivecoding_amd.predictive_coding as pc
import montecarlopredictivecoding_amd.utils.model as um
g = Golden("g2_cfgM_b64"); case = dict(g.case); case.pop("rec_chains", None); call = dict(case["calls"][0])
up, acc = g.schedules(call)
T = call["T"]
ref = mo.run(g.net(), g.inputs, g.X0, g.loss_spec(), g.xopt(call), T, noise=g.noise(0), accumulate_p_at=acc, record_at=list(range(T)))
dev = "cuda:0"
model, lins = gen_golden.build_reference_model(pc, case, g.W, g.b, g.X0, device=dev)
XI = [[philox.layer_normals(case["seed"], t, l, 0, case["B"], n) for l, n in enumerate(case["sizes"])] for t in range(T)]
call["record_at"] = list(range(T))
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    out, tr = gen_golden.run_reference_call(pc, um, model, call, g.inputs, g.target, XI, case, device=dev)
print(tr.last_call_mode, sorted(out)[:8])
for t in (0, 9, 30, 50, 60, 66, 67, 68, 69, 70, 72, 80, 99):
    dx = [float(np.abs(out[f"x_t{t}_l{l}"] - ref.rec_xs[t][l]).max()) for l in range(3)]
    where = [np.unravel_index(np.argmax(np.abs(out[f"x_t{t}_l{l}"] - ref.rec_xs[t][l])), ref.rec_xs[t][l].shape) for l in range(3)]
    print(t, "max|dx| per layer", dx, where, "energy rel", abs(out["energy"][t] - ref.energy[t]) / ref.energy[t], "loss rel", abs(out["loss"][t] - ref.loss[t]) / ref.loss[t])
