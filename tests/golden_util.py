"""Load tests/golden/*.npz fixtures and turn them into oracle / engine inputs."""
import glob
import json
import os

import numpy as np

from oracle import mcpc_oracle as mo
from oracle.cases import make_case_inputs, call_noise

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ACT = {"identity": mo.ACT_IDENTITY, "relu": mo.ACT_RELU, "tanh": mo.ACT_TANH}


def fixture_names(prefix="", exclude=()):
    """Fixture names; ``exclude`` drops prefixes (g9_* exercise trainer control flow -- per-step parameter updates,
    dynamic x learning rate, user callbacks, custom x optimizers -- that only the facade replays, not the
    closed-form oracle or a single mcpc_run)."""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    # g10_* / g11_* / g12_* are the evaluator, checkpoint and learning-run fixtures of oracle/gen_golden_eval.py: another
    # format, consumed by tests/test_evaluators_golden.py and tests/test_gpu_widening.py / test_gpu_learning.py
    # g13_* (stationary statistics of the reference's own sampler, oracle/gen_golden_sampling.py) and g14_* (get_representations)
    # are consumed by tests/test_gpu_sampling.py and tests/test_sampling_fixtures.py
    # g15_* (the keyword surface outside the kernels, oracle/gen_golden_generic.py) are consumed by tests/test_generic_loop.py
    exclude = tuple(exclude) + ("g10_", "g11_", "g12_", "g13_", "g14_", "g15_")
    return [n for n in names if not any(n.startswith(x) for x in exclude)]


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.case = json.loads(str(self.z["case_json"]))
        c = self.case
        if "inputs" in self.z:
            nlin = len(c["sizes"]) + (1 if c["n_out"] else 0)
            self.W = [self.z[f"W{j}"] for j in range(nlin)]
            self.b = [self.z[f"b{j}"] if f"b{j}" in self.z else None for j in range(nlin)]
            self.X0 = [self.z[f"X0_{l}"] for l in range(len(c["sizes"]))]
            self.inputs = self.z["inputs"]
            self.target = self.z["target"] if "target" in self.z else None
        else:   # seed-derived inputs (big nets)
            self.W, self.b, self.X0, self.inputs, self.target = make_case_inputs(c)

    def loss_spec(self):
        c = self.case
        kind = c["loss"]
        if kind in ("none", "zero"):
            return mo.LossSpec(mo.LOSS_NONE)
        k = mo.LOSS_GAUSSIAN if kind.startswith("gaussian") else mo.LOSS_BERNOULLI
        ms = mo.mask_start_from_perc(c["n_out"], c["perc"]) if kind.endswith("_mask") else 0
        return mo.LossSpec(k, self.target, c["var"], ms)

    def net(self, W=None, b=None):
        c = self.case
        return mo.NetSpec(sizes=list(c["sizes"]), acts=[ACT[a] for a in c["acts"]],
                          W=self.W if W is None else W, b=self.b if b is None else b,
                          ecoef=list(c["ecoef"]), has_head=bool(c["n_out"]))

    def xopt(self, call):
        return mo.XOpt(mo.OPT_ADAM if call["xopt"] == "adam" else mo.OPT_SGD, call["lr"])

    def schedules(self, call):
        T = call["T"]

        def conv(v):
            if v == "never":
                return []
            if v == "last":
                return [T - 1]
            if v == "all":
                return list(range(T))
            return list(v)
        return conv(call.get("update_p_at", "never")), conv(call.get("accumulate_p_at", "never"))

    def noise(self, ci):
        call = self.case["calls"][ci]
        return call_noise(self.case, ci) if call.get("noise", False) else None

    def get(self, ci, key):
        return self.z[f"c{ci}_{key}"]

    def has(self, ci, key):
        return f"c{ci}_{key}" in self.z
