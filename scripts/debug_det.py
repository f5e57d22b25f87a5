import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_fullsize import _problem, _engine, _run, B
W, b, y, xs = _problem()
runs = []
for tuning in (None, None, "no_ybits=1", "no_lean=1", "no_overlap=1"):
    eng = _engine(B, W, b, y, tuning=tuning)
    res, out = _run(eng, xs, 90, acc_begin=20, acc_end=90)
    runs.append(eng.read_param_grads_flat().cpu().numpy()); eng.close()
names = []
for j in range(4):
    names += [(f"W{j}", W[j].numel()), (f"b{j}", b[j].numel())]
for k, lab in ((1, "rerun"), (2, "no_ybits"), (3, "no_lean"), (4, "no_overlap")):
    off = 0
    for nm, n in names:
        a, c = runs[0][off:off + n], runs[k][off:off + n]
        nd = int((a != c).sum())
        if nd: print(lab, nm, "differs in", nd, "of", n, "max|d|", np.abs(a - c).max(), "max|ref|", np.abs(a).max(), "first", int(np.argmax(a != c)))
        off += n
print("done")
