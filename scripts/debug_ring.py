"""Developer probe: which tensor of the Hebbian bucket differs between ring configurations (see tests/test_gpu_fullsize.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_fullsize import _problem, _engine, _run, B, SIZES, N_OUT

W, b, y, xs = _problem()
T, acc0 = int(sys.argv[1]) if len(sys.argv) > 1 else 420, 20
runs = {}
cfgs = [("overlap", None), ("serial", "no_overlap=1,slot_cap=64"), ("nowrap", "no_overlap=1,slot_cap=448,spill_gb=24"),
        ("nowrap2", "no_overlap=1,slot_cap=448,spill_gb=24,dw_ksplit=64")]
for key, tuning in cfgs:
    eng = _engine(B, W, b, y, tuning=tuning)
    res, out = _run(eng, xs, T, acc_begin=acc0, acc_end=T)
    runs[key] = eng.read_param_grads_flat().cpu().numpy()
    print(key, eng.query())
    eng.close()
names = []
for j in range(4):
    names += [(f"W{j}", W[j].numel()), (f"b{j}", b[j].numel())]
for other in ("serial", "nowrap", "nowrap2"):
    off = 0
    for nm, n in names:
        a, c = runs["overlap"][off:off + n], runs[other][off:off + n]
        d = np.abs(a - c).max()
        print(f"overlap vs {other:8s} {nm}: max|diff| {d:.4g}  max|ref| {np.abs(c).max():.4g}  first bad idx {np.argmax(np.abs(a-c))}")
        off += n
