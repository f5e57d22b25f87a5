"""The C ABI driven by a plain C program (examples/c_host/mcpc_host.c: gcc, the HIP runtime's C API, no Python, no torch in the
process): one MCPC learning call with the fused Philox kick, the gradient bucket through the library's own RCCL path, checked
against the NumPy oracle fed with the Philox twin's normals."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import parity_log

from oracle import mcpc_oracle as mo
from oracle import philox
from oracle.cases import make_case_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "examples", "c_host", "mcpc_host")


def write_case(path, case, W, b, X0, target, T, acc_begin, loss_kind, act, lr, noise_var, loss_var, seed):
    sizes = case["sizes"]
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", 0x4d435043, len(sizes), case["n_in"], case["n_out"], case["B"], T, acc_begin, loss_kind))
        f.write(struct.pack("<6i", *([act] * len(sizes) + [0] * (6 - len(sizes)))))
        f.write(struct.pack("<6i", *(sizes + [0] * (6 - len(sizes)))))
        f.write(struct.pack("<3d", lr, noise_var, loss_var))
        f.write(struct.pack("<Q", seed))
        for w, bb in zip(W, b):
            f.write(np.ascontiguousarray(w, np.float32).tobytes())
            f.write(np.ascontiguousarray(bb, np.float32).tobytes())
        if case["n_out"]:
            f.write(np.ascontiguousarray(target, np.float32).tobytes())
        for x in X0:
            f.write(np.ascontiguousarray(x, np.float32).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("loss,sizes,n_out,batch", [("bernoulli", [12, 48, 40], 100, 50), ("gaussian", [7, 33], 21, 19)])
def test_plain_c_host_matches_oracle(tmp_path, loss, sizes, n_out, batch):
    if not os.path.exists(HOST):          # normally built by __graft_entry__.build(); the box has the same gcc
        subprocess.run(["make", "-C", os.path.dirname(HOST)], check=True, capture_output=True)
    assert os.path.exists(HOST), f"{HOST} missing: python -c 'import __graft_entry__ as g; g.build()'"
    T, acc_begin, lr, noise_var, var, seed = 14, 4, 0.03, 2.0, 0.6, 77
    case = dict(sizes=sizes, acts=["relu"] * len(sizes), ecoef=[1.0] * len(sizes), n_in=sizes[0], n_out=n_out, loss=loss, var=var,
                perc=0.5, B=batch, seed=99, x0_range=1.5, calls=[dict(T=T)])
    W, b, X0, inputs, target = make_case_inputs(case)
    kind = mo.LOSS_BERNOULLI if loss == "bernoulli" else mo.LOSS_GAUSSIAN
    net = mo.NetSpec(sizes=sizes, acts=[mo.ACT_RELU] * len(sizes), W=W, b=b, ecoef=case["ecoef"], has_head=True)
    ref = mo.run(net, inputs, X0, mo.LossSpec(kind, target, var), mo.XOpt(mo.OPT_SGD, lr), T,
                 noise=lambda t, l: philox.layer_normals(seed, t, l, 0, batch, sizes[l]), noise_var=noise_var,
                 accumulate_p_at=list(range(acc_begin, T)))
    cpath, opath = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    write_case(cpath, case, W, b, X0, target, T, acc_begin, 2 if loss == "bernoulli" else 1, 1, lr, noise_var, var, seed)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([HOST, cpath, opath], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "mcpc_host:" in run.stdout
    raw = np.fromfile(opath, dtype=np.uint8)
    off = 0
    en = raw[off:off + T * 8 * 8].view(np.float64).reshape(T, 8); off += T * 8 * 8
    group = "plain C host (examples/c_host) vs oracle"
    parity_log.close(group, "overall[t]", en[:, -1], ref.overall, rtol=1e-6, atol=1e-6)
    parity_log.close(group, "loss[t]", en[:, 0], ref.loss, rtol=1e-6, atol=1e-6)
    for l, n in enumerate(sizes):
        x = raw[off:off + batch * n * 4].view(np.float32).reshape(batch, n); off += batch * n * 4
        parity_log.close(group, "states", x, ref.xs[l], rtol=0, atol=1e-5)
    flat = raw[off:].view(np.float32)
    want = np.concatenate([np.concatenate([gw.reshape(-1), gb.reshape(-1)]) for gw, gb in zip(ref.gW, ref.gb)]) / ((T - acc_begin) * batch)
    assert flat.size == want.size
    parity_log.close(group, "gradient bucket", flat, want, rtol=2e-4, atol=2e-5 * max(1e-3, np.abs(want).max()))


def test_c_host_builds_against_the_header(tmp_path):
    """The example compiles and links with gcc against include/mcpc.h and libmcpc.so (no GPU needed to build)."""
    out = tmp_path / "mcpc_host"
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_host", "mcpc_host.c"), "-L" + os.path.join(ROOT, "montecarlopredictivecoding_amd"), "-lmcpc",
           "-L/opt/rocm/lib", "-lamdhip64", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.exists()
