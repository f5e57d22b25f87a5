#!/usr/bin/env python3
"""Reproducer of round 3's intermittent NaN, deterministic.

    python scripts/nan_repro.py LIB [LIB ...]      (each LIB: a libmcpc build, e.g. scripts/bin/libmcpc_71cd49e.so)

What happened.  The bf16x6 GEMM core reads its LDS operand in whole 32-deep k-blocks.  A row whose width is 16 mod 32 (200 -> 208
padded) is over-read by 12 floats; the weights there are zeros, so the products were exact zeros AS LONG AS the over-read values
were finite.  For the LAST row of the LAST operand region of a plan the over-read left the plan: it returned whatever an earlier
kernel had left in that CU's LDS.  0 x NaN = NaN entered the back-projection of that row's chain.

This script makes "whatever an earlier kernel left" deterministic: libmcpc's diagnostic entry mcpc_debug_poison_lds (HEAD's build,
always) fills the LDS of every CU with signalling NaNs before every engine call, then runs the net of the failing test
(tests/test_gpu_fuzz.py::test_wide_networks_against_oracle[sizes0-784]: 40-384-200-784) on the library under test through a minimal
ctypes driver of its own (the C ABI's structs did not change between ABI 2 and 3), in the workgroup forms the library has:
  * B = 40, the test's batch: the over-reading row is a PADDING chain of the second 32-chain workgroup -- its NaN reached the
    energies while padding chains were masked by a product with 0 (before 4cbbd6a), i.e. the symptom the suite showed;
  * B = 64: the row is a LIVE chain -- its state goes NaN on every build before 992f3d2, whatever the energy mask.
Prints one line per (library, form, batch): how many energies / state values came out non-finite with the LDS poisoned (poison=1)
and with the LDS of every CU cleared to zeros instead (poison=0: what the same call finds on a chip nobody has used before -- the reason
the failure was intermittent).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlopredictivecoding_amd import _lib as L          # noqa: E402  (struct layouts, constants)
from oracle.cases import make_case_inputs                      # noqa: E402

DEV = torch.device("cuda", 0)
SIZES, N_OUT, T = [40, 384, 200], 784, 5


def ptr(t):
    return C.c_void_p(t.data_ptr())


def run_case(lib, tuning, batch, poison):
    head = C.CDLL(L.LIB_PATH)                   # HEAD's build: the poison entry
    head.mcpc_debug_poison_lds.argtypes = [C.c_int, C.c_uint32, C.c_void_p]
    case = dict(sizes=SIZES, acts=["relu"] * 3, ecoef=[1.0] * 3, n_in=SIZES[0], n_out=N_OUT, loss="bernoulli", var=1.0, perc=0.5,
                B=batch, seed=77, x0_range=1.0, calls=[dict(T=T)])
    W, b, X0, inputs, target = make_case_inputs(case)
    d = L.NetDesc()
    d.abi_version = lib.mcpc_abi_version()
    d.n_latent, d.n_in, d.n_out, d.batch, d.device = 3, SIZES[0], N_OUT, batch, 0
    for i, n in enumerate(SIZES):
        d.sizes[i] = n; d.acts[i] = L.ACT_RELU; d.ecoef[i] = 1.0
    d.tuning = tuning.encode() if tuning else None
    h = C.c_void_p()
    if lib.mcpc_create(C.byref(d), C.byref(h)) != 0:
        lib.mcpc_last_error.restype = C.c_char_p
        return "create failed: " + lib.mcpc_last_error().decode()
    keep = []
    for j in range(4):
        w, bb = torch.from_numpy(W[j]).to(DEV), torch.from_numpy(b[j]).to(DEV)
        keep += [w, bb]
        lib.mcpc_bind_params(h, j, ptr(w), ptr(bb))
    lib.mcpc_params_changed(h, None)
    lib.mcpc_bind_inputs(h, None, None)
    y = torch.from_numpy(target).to(DEV)
    lib.mcpc_bind_target(h, ptr(y), None)
    xs = [torch.from_numpy(x).to(DEV) for x in X0]
    arr = (C.c_void_p * 3)(*[x.data_ptr() for x in xs])
    lib.mcpc_load_state(h, arr, None)
    en = torch.zeros(T, L.ENERGY_COLS, dtype=torch.float64, device=DEV)
    r = L.RunDesc()
    r.T, r.t_begin, r.n_steps = T, 0, T
    r.loss_kind, r.loss_var, r.xopt_kind, r.lr = L.LOSS_BERNOULLI, 1.0, L.XOPT_SGD, 0.02
    r.beta1, r.beta2, r.eps = 0.9, 0.999, 1e-8
    r.update_x, r.noise_mode, r.noise_var, r.seed = 1, L.NOISE_PHILOX, 2.0, 9
    r.acc_begin, r.acc_end, r.acc_reset = 1, T, 1
    r.energy_mode, r.energies_out = L.ENERGY_ALL, en.data_ptr()
    torch.cuda.synchronize()
    # poison = 1: signalling NaNs; poison = 0: zeros (a CLEAN chip: what a process finds when no other kernel has used the LDS before)
    rc = head.mcpc_debug_poison_lds(0, 0x7FA00000 if poison else 0, None)
    assert rc == 0, rc
    rc = lib.mcpc_run(h, C.byref(r), None)
    lib.mcpc_store_state(h, arr, None)
    torch.cuda.synchronize()
    q = [C.c_int32() for _ in range(4)]
    lib.mcpc_query(h, *[C.byref(v) for v in q])
    q = [v.value for v in q]
    lib.mcpc_destroy(h)
    bad_e = int((~torch.isfinite(en)).sum().item())
    bad_x = sum(int((~torch.isfinite(x)).sum().item()) for x in xs)
    steps = [int(t) for t in torch.nonzero(~torch.isfinite(en[:, -1])).flatten().tolist()]
    return f"rc={rc} chains/wg={q[1]:2d} lds={q[0]:6d}: non-finite energies {bad_e:3d} (overall NaN at steps {steps}), non-finite state values {bad_x}"


def main():
    torch.cuda.init()
    for path in sys.argv[1:] or [L.LIB_PATH]:
        lib = C.CDLL(os.path.abspath(path))
        lib.mcpc_create.argtypes = [C.POINTER(L.NetDesc), C.POINTER(C.c_void_p)]
        lib.mcpc_run.argtypes = [C.c_void_p, C.POINTER(L.RunDesc), C.c_void_p]
        for fn in ("mcpc_bind_params", "mcpc_params_changed", "mcpc_bind_inputs", "mcpc_bind_target", "mcpc_load_state", "mcpc_store_state",
                   "mcpc_destroy", "mcpc_query"):
            getattr(lib, fn).argtypes = None
        lib.mcpc_bind_params.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.mcpc_params_changed.argtypes = [C.c_void_p, C.c_void_p]
        lib.mcpc_bind_inputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mcpc_bind_target.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mcpc_load_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mcpc_store_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mcpc_destroy.argtypes = [C.c_void_p]
        lib.mcpc_query.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        abi = lib.mcpc_abi_version()
        tunings = [None, "ws=0"] + (["ws=2"] if abi < 3 else ["overlay16=1"])      # ABI 2: ws=2 = the 32-chain in-place form
        for tuning in tunings:
            for batch in (40, 64):
                for poison in (False, True):
                    print(f"{os.path.basename(path):28s} tuning={str(tuning):12s} B={batch:3d} poison={int(poison)}: {run_case(lib, tuning, batch, poison)}", flush=True)


if __name__ == "__main__":
    main()
