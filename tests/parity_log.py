"""Achieved errors of the parity comparisons, written where a report can be made of them.

`close(group, quantity, got, want, rtol, atol)` asserts like numpy.testing.assert_allclose AND appends one JSON line with the
achieved error to gpurun_out/parity_errors.jsonl (MCPC_PARITY_LOG overrides the path; scripts/parity_report.py reduces the file to
profiles/r04_parity_errors.txt: per group and quantity the largest achieved error beside the tolerance the test states)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("MCPC_PARITY_LOG") or os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl")


def close(group, quantity, got, want, rtol=0.0, atol=0.0, err_msg=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (group, quantity, got.shape, want.shape)
    diff = np.abs(got - want)
    finite = np.isfinite(diff)
    max_abs = float(diff[finite].max()) if finite.any() else 0.0
    scale = float(np.abs(want).max()) if want.size else 0.0
    # "excess": the largest |got - want| / (atol + rtol |want|): <= 1 passes; what a tolerance k times tighter would need is excess * k
    allowed = atol + rtol * np.abs(want)
    with np.errstate(divide="ignore", invalid="ignore"):
        excess = float(np.nanmax(np.where(allowed > 0, diff / allowed, np.where(diff > 0, np.inf, 0.0)))) if want.size else 0.0
        rel = diff / np.abs(want)
        max_rel = float(np.nanmax(np.where(np.abs(want) > 1e-3 * max(scale, 1e-30), rel, 0.0))) if want.size else 0.0
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        with open(PATH, "a") as f:
            f.write(json.dumps({"group": group, "quantity": quantity, "n": int(want.size), "max_abs": max_abs, "max_rel": max_rel,
                                "scale": scale, "rtol": float(rtol), "atol": float(atol), "excess": excess,
                                "all_finite": bool(finite.all())}) + "\n")
    except OSError:
        pass
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=f"{group}: {quantity} {err_msg}")
