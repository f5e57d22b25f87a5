#!/usr/bin/env python3
"""gpurun_out/parity_errors.jsonl (written by the GPU tests through tests/parity_log.py) -> a table: per test group and quantity the
number of comparisons, the largest achieved absolute / relative error and the tolerance the tests state.
    python scripts/parity_report.py [in.jsonl] > profiles/r04_parity_errors.txt"""
import collections
import json
import os
import sys

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_errors.jsonl")
rows = collections.OrderedDict()
for line in open(path):
    r = json.loads(line)
    k = (r["group"], r["quantity"])
    a = rows.setdefault(k, dict(n=0, elems=0, max_abs=0.0, max_rel=0.0, excess=0.0, rtol=set(), atol=set(), scale=0.0, finite=True))
    a["n"] += 1; a["elems"] += r["n"]
    a["max_abs"] = max(a["max_abs"], r["max_abs"]); a["max_rel"] = max(a["max_rel"], r["max_rel"]); a["excess"] = max(a["excess"], r["excess"])
    a["rtol"].add(r["rtol"]); a["atol"].add(round(r["atol"], 12)); a["scale"] = max(a["scale"], r["scale"]); a["finite"] &= r["all_finite"]
print("# Achieved errors of the GPU parity comparisons (one MI355X; tests/parity_log.py).  excess = max |got - want| / (atol + rtol |want|):")
print("# the fraction of the stated tolerance the worst element used (<= 1 passes).  max_rel is over elements above 1e-3 of the largest |want|.")
print(f"{'group':58s} {'quantity':22s} {'cmp':>5s} {'elements':>10s} {'max_abs':>10s} {'max_rel':>10s} {'|want|max':>10s} {'rtol':>14s} {'atol':>20s} {'excess':>7s}")
for (g, q), a in rows.items():
    rt = "/".join(f"{v:g}" for v in sorted(a["rtol"])); at = "/".join(f"{v:.3g}" for v in sorted(a["atol"])[:3]) + ("..." if len(a["atol"]) > 3 else "")
    print(f"{g[:58]:58s} {q[:22]:22s} {a['n']:5d} {a['elems']:10d} {a['max_abs']:10.3g} {a['max_rel']:10.3g} {a['scale']:10.3g} {rt:>14s} {at:>20s} {a['excess']:7.3f}"
          + ("" if a["finite"] else "  NON-FINITE"))
