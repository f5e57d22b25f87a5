#!/usr/bin/env python3
"""How much of the Hebbian flush runs BESIDE the step kernel, and how concurrent the two halves of a mixed segment are: from a
rocprofv3 --kernel-trace CSV (start / end timestamps per dispatch, hardware queue per dispatch).

    overlap_from_trace.py <kernel_trace.csv>      -> a few lines of text (committed under profiles/ by hand)"""
import csv
import sys
from collections import defaultdict


def union_length(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def intersect_length(a, b):
    """Total time during which some interval of a AND some interval of b are open (a, b: lists of (s, e))."""
    ev = [(s, 0, 1) for s, e in a] + [(e, 0, -1) for s, e in a] + [(s, 1, 1) for s, e in b] + [(e, 1, -1) for s, e in b]
    ev.sort()
    open_ = [0, 0]
    last, tot = None, 0
    for t, which, d in ev:
        if last is not None and open_[0] > 0 and open_[1] > 0:
            tot += t - last
        open_[which] += d
        last = t
    return tot


def main(path):
    groups = defaultdict(list)
    queues = defaultdict(set)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        key = None
        if "mcpc_steps_ws2_kernel<2, false>" in n:
            key = "K1 plain <2,false>"
        elif "mcpc_steps_ws2_mixed_kernel" in n:
            key = "K1 mixed schedule (one launch per segment)"
        elif "mcpc_steps_ws2_kernel<2, true>" in n:
            key = "K1 mixed, paired half <2,true>"
        elif "mcpc_steps_ws2_kernel<1, true>" in n:
            key = "K1 mixed, split half <1,true>"
        elif "mcpc_heb_kernel" in n or "mcpc_reduce_jobs" in n or "mcpc_dw_kernel" in n:
            key = "Hebbian flush (heb + reduce)"
        if key:
            groups[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            queues[key].add(r["Queue_Id"])
    for k, iv in groups.items():
        print(f"{k:34s}: {len(iv):5d} dispatches, busy {union_length(iv) / 1e6:9.2f} ms, hardware queue(s) {sorted(queues[k])}")
    a, f = groups.get("K1 plain <2,false>", []), groups.get("Hebbian flush (heb + reduce)", [])
    if a and f:
        both = intersect_length(a, f)
        print(f"flush time that runs beside the plain step kernel: {both / 1e6:.2f} of {union_length(f) / 1e6:.2f} ms = {both / union_length(f):.1%}; "
              f"step-kernel time with a flush beside it: {both / union_length(a):.1%}")
    p, s = groups.get("K1 mixed, paired half <2,true>", []), groups.get("K1 mixed, split half <1,true>", [])
    if p and s:
        both = intersect_length(p, s)
        print(f"mixed schedule: both halves in flight for {both / 1e6:.2f} ms of {union_length(p + s) / 1e6:.2f} ms = {both / union_length(p + s):.1%} "
              f"(paired half busy {union_length(p) / 1e6:.2f} ms, split half {union_length(s) / 1e6:.2f} ms)")


if __name__ == "__main__":
    main(sys.argv[1])
