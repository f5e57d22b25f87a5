#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --warmup 0 --no-secondary --no-cpu-baseline`
into profiles/hbm_traffic.json: HBM bytes per Langevin step of mcpc_steps_kernel.

MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of a
wide coalesced read stream, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
usage: collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> <out.json>"""
import csv
import json
import sys


def total(path, name):
    s = n = 0
    for r in csv.DictReader(open(path)):
        if "mcpc_steps" in r["Kernel_Name"] and r["Counter_Name"] == name:
            s += float(r["Counter_Value"]); n += 1
    return s, n


fetch, nf = total(sys.argv[1], "FETCH_SIZE")
write, nw = total(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {
    "kernel": "mcpc_steps*", "launches": nf, "steps": steps,
    "fetch_kib_raw": fetch, "write_kib_raw": write,
    "read_bytes_per_step": 2.0 * fetch * 1024 / steps, "write_bytes_per_step": write * 1024 / steps,
    "bytes_per_step": (2.0 * fetch + write) * 1024 / steps,
    "correction": "FETCH_SIZE x2 (gfx950 wide-read under-report), WRITE_SIZE as is; KiB -> bytes",
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps %d --warmup 0 --no-secondary --no-cpu-baseline" % steps,
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
