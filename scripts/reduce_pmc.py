#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection.csv files to per-kernel sums, and merge the passes of scripts/pmc_round.sh.

    reduce_pmc.py <counter_collection.csv> <out.json>       one pass -> {kernel: {dispatches, counters: {name: sum}}}
    reduce_pmc.py --merge <dir> <out.json>                  all sum_<mode>_<group>.json of a round + derived ratios

Derived figures follow MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles),
SQ_VALU_MFMA_BUSY_CYCLES counts cycles; FETCH_SIZE / WRITE_SIZE are KiB and FETCH_SIZE is doubled on gfx950."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("mcpc::", "")


def reduce_one(path, out):
    acc = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k.startswith("mcpc_"):
            continue
        d = acc.setdefault(k, {"dispatch_ids": set(), "counters": {}})
        d["dispatch_ids"].add(r["Dispatch_Id"])
        d["counters"][r["Counter_Name"]] = d["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    res = {k: {"dispatches": len(v["dispatch_ids"]), "counters": v["counters"]} for k, v in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(out, {k: v["dispatches"] for k, v in res.items()})


def merge(d, out):
    res = {}
    for path in sorted(glob.glob(os.path.join(d, "sum_*_*.json"))):
        m = re.match(r"sum_(learning|inference)_(.*)\.json", os.path.basename(path))
        mode = m.group(1)
        for k, v in json.load(open(path)).items():
            e = res.setdefault(mode, {}).setdefault(k, {"dispatches": v["dispatches"], "counters": {}})
            e["counters"].update(v["counters"])
    for mode, kernels in res.items():
        for k, e in kernels.items():
            c = e["counters"]
            dv = {}
            if c.get("SQ_BUSY_CYCLES"):
                # SQ_BUSY_CYCLES: per-SE busy cycles summed over the shader engines (guide: 8 XCDs x ... report the ratio only)
                pass
            if c.get("SQ_WAVE_CYCLES"):
                wc = 4.0 * c["SQ_WAVE_CYCLES"]
                dv["wave_cycles"] = wc
                for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                    if nm in c:
                        dv[nm.lower() + "_frac_of_wave_cycles"] = 4.0 * c[nm] / wc
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                    dv["mfma_busy_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
                if "SQ_INSTS_VALU_MFMA_F32" in c:
                    dv["mfma_f32_insts"] = c["SQ_INSTS_VALU_MFMA_F32"]
                    dv["mfma_ideal_cycles_32_per_inst"] = 32.0 * c["SQ_INSTS_VALU_MFMA_F32"]
            if c.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in c:
                dv["valu_insts_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
                dv["valu_mfma_coexec_cycles"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"]
            if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
                dv["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c and c["TCP_TOTAL_CACHE_ACCESSES_sum"] > 0:
                dv["l1_read_requests_to_l2_per_access"] = c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
            if "FETCH_SIZE" in c:
                dv["hbm_read_bytes"] = 2.0 * 1024.0 * c["FETCH_SIZE"]
            if "WRITE_SIZE" in c:
                dv["hbm_write_bytes"] = 1024.0 * c["WRITE_SIZE"]
            e["derived"] = dv
    # meta: what the passes profiled, from THEIR OWN bench lines (<mode>_<group>.json next to the sums): bench.py prints
    # roofline.traffic only when this matches the run it is printed in (kernel sources, batch, T, launches)
    meta = {}
    for path in sorted(glob.glob(os.path.join(d, "*_*.json"))):
        if os.path.basename(path).startswith("sum_"):
            continue
        try:
            line = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
        except (IndexError, ValueError):
            continue
        b, cfg = line.get("build_info") or {}, line.get("config") or {}
        this = {"csrc": b.get("csrc"), "commit": b.get("commit"), "exp": b.get("exp"), "batch": cfg.get("batch"), "T": cfg.get("T")}
        if meta and any(meta[k] != v for k, v in this.items()):
            raise SystemExit(f"the passes under {d} profiled different programs: {meta} vs {this} ({path})")
        meta.update(this)
    meta["command"] = ("rocprofv3 --pmc <one group per pass> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-self-check "
                       "{--no-secondary | --only-inference} (scripts/pmc_round.sh)")
    meta["corrections"] = "FETCH_SIZE and WRITE_SIZE in KiB -> bytes; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM traffic from PMC)"
    res["meta"] = meta
    json.dump(res, open(out, "w"), indent=1)
    print("merged ->", out, meta)


if __name__ == "__main__":
    if sys.argv[1] == "--merge":
        merge(sys.argv[2], sys.argv[3])
    else:
        reduce_one(sys.argv[1], sys.argv[2])
