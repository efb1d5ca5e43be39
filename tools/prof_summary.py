#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs of tools/profile_bench.sh into one JSON: per kernel (name matching the
filter), the mean per-launch value of every counter, plus the kernel-trace statistics; derived figures
(VALU issue fraction, lanes per VALU instruction, HBM-side bytes, L2 hit rate) are added per kernel.
Usage: tools/prof_summary.py <profile dir> <summary.json> [--filter REGEX] [--meta key=value ...]"""
import collections
import csv
import glob
import json
import os
import re
import sys

SHADER_GHZ, N_SIMD = 2.4, 1024


def main():
    src, dst = sys.argv[1], sys.argv[2]
    filt = r"trace_kernel|str_|or_|k_triangle_interactions|wf_|film_"
    meta = {}
    args = sys.argv[3:]
    while args:
        a = args.pop(0)
        if a == "--filter":
            filt = args.pop(0)
        elif a == "--meta":
            while args and "=" in args[0] and not args[0].startswith("--"):
                k, v = args.pop(0).split("=", 1)
                meta[k] = int(v) if v.lstrip("-").isdigit() else v
    rx = re.compile(filt)
    out = dict(meta)
    out.update({"counters_per_launch_mean": {}, "kernel_stats": [], "derived": {}})

    def short(k):
        return k.replace("void nnbvh::", "").replace("nnbvh::", "").split("(")[0]

    for f in sorted(glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not rx.search(k):
                continue
            agg[(short(k), r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"])][r["Counter_Name"]].append(
                float(r["Counter_Value"]))
        for (k, grid, vgpr, lds), cs in agg.items():
            d = out["counters_per_launch_mean"].setdefault(k, {"grid": grid, "vgpr": vgpr, "lds": lds})
            d.update({c: sum(v) / len(v) for c, v in cs.items()})
            d["launches_profiled"] = len(next(iter(cs.values())))
    for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
        out["kernel_stats"] = list(csv.DictReader(open(f)))
    for k, c in out["counters_per_launch_mean"].items():
        ks = next((s for s in out["kernel_stats"] if short(s["Name"]) == k), None)
        d = {}
        if ks:
            ns = float(ks["AverageNs"])
            d["avg_launch_ms"] = round(ns / 1e6, 4)
            d["launches"] = int(ks["Calls"])
            if "SQ_INSTS_VALU" in c:
                d["valu_issue_frac"] = round(c["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * ns * SHADER_GHZ), 4)
        if "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
            d["lanes_per_valu"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 2)
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            d["wait_frac_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        if "SQ_INSTS_SALU" in c and c.get("SQ_INSTS_VALU"):
            d["salu_per_valu"] = round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3)
        if "TCC_EA0_RDREQ_sum" in c and "WRITE_SIZE" in c:
            rd = c["TCC_EA0_RDREQ_sum"] * 128 - c.get("TCC_EA0_RDREQ_32B_sum", 0) * 96 - c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64
            d["hbm_read_bytes"], d["hbm_write_bytes"] = round(rd), round(c["WRITE_SIZE"] * 1024)
            if ks:
                d["hbm_tb_s"] = round((rd + c["WRITE_SIZE"] * 1024) / float(ks["AverageNs"]) / 1e3, 3)
        if "TCC_HIT_sum" in c:
            d["l2_hit"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
        if c.get("TCP_TCC_READ_REQ_sum"):
            d["l1_to_l2_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
        out["derived"][k] = d
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
    for k, d in out["derived"].items():
        c = out["counters_per_launch_mean"][k]
        print(k, f"vgpr {c['vgpr']} lds {c['lds']} grid {c['grid']}", json.dumps(d))


if __name__ == "__main__":
    main()
