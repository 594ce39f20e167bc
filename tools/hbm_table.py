#!/usr/bin/env python3
"""Achieved HBM GB/s per kernel: PMC bytes per launch (profiles/<tag>_pmc_traffic_<w>.json) over the kernel's
average duration in the rocprofv3 kernel trace of the same build (profiles/<tag>_<w>_kernel_stats.csv).
    python tools/hbm_table.py [tag, default r03]      -> a markdown table"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
PEAK = 8000.0


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name).strip()


print("| workload | kernel | avg µs | HBM MB / launch | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for w in ("cloth1m", "boxes1m", "sort16m"):
    pj = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_{w}.json")
    ks = os.path.join(ROOT, "profiles", f"{tag}_{w}_kernel_stats.csv")
    apart = os.path.join(ROOT, "profiles", f"{tag}_{w}_passes_apart_kernel_stats.csv")  # one kernel at a time on the chip
    if os.path.exists(apart):
        ks = apart
    if not (os.path.exists(pj) and os.path.exists(ks)):
        continue
    traffic = {short(k): v["hbm_bytes_per_launch_corrected"] for k, v in json.load(open(pj))["kernels"].items()}
    rows = []
    for r in csv.DictReader(open(ks)):
        k = short(r["Name"])
        if k in traffic and not k.startswith("__amd") and not k.startswith("at::") and "elementwise" not in k and "mesh_index" not in k and "pack_" not in k:
            us = float(r["AverageNs"]) / 1e3
            rows.append((float(r["TotalDurationNs"]), k, us, traffic[k]))
    for _, k, us, b in sorted(rows, reverse=True):
        gbs = b / (us * 1e-6) / 1e9
        print(f"| {w} | `{k}` | {us:.1f} | {b / 1e6:.1f} | {gbs:.0f} | {gbs / PEAK:.2f} |")
