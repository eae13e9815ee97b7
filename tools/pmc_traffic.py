#!/usr/bin/env python3
"""Turns the per-kernel FETCH_SIZE / WRITE_SIZE averages of two `rocprofv3 --pmc` passes into
profiles/pmc_traffic.json (the `traffic` field of bench.py).  usage: pmc_traffic.py <pmc_summary.json> <key>"""
import json
import sys

KIB = 1024
src, key = sys.argv[1], sys.argv[2]
alg = int(sys.argv[3]) if len(sys.argv) > 3 else None
d = json.load(open(src))
out_path = "profiles/pmc_traffic.json"
try:
    rec = json.load(open(out_path))
except Exception:
    rec = {}
rec["_method"] = (
    "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only "
    "(python bench.py --steps 3 --warmup 1 --cpu-sweeps 0); per-dispatch averages in KiB. Correction per "
    "/opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of "
    "coalesced streaming reads, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact "
    "(calibrated on pr_init_kernel: two 268,435,456-byte arrays written -> WRITE_SIZE 524288 KiB).")
entry = {}
total = 0
for name, c in d.items():
    if any(k in name for k in ("pb_bin_kernel", "pb_accum_kernel", "pb_hub_kernel", "pr_tile_kernel", "pb_hot_gather")):
        short = name.split("::")[-1].split("(")[0].split("<")[0]
        b = int(2 * c.get("FETCH_SIZE", 0) * KIB + c.get("WRITE_SIZE", 0) * KIB)
        entry[short] = {"FETCH_SIZE_KiB": c.get("FETCH_SIZE"), "WRITE_SIZE_KiB": c.get("WRITE_SIZE"), "hbm_bytes": b}
        total += b
entry["hbm_bytes_per_launch"] = total
if alg:
    entry["algorithmic_bytes_per_launch"] = alg
    entry["traffic_over_algorithmic"] = round(total / alg, 3)
rec[key] = entry
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(entry, indent=1))
