#!/usr/bin/env python3
"""Turns the per-kernel FETCH_SIZE / WRITE_SIZE averages of two `rocprofv3 --pmc` passes into
profiles/pmc_traffic.json (the `traffic` field of bench.py).  usage: pmc_traffic.py <pmc_summary.json> <key> [algorithmic bytes] [note]"""
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
    "(python bench.py --steps 3 --warmup 1 --cpu-sweeps 0); per-SWEEP sums in KiB (a kernel's dispatches of the run / the number of sweeps). Correction per "
    "/opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of "
    "coalesced streaming reads, so fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact "
    "(calibrated on pr_init_kernel: two 268,435,456-byte arrays written -> WRITE_SIZE 524288 KiB).")
entry = {}
total = 0
# one sweep = one pb_accum_kernel (or pr_tile_kernel) dispatch; kernels launched several times per sweep (round 3's pb_hub_kernel: the
# long chains' first blocks, their fall-back, the other groups) count with all their dispatches.  pb_bin_kernel also runs
# outside sweeps (the timed placement draws of pb_scratch_create): its per-dispatch average is what one sweep moves.
sweeps = max([c.get("FETCH_SIZE_dispatches", 0) for name, c in d.items() if "pb_accum_kernel" in name or "pr_tile_kernel" in name] or [0])
for name, c in d.items():
    if any(k in name for k in ("pb_bin_kernel", "pb_accum_kernel", "pb_hub_kernel", "pb_hubchain", "pb_hubseq_kernel", "pb_hublong_kernel", "pr_tile_kernel", "pb_hot_gather")):
        short = name.split("::")[-1].split("(")[0].split("<")[0]
        if sweeps and "FETCH_SIZE_total" in c and "pb_bin_kernel" not in name:
            f, w = c.get("FETCH_SIZE_total", 0) / sweeps, c.get("WRITE_SIZE_total", 0) / sweeps
            per = c.get("FETCH_SIZE_dispatches", 0) / sweeps
        else:
            f, w, per = c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0), 1
        b = int(2 * f * KIB + w * KIB)
        e = entry.setdefault(short, {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "hbm_bytes": 0, "dispatches_per_sweep": 0.0})
        e["FETCH_SIZE_KiB"] += f
        e["WRITE_SIZE_KiB"] += w
        e["hbm_bytes"] += b
        e["dispatches_per_sweep"] += per
        total += b
entry["hbm_bytes_per_launch"] = total
# the record belongs to ONE library: bench.py quotes it only when the library it loaded has this hash
import hashlib, os, time
lib_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graph_amd", "libgraph_mi355x.so")
entry["library_sha256"] = hashlib.sha256(open(lib_path, "rb").read()).hexdigest()
entry["measured"] = time.strftime("%Y-%m-%d") + (f", {sys.argv[4]}" if len(sys.argv) > 4 else "")
if alg:
    entry["algorithmic_bytes_per_launch"] = alg
    entry["traffic_over_algorithmic"] = round(total / alg, 3)
rec[key] = entry
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(entry, indent=1))
