#!/usr/bin/env python3
"""HBM traffic per API call of WCC / SSSP / triangle count from the per-call counter sums of tools/algos_profile.py
(ALGOS_PROFILE_JSON), stamped with the sha256 of the library that was measured (tools/bench_algos.py quotes the record only
when that is the library it loaded).  Corrections as for the PageRank record (tools/pmc_traffic.py, MI355X_MICROARCH.md
"HBM"): FETCH_SIZE counts a 128-byte request as 64 bytes on gfx950 -> doubled; WRITE_SIZE as reported (KiB).

    algos_traffic.py <algos_profile.json> <note>   ->  profiles/algos_traffic.json"""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof, note = json.load(open(sys.argv[1])), sys.argv[2] if len(sys.argv) > 2 else ""
lib = os.path.join(ROOT, "graph_amd", "libgraph_mi355x.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
LABELS = {"wcc": "wcc steady", "sssp": "sssp steady (call 3+)", "tc": "tc steady"}
OURS = ("wcc_", "sssp_", "tc_")
out = {"library_sha256": sha, "measured": note,
       "corrections": "FETCH_SIZE x 2 (128-byte requests tallied as 64 on gfx950: calibrated for these kernels' access patterns — random "
                      "128-byte records, random 4-byte probes plain and sc1, 8 B/lane streams — in profiles/r06_fetch_size_calibration.txt), "
                      "WRITE_SIZE as reported (a random 4-byte store = one 32-byte write request); KiB -> bytes"}
for key, label in LABELS.items():
    seg = prof.get(label)
    if not seg or "counters" not in seg:
        continue
    c = seg["counters"]
    fetch = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
    write = c.get("WRITE_SIZE", 0.0) * 1024.0
    kern_us = sum(v["total_us"] for k, v in seg.get("kernels", {}).items() if k.startswith(OURS))
    out[key] = {"hbm_bytes_per_call": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
                "kernel_us_under_the_tracer": round(kern_us, 1), "l2_hit": c.get("TCC_HIT_sum"), "l2_miss": c.get("TCC_MISS_sum"),
                "segment": label}
json.dump(out, open(os.path.join(ROOT, "profiles", "algos_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
