#!/usr/bin/env python3
"""EMULATED strong scaling of the PageRank sweep: every rank's slice of an N-way partition timed on ONE device
(bench.py --emulate-parts N --emulate-rank r: the exchange replaced by a local copy), plus a link model for the exchange.
Nothing here ran on more than one GPU.  usage: partition_emulated.py [--scale 26] [--parts 1,2,4,8] > table.json"""
import argparse, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=26)
ap.add_argument("--parts", default="1,2,4,8")
ap.add_argument("--link-gbs", type=float, default=153.0, help="xGMI per-link peak, /opt/skills/guides/MI355X_MICROARCH.md")
ap.add_argument("--link-eff", type=float, default=0.75, help="assumed achievable fraction of the link peak")
args = ap.parse_args()


def run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scale", str(args.scale), "--cpu-sweeps", "0"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
    return json.loads(out[-1])


table = []
for n_parts in [int(p) for p in args.parts.split(",")]:
    ranks = []
    for r in range(n_parts):
        d = run(["--emulate-parts", str(n_parts), "--emulate-rank", str(r)] if n_parts > 1 else [])
        h = d["config"]["hub_rows_in_reference_order"] or {}
        m = re.search(r"all-gather of (\d+) B/rank/sweep", d["config"]["partition"])
        ranks.append({"rank": r, "sweep_ms": d["ms_per_step"], "edges": d["roofline"]["edges_per_launch"],
                      "rows": d["roofline"]["rows_per_launch"], "exchange_bytes_sent": int(m.group(1)) if m else 0,
                      "long_rows": h.get("long_rows"), "long_row_terms": h.get("long_row_terms")})
        edges_total, device = d["config"]["edges"], d["config"]["device"]
        print(f"parts {n_parts} rank {r}: {d['ms_per_step']} ms", file=sys.stderr, flush=True)
    slow = max(x["sweep_ms"] for x in ranks)
    # every pair of the N <= 8 GPUs of a node has its own xGMI link: in a direct all-gather a rank receives the N - 1 slices
    # over N - 1 links at once, so the exchange takes (one slice) / (link rate)
    slice_bytes = max(x["exchange_bytes_sent"] for x in ranks)
    ex_ms = slice_bytes / (args.link_gbs * args.link_eff * 1e9) * 1e3 if n_parts > 1 else 0.0
    lo, hi = max(slow, ex_ms), slow + ex_ms / 2.0  # K = 2 regions: between "all of it hidden" and "the second region's half exposed"
    table.append({"gpus": n_parts, "ranks": ranks, "slowest_rank_ms": slow, "fastest_rank_ms": min(x["sweep_ms"] for x in ranks),
                  "exchange_ms_model": round(ex_ms, 4), "projected_sweep_ms": [round(lo, 4), round(hi, 4)],
                  "projected_gteps": [round(edges_total / hi / 1e6, 2), round(edges_total / lo / 1e6, 2)]})
base = table[0]["projected_sweep_ms"][0] if table and table[0]["gpus"] == 1 else None
for t in table:
    if base:
        t["projected_speedup"] = [round(base / t["projected_sweep_ms"][1], 2), round(base / t["projected_sweep_ms"][0], 2)]
print(json.dumps({
    "label": "EMULATED: each rank's slice timed alone on one MI355X, the exchange replaced by a local copy; exchange time from a link "
             "model, not measured.  No multi-GPU run stands behind these numbers.",
    "link_model": f"direct all-gather, one xGMI link per pair ({args.link_gbs} GB/s peak per link, {args.link_eff:.0%} assumed "
                  f"achievable); projected sweep = [max(slowest rank, exchange), slowest rank + exchange / 2] for K = 2 regions",
    "scale": args.scale, "device": device, "table": table}, indent=1))
