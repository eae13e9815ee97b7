#!/usr/bin/env python3
"""PageRank parity at BASELINE sizes: the HIP engine against the oracle's restatement of the reference's
threaded path (orc_page_rank_chunked, crates/algos/src/page_rank.rs:113-168), both run to their fixed
point with PageRankConfig::new(200, 1e-10, 0.85), on RMAT scale S built on the device.

Reports the maximum relative difference over ALL rows, per in-degree class, the number of rows above
1e-5 and the worst rows.  One JSON line on stdout (committed under profiles/ per round).

    python tools/parity_pagerank.py --scale 26 [--mode pb|pull|auto|reforder] [--threads 0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CLASSES = [(0, 1), (1, 64), (64, 1024), (1024, 4096), (4096, 16384), (16384, 65536), (65536, 262144), (262144, 1 << 32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--mode", default="auto", choices=["auto", "pb", "pull", "reforder"])
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--tolerance", type=float, default=1e-10)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--layout", default="sorted", choices=["sorted", "unsorted"])
    args = ap.parse_args()
    import numpy as np
    import torch

    from graph_amd import prelude as P
    from graph_amd import synth
    from oracle import oracle as O  # the checker

    sc, n = args.scale, 1 << args.scale
    layout = P.CsrLayout.Sorted if args.layout == "sorted" else P.CsrLayout.Unsorted
    src, dst = synth.rmat_edges(sc, 42)
    m = int(src.numel())
    g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, layout),
                           synth.build_csr(n, src, dst, P.Direction.Incoming, layout), layout)
    del src, dst
    torch.cuda.empty_cache()
    mode = {"auto": P.PageRankMode.Auto, "pb": P.PageRankMode.JacobiPB, "pull": P.PageRankMode.JacobiPull,
            "reforder": P.PageRankMode.JacobiRefOrder}[args.mode]
    cfg = P.PageRankConfig(args.iterations, args.tolerance, 0.85)
    t = time.perf_counter()
    got, it_g, err_g = P.page_rank(g, cfg, mode)
    t_first = time.perf_counter() - t
    t = time.perf_counter()
    got2, _, _ = P.page_rank(g, cfg, mode)
    t_gpu = time.perf_counter() - t
    reproducible = bool(np.array_equal(got, got2))
    del got2

    ioff, itgt, _ = g.csr_inc.host()
    out_deg = g.csr_out.degrees().astype(np.uint32)
    threads = args.threads or O.effective_cores()
    t = time.perf_counter()
    ref, it_r, err_r = O.page_rank_chunked(ioff, itgt, out_deg, args.iterations, args.tolerance, 0.85, threads)
    t_cpu = time.perf_counter() - t

    deg = np.diff(ioff.astype(np.int64))
    rel = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / ref.astype(np.float64)
    worst = np.argsort(rel)[-8:][::-1]
    classes = []
    for lo, hi in CLASSES:
        sel = (deg >= lo) & (deg < hi)
        k = int(sel.sum())
        if k:
            r = rel[sel]
            classes.append({"in_degree": [lo, hi], "rows": k, "edge_share": round(float(deg[sel].sum()) / max(m, 1), 5),
                            "max_rel": float(r.max()), "rms_rel": float(np.sqrt((r * r).mean())),
                            "rows_over_1e-5": int((r > 1e-5).sum())})
    out = {
        "tool": "parity_pagerank", "scale": sc, "nodes": n, "edges": m, "layout": args.layout, "mode": args.mode,
        "config": {"max_iterations": args.iterations, "tolerance": args.tolerance, "damping": 0.85},
        "device": {"iterations": it_g, "error": err_g, "first_call_s": round(t_first, 3), "next_call_s": round(t_gpu, 3),
                   "bit_reproducible": reproducible},
        "reference": {"impl": "oracle/graph_oracle.c:orc_page_rank_chunked (page_rank.rs:113-168)", "threads": threads,
                      "iterations": it_r, "error": err_r, "seconds": round(t_cpu, 2)},
        "max_rel_vs_reference": float(rel.max()), "rows_over_1e-5": int((rel > 1e-5).sum()),
        "rows_over_5e-6": int((rel > 5e-6).sum()), "max_in_degree": int(deg.max()),
        "by_in_degree": classes,
        "worst_rows": [{"row": int(r), "in_degree": int(deg[r]), "rel": float(rel[r])} for r in worst],
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
