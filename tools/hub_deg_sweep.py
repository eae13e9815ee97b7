#!/usr/bin/env python3
"""What the hub threshold costs and buys: rows with at least GM_PB_HUB_DEG in-edges are summed in the
reference's left-to-right f32 order (crates/algos/src/page_rank.rs:143-146), the others exactly rounded.  One graph, one
oracle run (orc_page_rank_chunked to its fixed point), then for every threshold a PRIVATE plan (GM_PB_NOCACHE): ms per sweep
over 20 timed sweeps, and the engine run on to its fixed point compared with the oracle on every row.

    python tools/hub_deg_sweep.py --scale 26 --degs 4096,2048,1024      one JSON object on stdout
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--degs", default="4096,2048,1024")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    import torch

    from graph_amd import synth
    from graph_amd._lib import check, lib, vp
    from graph_amd.engine import PageRankEngine
    from graph_amd.prelude import CsrLayout, Direction
    from oracle import oracle as O  # the checker

    sc, n = args.scale, 1 << args.scale
    src, dst = synth.rmat_edges(sc, 42)
    m = int(src.numel())
    out_deg = torch.bincount(src, minlength=n).to(torch.int32)
    in_csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
    del src, dst
    torch.cuda.empty_cache()
    off_h = np.empty(n + 1, np.uint32)
    tgt_h = np.empty(m, np.uint32)
    check(lib().gm_csr_download(in_csr.handle, off_h.ctypes.data_as(vp), tgt_h.ctypes.data_as(vp), None))
    od_h = out_deg.cpu().numpy().astype(np.uint32)
    cores = O.effective_cores()
    t = time.perf_counter()
    ref, it_ref, _ = O.page_rank_chunked(off_h, tgt_h, od_h, 200, 1e-10, 0.85, cores)
    t_ref = time.perf_counter() - t
    ref = ref.astype(np.float64)
    deg_h = np.diff(off_h.astype(np.int64))
    del tgt_h
    rows = []
    os.environ["GM_PB_NOCACHE"] = "1"
    for d in [int(x) for x in args.degs.split(",")]:
        os.environ["GM_PB_HUB_DEG"] = str(d)
        eng = PageRankEngine(in_csr.handle, n, 0, out_deg, 0.85, x_len=n, engine=2)
        scores = torch.zeros(n, dtype=torch.float32, device="cuda")
        err = torch.zeros(1, dtype=torch.float64, device="cuda")
        x = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(2)]
        eng.init(scores, x[0])
        for it in range(40):  # clock ramp + warm-up
            eng.sweep(x[it % 2], x[1 - it % 2], scores, err)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for it in range(args.steps):
            eng.sweep(x[it % 2], x[1 - it % 2], scores, err)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / args.steps
        eng.init(scores, x[0])
        sweeps = 0
        for it in range(200):
            eng.sweep(x[it % 2], x[1 - it % 2], scores, err)
            sweeps += 1
            if float(err.item()) < 1e-10:
                break
        got = scores.cpu().numpy().astype(np.float64)
        rel = np.abs(got - ref) / ref
        hub = deg_h >= d
        info = eng.plan_info()
        rows.append({"hub_deg": d, "ms_per_sweep": round(ms, 4), "max_rel_vs_reference": float(rel.max()),
                     "rows_over_1e-5": int((rel > 1e-5).sum()), "max_rel_hub_rows": float(rel[hub].max()) if hub.any() else None,
                     "max_rel_ordinary_rows": float(rel[~hub].max()), "worst_ordinary_in_degree": int(deg_h[~hub][np.argmax(rel[~hub])]),
                     "hub_rows": info.get("hub_rows"), "hub_edges": info.get("hub_edges"), "hub_groups": info.get("hub_groups"),
                     "long_rows": info.get("long_rows"), "value_entries": info.get("value_entries"),
                     "hot_edges": info.get("hot_edges"), "hub_hot_edges": info.get("hub_hot_edges"),
                     "plan_build_ms": round(info.get("plan_build_us", 0) / 1e3, 1), "device_sweeps_to_fixed_point": sweeps,
                     "level_draw_best_us": info.get("draw_best_us")})
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
        del eng, scores, err, x
        torch.cuda.empty_cache()
    print(json.dumps({"tool": "hub_deg_sweep", "scale": sc, "edges": m, "reference": {"impl": "orc_page_rank_chunked", "threads": cores,
                                                                                       "iterations": int(it_ref), "seconds": round(t_ref, 2)},
                      "rows": rows}), flush=True)


if __name__ == "__main__":
    main()
