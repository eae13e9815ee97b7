#!/usr/bin/env python3
"""case 12 of tests/test_gpu_hub_adversarial.py::test_random_small_graphs_through_the_hub_path, row by row"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_PB_HUB_DEG"] = "64"; os.environ["GM_PB_NOCACHE"] = "1"
import numpy as np
from graph_amd import prelude as P
from oracle import oracle as O
rng = np.random.default_rng(20260925)
for case in range(13):
    n = int(rng.integers(200, 6000)); m = int(rng.integers(2000, 60000))
    s = rng.integers(0, n, m).astype(np.uint32); d = rng.integers(0, n, m).astype(np.uint32)
    for _ in range(int(rng.integers(1, 6))):
        a, b = sorted(rng.integers(0, m, 2)); d[a:b] = rng.integers(0, n)
inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
g = P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted)
ioff, itgt, _ = inc.host(); od = np.bincount(s, minlength=n).astype(np.uint32); deg = np.diff(ioff.astype(np.int64))
for sweeps in (1, 2, 4):
    scores = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    with np.errstate(divide="ignore"):
        x = (scores / od.astype(np.float32)).astype(np.float32)
    for _ in range(sweeps):
        x, _ = O.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, scores, np.where(np.isfinite(x), x, np.float32(0)))
    for replay in ("0", "1", "2"):
        os.environ["GM_PB_HUB_REPLAY"] = replay
        got, it, _ = P.page_rank(g, P.PageRankConfig(sweeps, 0.0, 0.85), P.PageRankMode.JacobiPB)
        rel = np.abs(got.astype(np.float64) - scores) / scores
        w = np.argsort(rel)[-3:][::-1]
        print(f"n {n} m {m} sweeps {sweeps} replay {replay}: max rel {rel.max():.2e}; worst rows " +
              ", ".join(f"{int(r)} (in-degree {int(deg[r])}, {rel[r]:.1e})" for r in w), flush=True)
print("rows with >= 64 in-edges:", int((deg >= 64).sum()), "largest in-degrees:", np.sort(deg)[-6:])
