#!/usr/bin/env python3
"""two hub rows x 300000 equal terms (with and without a giant last term): emulated vs sequential sum, graded first block on / off"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_PB_NOCACHE"] = "1"
import numpy as np, torch
from graph_amd import prelude as P
from graph_amd.engine import PageRankEngine
from oracle import oracle as O
def run(hubs, sources, terms):
    s = np.repeat(np.arange(hubs, hubs + sources, dtype=np.uint32), hubs); d = np.tile(np.arange(hubs, dtype=np.uint32), sources)
    n = hubs + sources
    x0 = np.full(n, np.inf, np.float32); x0[hubs:] = terms.astype(np.float32)
    scores0 = np.full(n, np.float32(1.0) / np.float32(n), np.float32)
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    ioff, itgt, _ = inc.host(); od = np.bincount(s, minlength=n).astype(np.uint32)
    eng = PageRankEngine(inc.handle, n, 0, torch.from_numpy(od.astype(np.int32)).cuda(), 0.85, engine=PageRankEngine.PB)
    sc = torch.from_numpy(scores0.copy()).cuda(); xi = torch.from_numpy(x0.copy()).cuda(); xo = torch.empty_like(xi)
    err = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.sweep(xi, xo, sc, err); torch.cuda.synchronize()
    seq = scores0.copy(); O.page_rank_jacobi_sweep(ioff, itgt, od, 0.85, seq, np.where(np.isfinite(x0), x0, np.float32(0)))
    base = (np.float32(1) - np.float32(0.85)) / np.float32(n)
    got = sc.cpu().numpy()
    # back out the row sums: score = base + 0.85 * sum
    return (got[:hubs].astype(np.float64) - base) / 0.85, (seq[:hubs].astype(np.float64) - base) / 0.85, float(terms.astype(np.float32).astype(np.float64).sum())
for graded in ("1", "0"):
    os.environ["GM_PB_HUB_GRADED"] = graded
    for hubs, sources in ((2, 300000), (1, 300000), (3, 300000)):
        for name, terms in (("equal 3e-10", np.full(sources, 3e-10)), ("giant last", np.concatenate([np.full(sources - 1, 3e-10), [1e-3]]))):
            g, q, exact = run(hubs, sources, terms)
            print(f"graded {graded} {hubs} x {sources} {name}: device sum {g[0]:.9e} sequential {q[0]:.9e} exact {exact:.9e}  rel(dev,seq) {abs(g[0]-q[0])/q[0]:.2e}", flush=True)
