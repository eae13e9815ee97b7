#!/usr/bin/env python3
"""RMAT edges on a node set that is NOT a power of two (isolated nodes appended: n changes, and with it the bits of (1 - d) / n — the
score every node without in-edges carries): the propagation-blocking engine against the reference's threaded path on every row.
usage: pad_n_probe.py <scale> <n> [n ...]     (GM_PB_HUB_DEG as set in the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from graph_amd import prelude as P
from oracle import oracle as O  # (a tool, like tests/: the checker)

scale = int(sys.argv[1])
s, d = O.rmat_edges(scale, seed=42)
for n in [int(a) for a in sys.argv[2:]]:
    assert n >= 1 << scale
    out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
    inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    ioff, itgt = O.csr_build(n, s, d, O.INCOMING, O.SORTED)
    od = O.out_degrees_from(n, s)
    ref, it_r, _ = O.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
    got, it_g, _ = P.page_rank(P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted), P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)
    got = np.asarray(got)
    rel = np.abs(got.astype(np.float64) - ref) / ref
    deg = np.diff(ioff.astype(np.int64))
    w = int(rel.argmax())
    thr = int(os.environ.get("GM_PB_HUB_DEG", "4096"))
    lo = rel[deg < thr] if thr else rel
    print(f"scale {scale} edges on n = {n} (hub threshold {thr}): max rel {rel.max():.2e} (row {w}, in-degree {int(deg[w])}), rows over 1e-5: {int((rel > 1e-5).sum())}, "
          f"rows at or above the threshold: {rel[deg >= thr].max() if thr and (deg >= thr).any() else 0:.2e}", flush=True)
