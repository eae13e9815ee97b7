#!/usr/bin/env python3
"""Block-Gauss-Seidel sweeps against the synchronous ones and the oracle's threaded path at their fixed points: which rows are
farthest from the reference, for K blocks / hub rows per block or beside block 0.  usage: gs_diag.py [scale] [layout]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from graph_amd import prelude as P
from oracle import oracle as O
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
lay = getattr(P.CsrLayout, sys.argv[2] if len(sys.argv) > 2 else "Sorted")
s, d = O.rmat_edges(scale, seed=42); n = 1 << scale
g = P.DirectedCsrGraph(P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, lay), P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, lay), lay)
ioff, itgt, _ = g.csr_inc.host(); od = O.out_degrees_from(n, s); deg = np.diff(ioff.astype(np.int64))
cfg = P.PageRankConfig(300, 1e-10, 0.85)
refs = [O.page_rank_chunked(ioff, itgt, od, 300, 1e-10, 0.85) for _ in range(3)]
seq = O.page_rank_seq(ioff, itgt, od, 300, 1e-10, 0.85)
exact = O.page_rank_f64(ioff, itgt, od)[0]
ref = refs[0][0].astype(np.float64)
print("reference runs against each other:", [float((np.abs(r[0] - ref) / ref).max()) for r in refs[1:]], "iterations", [r[1] for r in refs],
      "; sequential order vs threaded:", float((np.abs(seq[0] - ref) / ref).max()), seq[1])
def show(name, got, it):
    rel = np.abs(got.astype(np.float64) - ref) / ref
    w = np.argsort(-rel)[:4]
    print(f"{name}: {it} sweeps, max rel vs threaded reference {rel.max():.2e} (vs sequential {float((np.abs(got - seq[0]) / seq[0]).max()):.2e}, vs f64-exact "
          f"{float((np.abs(got - exact) / exact).max()):.2e}); worst rows (in-degree, rel): {[(int(deg[i]), float(rel[i])) for i in w]}; rows over 1e-5: {int((rel > 1e-5).sum())}")
jac, itj, _ = P.page_rank(g, cfg, P.PageRankMode.JacobiPB); show("synchronous", jac, itj)
for K, hubs in (("16", "1"), ("16", "0"), ("2", "1"), ("4", "1"), ("8", "1"), ("64", "1")):
    os.environ["GM_PR_BLOCK_GS"] = K; os.environ["GM_PR_GS_HUBS"] = hubs; os.environ["GM_PB_NOCACHE"] = "1"
    got, it, _ = P.page_rank(g, cfg, P.PageRankMode.BlockGS); show(f"block-GS K={K} hubs_by_block={hubs}", got, it)
# the same on PARKED engines (no GM_PB_NOCACHE): block-GS as the first call of a handle, and behind synchronous calls
for k in ("GM_PR_BLOCK_GS", "GM_PR_GS_HUBS", "GM_PB_NOCACHE"):
    os.environ.pop(k, None)
def fresh():
    return P.DirectedCsrGraph(P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, lay), P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, lay), lay)
g1 = fresh()
got, it, _ = P.page_rank(g1, cfg, P.PageRankMode.BlockGS); show("parked engine, block-GS first", got, it)
got, it, _ = P.page_rank(g1, cfg, P.PageRankMode.BlockGS); show("parked engine, block-GS again", got, it)
g2 = fresh()
j2, itj2, _ = P.page_rank(g2, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB); show("parked engine, synchronous first (200, 1e-10)", j2, itj2)
got, it, _ = P.page_rank(g2, cfg, P.PageRankMode.BlockGS); show("parked engine, block-GS behind it", got, it)
j3, itj3, _ = P.page_rank(g2, cfg, P.PageRankMode.JacobiPB); show("parked engine, synchronous again", j3, itj3)
print("synchronous runs equal:", bool(np.array_equal(jac, j2)), bool(np.array_equal(j2, j3)), itj, itj2, itj3)
for _ in range(3):
    g3 = fresh()
    j4, itj4, e4 = P.page_rank(g3, cfg, P.PageRankMode.JacobiPB)
    print("   another fresh handle, synchronous:", itj4, e4, bool(np.array_equal(j4, jac)))
