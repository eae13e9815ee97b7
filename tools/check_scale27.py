#!/usr/bin/env python3
"""One-off: PageRank (PB engine) on RMAT scale 27 with 16384- and 32768-source tiles gives identical bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graph_amd import prelude as P, synth
scale = 27; n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
out = []
for slog in ("14", "15"):
    os.environ["GM_PB_SLOG"] = slog; os.environ["GM_PB_NOCACHE"] = "1"
    g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                           synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
    r = P.page_rank(g, P.PageRankConfig(4, 0.0, 0.85), P.PageRankMode.JacobiPB)
    out.append(r); print("slog", slog, "iters", r[1], "err", r[2], "sum", float(r[0].astype(np.float64).sum()), flush=True)
    del g; torch.cuda.empty_cache()
print("IDENTICAL" if np.array_equal(out[0][0], out[1][0]) else "DIFFERENT")
