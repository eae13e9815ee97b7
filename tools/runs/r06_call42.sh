#!/bin/bash
# round 6, GPU call 42: which row of the scale-16 + fans graph on n = 297676 is 4.3e-5 away from the reference
export TMPDIR=/tmp
timeout 600 python - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np
from graph_amd import prelude as P
from oracle import oracle as O
scale, fans = 16, [1000, 2687, 4095]
s, d = O.rmat_edges(scale, seed=42); n0 = 1 << scale
centres = n0 + np.arange(len(fans)); at, ls, ld = n0 + len(fans), [], []
for c, k in zip(centres, fans):
    ls.append(np.arange(at, at + k, dtype=np.uint32)); ld.append(np.full(k, c, np.uint32)); at += k
s = np.concatenate([s] + ls + [centres.astype(np.uint32)]); d = np.concatenate([d] + ld + [np.zeros(len(fans), np.uint32)])
for n in (int(at), 297676, 131072 + 7789):
    ioff, itgt = O.csr_build(n, s, d, O.INCOMING, O.SORTED); od = O.out_degrees_from(n, s)
    ref, _, _ = O.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
    out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted); inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
    got = np.asarray(P.page_rank(P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted), P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)[0])
    rel = np.abs(got.astype(np.float64) - ref) / ref
    deg = np.diff(ioff.astype(np.int64))
    order = np.argsort(rel)[::-1][:6]
    for w in order:
        src = itgt[ioff[w]:ioff[w + 1]]
        leaf = int((deg[src] == 0).sum())
        print(f"n {n}: row {w} rel {rel[w]:.2e} in-degree {int(deg[w])}, {leaf} of its sources have no in-edges; out-degrees of those: {np.unique(od[src][deg[src] == 0])[:8]}")
PY
