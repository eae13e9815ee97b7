#!/bin/bash
# round 6, GPU call 52: the rule for rows of constant terms on the other paths — Unsorted lists (hub rows in CSR order), the default call
# (block-Gauss-Seidel sweeps), Unsorted over three virtual ranks — on the fan graph at n = 297,676, against the oracle to convergence
export TMPDIR=/tmp
timeout 900 python - <<'PY'
import sys; sys.path.insert(0, '.')
import os
import numpy as np
from graph_amd import prelude as P
from oracle import oracle as O
os.environ["GM_MULTI_ENGINE"] = "pb"
scale, fans = 16, [300, 511, 1000, 2687, 4095]
s, d = O.rmat_edges(scale, seed=42); n0 = 1 << scale
centres = n0 + np.arange(len(fans)); at, ls, ld = n0 + len(fans), [], []
for c, k in zip(centres, fans):
    ls.append(np.arange(at, at + k, dtype=np.uint32)); ld.append(np.full(k, c, np.uint32)); at += k
s = np.concatenate([s] + ls + [centres.astype(np.uint32)]); d = np.concatenate([d] + ld + [np.zeros(len(fans), np.uint32)])
n = 297676
rng = np.random.default_rng(5); perm = rng.permutation(s.size); su, du = s[perm], d[perm]   # an arrival order for the Unsorted layout
for name, layout, olay, (ss, dd) in (("Sorted", P.CsrLayout.Sorted, O.SORTED, (s, d)), ("Unsorted", P.CsrLayout.Unsorted, O.UNSORTED, (su, du))):
    out = P.DeviceCsr.from_edges(n, ss, dd, None, P.Direction.Outgoing, layout); inc = P.DeviceCsr.from_edges(n, ss, dd, None, P.Direction.Incoming, layout)
    g = P.DirectedCsrGraph(out, inc, layout)
    ioff, itgt = inc.host()[0], inc.host()[1]          # the very lists the device holds (Unsorted: the device build's order)
    od = O.out_degrees_from(n, ss)
    ref, it_r, _ = O.page_rank_chunked(np.asarray(ioff), np.asarray(itgt), od, 200, 1e-10, 0.85)
    for mode_name, run in (("synchronous PB", lambda: P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)),
                           ("default call (block-GS)", lambda: P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85))),
                           ("three virtual ranks", lambda: P.page_rank_multi(g, P.PageRankConfig(200, 1e-10, 0.85), devices=[0, 0, 0]))):
        got, it, err = run(); got = np.asarray(got)
        rel = np.abs(got.astype(np.float64) - ref) / ref
        print(f"{name:8s} {mode_name:24s}: {it:3d} sweeps, max rel on every row {rel.max():.2e}, on the fans' rows {rel[centres].max():.2e}, rows over 1e-5: {int((rel > 1e-5).sum())}", flush=True)
PY
