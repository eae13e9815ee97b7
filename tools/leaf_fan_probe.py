#!/usr/bin/env python3
"""Rows BELOW the hub threshold whose in-neighbours all carry the same out_score — a node with k leaf followers (in-degree 0,
out-degree 1: each at (1 - d) / n exactly) — against the reference's left-to-right f32 sum.  The reference's sum of k EQUAL terms
drifts systematically (each add rounds the same way); an exactly rounded sum does not follow it.
usage: leaf_fan_probe.py [scale=18] [fans: k,k,...]      (propagation-blocking engine, GM_PB_HUB_DEG as set in the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from graph_amd import prelude as P
from oracle import oracle as O  # (a tool, like tests/: the checker)

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
fans = [int(k) for k in (sys.argv[2] if len(sys.argv) > 2 else "300,700,1000,1500,2000,2687,3000,3500,4000,4095").split(",")]
n0 = 1 << scale
s, d = O.rmat_edges(scale, seed=42)
# append the fans: centre c_i = n0 + i, its leaves behind all centres
centres = n0 + np.arange(len(fans))
first_leaf = n0 + len(fans)
ls, ld = [], []
at = first_leaf
for c, k in zip(centres, fans):
    ls.append(np.arange(at, at + k, dtype=np.uint32)); ld.append(np.full(k, c, np.uint32)); at += k
# every centre points back into the graph (so it is no sink) — one edge to node 0
s = np.concatenate([s] + ls + [centres.astype(np.uint32)]); d = np.concatenate([d] + ld + [np.zeros(len(fans), np.uint32)])
n = int(at)
out = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Outgoing, P.CsrLayout.Sorted)
inc = P.DeviceCsr.from_edges(n, s, d, None, P.Direction.Incoming, P.CsrLayout.Sorted)
g = P.DirectedCsrGraph(out, inc, P.CsrLayout.Sorted)
ioff, itgt = O.csr_build(n, s, d, O.INCOMING, O.SORTED)
od = O.out_degrees_from(n, s)
ref, it_r, _ = O.page_rank_chunked(ioff, itgt, od, 200, 1e-10, 0.85)
got, it_g, _ = P.page_rank(g, P.PageRankConfig(200, 1e-10, 0.85), P.PageRankMode.JacobiPB)
got = np.asarray(got)
rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-300)
print(f"scale {scale} + {len(fans)} leaf fans, GM_PB_HUB_DEG={os.environ.get('GM_PB_HUB_DEG', '(default)')}: reference {it_r} sweeps, device {it_g}; "
      f"max rel over the RMAT rows {rel[:n0].max():.2e}")
for c, k in zip(centres, fans):
    print(f"   fan of {k:5d} equal terms: reference {ref[c]:.9e}  device {got[c]:.9e}  rel {rel[c]:.2e}" + ("   > 1e-5" if rel[c] > 1e-5 else ""))
