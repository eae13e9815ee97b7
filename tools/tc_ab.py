#!/usr/bin/env python3
"""Triangle count at RMAT scale S under several environments, ONE graph, alternating: usage tc_ab.py [scale] "A=1 B=2" "C=3" ..."""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from graph_amd import prelude as P
from graph_amd import synth

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
variants = sys.argv[2:] or [""]
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
ug = P.UndirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Undirected, P.CsrLayout.Deduplicated), P.CsrLayout.Deduplicated)
del src, dst
P.relabel_graph(ug)
first = P.global_triangle_count(ug)
best = {v: 1e9 for v in variants}
for rep in range(4):
    for v in variants:
        kv = dict(x.split("=") for x in v.split()) if v else {}
        os.environ.update(kv)
        torch.cuda.synchronize()
        t = time.perf_counter()
        tri = P.global_triangle_count(ug)
        dt = (time.perf_counter() - t) * 1e3
        for k in kv:
            del os.environ[k]
        assert tri == first, (v, tri, first)
        best[v] = min(best[v], dt)
for v in variants:
    print(f"scale {scale} [{v or 'default'}]: best of 4 {best[v]:.2f} ms, {first} triangles")
