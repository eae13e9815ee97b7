#!/usr/bin/env python3
"""A few page_rank() drop-in calls with the default mode (block-Gauss-Seidel sweeps on the propagation-blocking engine) for a tracer:
usage: gs_call.py <scale> [calls=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth, prelude as P
scale = int(sys.argv[1]); calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted), synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
del src, dst
cfg = P.PageRankConfig(20, 1e-4, 0.85)
for k in range(calls):
    torch.cuda.synchronize(); t = time.perf_counter(); _, it, err = P.page_rank(g, cfg); torch.cuda.synchronize()
    print(f"scale {scale} call {k}: {1e3 * (time.perf_counter() - t):.3f} ms, {it} iterations, error {err:.3e}", flush=True)
