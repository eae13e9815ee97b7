#!/usr/bin/env python3
"""Wall time of the page_rank() drop-in call with block-Gauss-Seidel and with synchronous sweeps (host result buffers, plan cached):
usage: gs_time.py <scale> [iterations tolerance]   (environment knobs apply: GM_PR_BLOCK_GS, GM_PR_GS_HUBS, GM_PB_HUB_FORK)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth, prelude as P
scale = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20; tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted), synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
del src, dst
cfg = P.PageRankConfig(iters, tol, 0.85)
out = {}
for name, mode in (("block_gs", P.PageRankMode.BlockGS), ("synchronous", P.PageRankMode.JacobiPB)):
    P.page_rank(g, cfg, mode)
    best, it, err = None, None, None
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); _, it, err = P.page_rank(g, cfg, mode); torch.cuda.synchronize()
        dt = time.perf_counter() - t; best = dt if best is None else min(best, dt)
    out[name] = (round(best * 1e3, 3), it, float(f"{err:.3e}"))
print(f"scale {scale} ({iters}, {tol}) env {dict((k, v) for k, v in os.environ.items() if k.startswith('GM_'))}: {out}", flush=True)
