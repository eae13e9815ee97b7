#!/usr/bin/env python3
"""gm_page_rank_multi on ONE GPU with virtual ranks (a device listed several times: copies instead of RCCL) — what the C-ABI
multi-GPU entry does per call, with GM_LOG's phase lines: first call (partition, slices, engines built), second call on the
resident state, K = 2 regions against K = 1.  The times say nothing about real multi-GPU runs (every rank shares one GPU).
usage: multi_virtual.py [scale=22] [ranks=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_LOG"] = "1"
import numpy as np
from graph_amd import synth
import graph_amd.prelude as P

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
g = P.DirectedCsrGraph(synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted),
                       synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted), P.CsrLayout.Sorted)
del src, dst
cfg = P.PageRankConfig(20, 0.0, 0.85)
t = time.perf_counter(); one, it1, e1 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB); t_one = time.perf_counter() - t
t = time.perf_counter(); one, it1, e1 = P.page_rank(g, cfg, P.PageRankMode.JacobiPB); t_one2 = time.perf_counter() - t
print(f"== single engine: {t_one * 1e3:.1f} ms first, {t_one2 * 1e3:.1f} ms again", file=sys.stderr, flush=True)
for parts in ("2", "1"):
    os.environ["GM_MULTI_PARTS"] = parts
    for call in ("first", "again"):
        print(f"== {ranks} virtual ranks, K = {parts}, {call} call", file=sys.stderr, flush=True)
        t = time.perf_counter(); got, it, e = P.page_rank_multi(g, cfg, devices=[0] * ranks); dt = time.perf_counter() - t
        rel = float((np.abs(got.astype(np.float64) - one) / one).max())
        print(f"== {dt * 1e3:.1f} ms, {it} sweeps, error {e:.6e} (single engine {e1:.6e}), max rel vs single engine {rel:.2e}", file=sys.stderr, flush=True)
