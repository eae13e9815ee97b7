#!/usr/bin/env python3
"""Longer write runs against the slow level: private plans with 16384-source tiles (GM_PB_SLOG=14, the default) and 32768-source tiles
(15: (tile, bin) segments twice as long), alternating in one process; placement draws and sweep time of each.
usage: placement13.py [scale=26] [pairs=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_PB_NOCACHE"] = "1"
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
for i in range(pairs):
    for slog in ("14", "15"):
        os.environ["GM_PB_SLOG"] = slog
        eng = PageRankEngine(csr.handle, n, 0, od, 0.85, engine=2)
        eng.init(sc, x[0])
        for k in range(6):
            eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(20):
            eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        e1.record(); torch.cuda.synchronize()
        info = eng.plan_info()
        print(f"tiles of 2^{slog} sources: segments {info['segments']}, draws {info['draws_timed']} best {info['draw_best_us']} us, "
              f"sweep {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
        del eng
