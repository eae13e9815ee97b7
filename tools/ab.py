#!/usr/bin/env python3
"""A/B of propagation-blocking plan knobs inside ONE process: the sweep time depends on where the
driver places the big buffers (+-10 % between processes), so configurations are alternated and each
one rebuilt several times.  Usage: tools/ab.py <scale> <rounds> "K=V,K=V" "K=V" ...  ("-" = defaults)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_PB_NOCACHE"] = "1"
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale, rounds = int(sys.argv[1]), int(sys.argv[2])
configs = sys.argv[3:] or ["-"]
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
torch.cuda.empty_cache()
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda")
KNOBS = ["GM_PB_SLOG", "GM_PB_WGS", "GM_PB_RB", "GM_PB_HOT", "GM_PB_CHUNK", "GM_PB_XCD", "GM_PB_SPLIT", "GM_PB_ORDER"]
res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        for k in KNOBS:
            os.environ.pop(k, None)
        if c != "-":
            for kv in c.split(","):
                k, v = kv.split("=")
                os.environ[k] = v
        eng = PageRankEngine(csr.handle, n, 0, od, 0.85, engine=2)
        eng.init(sc, x[0])
        for k in range(10):
            eng.sweep_tiles(x[k % 2], x[1 - k % 2], sc)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for k in range(20):
            eng.sweep_tiles(x[k % 2], x[1 - k % 2], sc)
        ev[1].record()
        torch.cuda.synchronize()
        res[c].append(ev[0].elapsed_time(ev[1]) / 20)
        del eng
for c in configs:
    v = res[c]
    print(f"{c:40s} min {min(v):.4f}  median {statistics.median(v):.4f}  all {' '.join('%.3f' % t for t in v)}", flush=True)
del csr
