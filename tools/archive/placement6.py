#!/usr/bin/env python3
"""Which variants of the bin kernel's work order are sensitive to the pages of the value stream?  One process, one
graph; D draws of the stream's allocation (GM_PB_VALS_SHARE=<draw>: every engine created with that id sweeps over the
SAME allocation), and per draw one engine per variant of the plan (private plans, GM_PB_NOCACHE): work-item order
(GM_PB_XCD, GM_PB_WG_GROUP), chunk size, source-tile size.  Per (draw, variant): ms per sweep, ms of the bin kernel alone.

usage: placement6.py [scale] [draws]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
torch.cuda.empty_cache()
variants = [("default", {}), ("xcd0", {"GM_PB_XCD": "0"}), ("chunk8k", {"GM_PB_CHUNK": "8192"}),
            ("chunk16k", {"GM_PB_CHUNK": "16384"}), ("chunk64k", {"GM_PB_CHUNK": "65536"}),
            ("tile-wg", {"GM_PB_CHUNK": "1048576"}), ("slog15", {"GM_PB_SLOG": "15"}),
            ("group8", {"GM_PB_WG_GROUP": "8"}), ("group16", {"GM_PB_WG_GROUP": "16"}),
            ("group32", {"GM_PB_WG_GROUP": "32"}), ("group64", {"GM_PB_WG_GROUP": "64"}),
            ("group64-c16k", {"GM_PB_WG_GROUP": "64", "GM_PB_CHUNK": "16384"})]
keys = sorted({k for _, env in variants for k in env})
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
os.environ["GM_PB_NOCACHE"] = "1"
def timed(fn, reps):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
pads = []
for d in range(1, draws + 1):
    os.environ["GM_PB_VALS_SHARE"] = str(d)
    for name, env in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
        eng.init(sc, x[0])
        sweep = lambda k: eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        timed(sweep, 6)
        t = timed(sweep, 40)
        tb = timed(lambda k: eng.sweep_bin(x[0], 0, n), 20)
        print(f"draw {d} {name:14s} sweep {t:.3f} ms   bin kernel alone {tb:.3f} ms   rest {t - tb:.3f}", flush=True)
        del eng
    pads.append(torch.empty(int(1.7e9), dtype=torch.uint8, device="cuda"))  # shifts where the next draw lands
