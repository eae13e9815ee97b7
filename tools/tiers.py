#!/usr/bin/env python3
"""Sweep time against the number of hot-source tiers (GM_PB_TIERS), private plans on one resident graph.
usage: tiers.py [scale] [tiers ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
tiers = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 13, 20, 32]
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
def timed(fn, reps):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
os.environ["GM_PB_NOCACHE"] = "1"
ref = None
for t in tiers:
    os.environ["GM_PB_TIERS"] = str(t)
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    info = eng.plan_info()
    eng.init(sc, x[0])
    for k in range(4):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    torch.cuda.synchronize()
    res = sc.clone()
    same = "first" if ref is None else ("bit-identical" if torch.equal(res, ref) else f"DIFFERENT max {float((res - ref).abs().max()):.3e}")
    if ref is None:
        ref = res
    sweep = lambda k: eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    timed(sweep, 6)
    tt = timed(sweep, 40)
    tb = timed(lambda k: eng.sweep_bin(x[0], 0, n), 16)
    print(f"tiers {t:3d} (got {info['hot_tiers']}, {info['hot_sources']} sources, {info['hot_edges']} hot / {info['value_entries']} stream "
          f"entries, plan {info['plan_build_us'] / 1e3:.0f} ms)  sweep {tt:.3f}  bin {tb:.3f}  rest {tt - tb:.3f}   scores after 4 sweeps: {same}",
          flush=True)
    del eng
