#!/usr/bin/env python3
"""Sweep time against the POSITION of the value stream inside one large allocation: an engine whose stream has S GiB of
slack behind it (GM_PB_VALS_SLACK), the stream moved in steps of `step` MiB (GM_PB_VALS_OFFSET).  A fresh process's
large allocation is made of large physical blocks, so the position inside it is (piecewise) a physical position.
usage: placement7.py [scale] [slack GiB] [step MiB] [engines]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
slack_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 16
step_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 256
engines = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
torch.cuda.empty_cache()
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
os.environ["GM_PB_VALS_SLACK"] = str(slack_gib * 1024)
keep = []
for e in range(engines):
    os.environ["GM_PB_VALS_OFFSET"] = "0"
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    eng.init(sc, x[0])
    keep.append(eng)
    sweep = lambda k: eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    timed(sweep, 20)
    for off in range(0, slack_gib * 1024 + 1, step_mib):
        os.environ["GM_PB_VALS_OFFSET"] = str(off * 1024)
        timed(sweep, 3)
        t = timed(sweep, 16)
        tb = timed(lambda k: eng.sweep_bin(x[0], 0, n), 8)
        print(f"engine {e} offset {off:6d} MiB  sweep {t:.3f}  bin {tb:.3f}  rest {t - tb:.3f}", flush=True)
