#!/usr/bin/env python3
"""pb_accum_kernel with 2 / 4 / 6 register groups of the value stream in flight (GM_PB_ACC_DEPTH, read at every launch): sweep time,
alternating inside one process on one engine.  usage: ab_depth.py [scale=26]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
# the variants live in the MEASUREMENT library only (make -C graph_amd/csrc measure: the sources with -DGM_MEASURE)
import subprocess
subprocess.check_call(["make", "-C", os.path.join(ROOT, "graph_amd", "csrc"), "measure", "-j8"], stdout=subprocess.DEVNULL)
os.environ["GRAPH_MI355X_LIB"] = os.path.join(ROOT, "graph_amd", "libgraph_mi355x_measure.so")
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
eng = PageRankEngine(csr.handle, n, 0, od, 0.85, engine=2)
eng.init(sc, x[0])
def timed(reps):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for rnd in range(3):
    for depth in ("2", "4", "6"):
        os.environ["GM_PB_ACC_DEPTH"] = depth
        eng.init(sc, x[0])
        timed(6)
        t = timed(40)
        eng.init(sc, x[0])
        for k in range(4):
            eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        got = sc.clone()
        ref = got if ref is None else ref
        print(f"round {rnd} depth {depth}: sweep {t:.3f} ms   scores after 4 sweeps {'identical' if torch.equal(got, ref) else 'DIFFERENT'}", flush=True)
