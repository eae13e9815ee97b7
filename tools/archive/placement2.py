#!/usr/bin/env python3
"""Is the sweep time a property of where the buffers landed?  One process, one graph and plan; several engines, each
with its own value stream (3.6 GB at scale 26) and x / score vectors, all kept alive; 100 sweeps each, twice."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
torch.cuda.empty_cache()
sets = []
for k in range(count):
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.init(sc, x[0])
    sets.append((eng, x, sc, err))
    pad = torch.empty(int(1.3e9), dtype=torch.uint8, device="cuda")  # shifts where the next engine's buffers land
    sets.append(pad)
def measure(s, reps=100):
    eng, x, sc, err = s
    for k in range(10):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
engines = [s for s in sets if isinstance(s, tuple)]
for rnd in range(2):
    print("round", rnd, " ".join(f"{measure(s):.3f}" for s in engines), flush=True)
