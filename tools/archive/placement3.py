#!/usr/bin/env python3
"""Which buffer's placement decides the sweep time: the engine's value stream or the x / score vectors?
One process: 4 engines x 4 vector sets, every combination timed (100 sweeps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = 26
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
torch.cuda.empty_cache()
engines, vecs, pads = [], [], []
for k in range(4):
    engines.append(PageRankEngine(csr.handle, n, 0, od, 0.85))
    pads.append(torch.empty(int(0.9e9) + k * 12345678, dtype=torch.uint8, device="cuda"))
    vecs.append(([torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")], torch.zeros(n, device="cuda")))
err = torch.zeros(1, dtype=torch.float64, device="cuda")
def measure(eng, x, sc, reps=60):
    eng.init(sc, x[0])
    for k in range(6):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("rows: engines (value stream), columns: vector sets")
for e in engines:
    print(" ".join(f"{measure(e, x, sc):.3f}" for x, sc in vecs), flush=True)
