#!/usr/bin/env python3
"""Per-sweep kernel time over a long run (DVFS behaviour): prints the mean of every block of sweeps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
eng.init(sc, x[0]); torch.cuda.synchronize()
blk = 50
t0 = time.perf_counter()
for b in range(total // blk):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(blk):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    e1.record(); torch.cuda.synchronize()
    print(f"t={time.perf_counter()-t0:6.2f}s  ms/sweep={e0.elapsed_time(e1)/blk:.4f}", flush=True)
