#!/usr/bin/env python3
"""Which stretches of physical memory go together?  P physical pieces of 1 GiB created back to back; the value stream
(3.4 GiB) mapped from two pieces around index i and two around index j (GM_PB_VALS_PICK); bin-kernel time for a fixed i
and every j: where the time drops, i and j lie in stretches that use different DRAM resources.
usage: placement10.py [scale] [pieces]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_ARENA"] = "0"; os.environ["GM_PB_DRAWS"] = "0"
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
P = int(sys.argv[2]) if len(sys.argv) > 2 else 160
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
def bin_ms(picks):
    os.environ["GM_PB_VALS_PICK"] = f"1024,{P}," + ",".join(str(p) for p in picks)
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    eng.init(sc, x[0])
    timed(lambda k: eng.sweep_bin(x[0], 0, n), 3)
    t = timed(lambda k: eng.sweep_bin(x[0], 0, n), 10)
    del eng
    return t
print(f"free {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB, {P} pieces of 1 GiB", flush=True)
for i in (0, 40):
    print(f"i = {i}: consecutive pieces {i}..{i + 3}: {bin_ms([i, i + 1, i + 2, i + 3]):.3f} ms", flush=True)
    row = []
    for j in range(0, P - 1, 2):
        if abs(j - i) < 2:
            row.append("  -  ")
            continue
        row.append(f"{bin_ms([i, j, i + 1, j + 1]):.3f}")
        if len(row) == 10:
            print(f"  j = {j - 18:3d}..{j:3d}: " + " ".join(row), flush=True)
            row = []
    if row:
        print("  rest: " + " ".join(row), flush=True)
