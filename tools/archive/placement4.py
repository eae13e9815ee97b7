#!/usr/bin/env python3
"""Same pages, other offsets: one engine whose value stream has 256 MiB of slack behind it; the stream is moved inside
the allocation (GM_PB_VALS_OFFSET, KiB) between measurements.  If the sweep time follows the offset, the placement
effect is an address-interleaving one; if not, it belongs to the pages the allocation got."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_PB_VALS_SLACK"] = "256"
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
eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
def measure(reps=60):
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
for off_kib in (0, 4, 64, 256, 1024, 2048, 4096, 65536, 131072, 0, 2048):
    os.environ["GM_PB_VALS_OFFSET"] = str(off_kib)
    print(f"offset {off_kib:7d} KiB: {measure():.3f} ms", flush=True)
