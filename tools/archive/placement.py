#!/usr/bin/env python3
"""Does the sweep time depend on where the driver placed the buffers?  Re-creates the engine several
times in one process and times bin / accumulate separately."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if os.environ.get("PLACEMENT_REBUILD_PLAN"): os.environ["GM_PB_NOCACHE"] = "1"
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
torch.cuda.empty_cache()
junk = []
offs = [int(v) for v in os.environ.get("PLACEMENT_OFFS", "0,0,0,0,0").split(",")]
for trial, off in enumerate(offs):
    os.environ["GM_PB_VALS_OFF"] = str(off)
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.init(sc, x[0])
    for k in range(10):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * 20)]
    for k in range(20):
        ev[3 * k].record(); eng.sweep_tiles(x[k % 2], x[1 - k % 2], sc); ev[3 * k + 1].record()
        eng.sweep_fixup(x[1 - k % 2], sc, err); ev[3 * k + 2].record()
    torch.cuda.synchronize()
    main = sum(ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(20)) / 20
    print(f"trial {trial} off={off}: bin+accum {main:.4f} ms", flush=True)
    del eng
    if os.environ.get("PLACEMENT_JUNK") and trial % 2 == 0:
        junk.append(torch.empty((trial + 1) * 300_000_000, dtype=torch.uint8, device="cuda"))  # shift later placements
