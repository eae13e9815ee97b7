#!/usr/bin/env python3
"""Does the sweep get faster with the number of physical REGIONS the value stream is spread over?  Engines whose stream
is mapped (HIP virtual-memory API) from every stride-th of a sequence of 256 MiB physical pieces created back to back:
stride 1 = 3.6 GB in one stretch of physical memory, stride 8 = one piece every 2 GiB, ... (GM_PB_VALS_POOL).
usage: placement8.py [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
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
free_gib = torch.cuda.mem_get_info()[0] / 2**30
print(f"free {free_gib:.1f} GiB", flush=True)
variants = [("malloc", None)]
for mib in (256, 64):
    pieces = -(-3700 // mib)
    for span_gib in (0, 8, 16, 24, 48, 72, 96, 144, 200):
        stride = max(1, int(span_gib * 1024 / mib / pieces))
        pool = (pieces - 1) * stride + 1
        if pool * mib / 1024 > free_gib - 12:
            continue
        variants.append((f"{mib}MiB-stride{stride}-span{pool * mib / 1024:.0f}GiB", f"{mib},{pool},0,{stride}"))
for name, spec in variants + [("malloc-again", None)]:
    os.environ.pop("GM_PB_VALS_POOL", None)
    if spec:
        os.environ["GM_PB_VALS_POOL"] = spec
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    eng.init(sc, x[0])
    sweep = lambda k: eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    timed(sweep, 6)
    t = timed(sweep, 30)
    tb = timed(lambda k: eng.sweep_bin(x[0], 0, n), 12)
    print(f"{name:36s} sweep {t:.3f}  bin {tb:.3f}  rest {t - tb:.3f}", flush=True)
    del eng
