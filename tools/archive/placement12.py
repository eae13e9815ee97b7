#!/usr/bin/env python3
"""In a process whose arena candidates are all slow, is ANY way of obtaining the value stream fast?  One process, one plan; engines whose
value stream is: the default (arena candidates, timed), one VMM piece, VMM pieces of 1 GiB / 256 MiB, plain hipMalloc (twice).
Per engine the bin kernel alone and the sweep.  usage: placement12.py [scale=26]"""
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
M = "GM_PB_VALS_VMM"
variants = [("arena (default)", {}), ("one VMM piece", {M: "-1"}), ("VMM 1 GiB pieces", {M: "1024"}), ("VMM 256 MiB pieces", {M: "256"}),
            ("hipMalloc a", {"GM_PB_DRAWS": "0"}), ("hipMalloc b", {"GM_PB_DRAWS": "0"}), ("arena again", {})]
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
engines = []
for name, env in variants:
    for k in (M, "GM_PB_DRAWS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85, engine=2)
    eng.init(sc, x[0])
    engines.append((name, eng))
    sweep = lambda k: eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    timed(sweep, 6)
    t = timed(sweep, 20)
    tb = timed(lambda k: eng.sweep_bin(x[0], 0, n), 10)
    info = eng.plan_info()
    print(f"{name:20s} bin kernel alone {tb:.3f} ms  sweep {t:.3f} ms  (draws {info['draws_timed']}, best {info['draw_best_us']} us, from arena {info['value_stream_from_arena']})", flush=True)
