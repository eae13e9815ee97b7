#!/usr/bin/env python3
"""The value stream (and optionally the plan's two index streams) mapped from a pseudo-random subset of a pool of
physical pieces (GM_PB_SPREAD / GM_PB_SPREAD_PLAN = "<MiB>,<pool factor>,<seed>"): distribution of the sweep time over
piece sizes, pool factors and seeds, against plain hipMalloc draws.   usage: placement9.py [scale]"""
import os, sys, time
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
variants = [("malloc", {})]
for mib, factor in ((16, 8), (32, 8), (64, 1), (64, 4), (64, 8), (64, 32), (128, 8), (256, 8), (512, 8)):
    for seed in (1, 2, 3):
        variants.append((f"spread {mib} MiB x{factor} seed {seed}", {"GM_PB_SPREAD": f"{mib},{factor},{seed}"}))
for seed in (1, 2, 3):
    variants.append((f"spread 64x8 + plan seed {seed}", {"GM_PB_SPREAD": f"64,8,{seed}", "GM_PB_SPREAD_PLAN": f"64,8,{seed}",
                                                         "GM_PB_NOCACHE": "1"}))
variants += [("malloc", {}), ("malloc", {})]
for name, env in variants:
    for k in ("GM_PB_SPREAD", "GM_PB_SPREAD_PLAN", "GM_PB_NOCACHE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    t_create = (time.perf_counter() - t0) * 1e3
    eng.init(sc, x[0])
    sweep = lambda k: eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    timed(sweep, 6)
    t = timed(sweep, 30)
    tb = timed(lambda k: eng.sweep_bin(x[0], 0, n), 12)
    print(f"{name:34s} sweep {t:.3f}  bin {tb:.3f}  rest {t - tb:.3f}   (engine created in {t_create:.0f} ms)", flush=True)
    del eng
