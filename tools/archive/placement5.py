#!/usr/bin/env python3
"""Where do the two sweep-time levels come from?  One process, one graph and plan; one engine per VARIANT of how the
3.6 GB value stream is allocated — hipMalloc (several draws, padding allocations in between) and virtual ranges mapped
from physical pieces of 2 MiB ... one piece (GM_PB_VALS_VMM, HIP virtual-memory API), in order or shuffled.  Per engine:
ms per sweep and ms of the bin kernel alone.

usage: placement5.py [scale] [--pmc]     (--pmc: two sweeps per engine and nothing else — run under rocprofv3 --pmc;
                                          the dispatches of engine k are the k-th pair of every kernel name)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
args = [a for a in sys.argv[1:] if not a.startswith("--")]
pmc = "--pmc" in sys.argv
scale = int(args[0]) if args else 26
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
torch.cuda.empty_cache()
M = "GM_PB_VALS_VMM"
variants = [("malloc-a", {}), ("malloc-b", {}), ("vmm-2M", {M: "2"}), ("vmm-2M-shuf", {M: "2", M + "_SHUFFLE": "7"}),
            ("vmm-32M", {M: "32"}), ("vmm-1G", {M: "1024"}), ("vmm-one", {M: "-1"}),
            ("vmm-2M-va1G", {M: "2", M + "_ALIGN": "1024"}), ("vmm-one-va1G", {M: "-1", M + "_ALIGN": "1024"}),
            ("contig", {"GM_PB_VALS_CONTIG": "1"}), ("malloc-c", {}), ("malloc-d", {})]
if pmc:
    variants = [v for v in variants if v[0] in ("malloc-a", "malloc-b", "vmm-2M", "vmm-one", "contig", "malloc-c")]
keys = sorted({k for _, env in variants for k in env})
sets = []
for name, env in variants:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85)
    x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.init(sc, x[0])
    pad = torch.empty(int(1.3e9), dtype=torch.uint8, device="cuda")  # shifts where the next engine's buffers land
    sets.append((name, eng, x, sc, err, pad))
torch.cuda.synchronize()
def timed(fn, reps):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
if pmc:
    for name, eng, x, sc, err, _ in sets:
        for k in range(2):
            eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        torch.cuda.synchronize()
        print("pmc engine", name, flush=True)
    sys.exit(0)
for rnd in range(2):
    for name, eng, x, sc, err, _ in sets:
        sweep = lambda k: eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        timed(sweep, 8)
        t = timed(sweep, 60)
        tb = timed(lambda k: eng.sweep_bin(x[0], 0, n), 30)
        print(f"round {rnd} {name:14s} sweep {t:.3f} ms   bin kernel alone {tb:.3f} ms", flush=True)
