#!/usr/bin/env python3
"""Does building the plan AGAIN change the level the bin kernel runs at?  One process, one resident graph, N private plans
(GM_PB_NOCACHE) one after the other, each with its engine: the placement draws' best bin-kernel time and the sweep time.
usage: placement11.py [scale=26] [plans=6] [keep=0|1: keep the earlier plans alive]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["GM_PB_NOCACHE"] = "1"
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
plans = int(sys.argv[2]) if len(sys.argv) > 2 else 6
keep = len(sys.argv) > 3 and sys.argv[3] == "1"
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
kept = []
for i in range(plans):
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85, engine=2)
    eng.init(sc, x[0])
    for k in range(6):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(20):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
    e1.record(); torch.cuda.synchronize()
    info = eng.plan_info()
    print(f"plan {i}: build {info['plan_build_us'] / 1e3:.0f} ms, draws {info['draws_timed']} best {info['draw_best_us']} us worst "
          f"{info['draw_worst_us']} us, grown {info['arena_grown_pieces']} pieces, sweep {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
    if keep:
        kept.append(eng)
    else:
        del eng
