#!/usr/bin/env python3
"""Does the per-PROCESS 'placement level' of the PageRank sweep (fast / medium / slow) depend on the STREAM the
sweep is launched on?  One engine at scale S; the same sweeps timed on the null stream, on fresh torch pool streams, on high-priority
streams: ms per sweep (HIP events on that stream), and the bin / accumulate split from a second event in the middle.
usage: stream_level_probe.py [scale]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth, prelude as P
from graph_amd.engine import PageRankEngine
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
inc = synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Sorted)
del src, dst
eng = PageRankEngine(inc.handle, n, 0, od, 0.85, engine=PageRankEngine.PB)
info = eng.plan_info()
dev = torch.device("cuda", 0)
x = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]
sc = torch.zeros(n, device=dev); err = torch.zeros(1, dtype=torch.float64, device=dev)
eng.init(sc, x[0])
torch.cuda.synchronize()
def timed(stream, sweeps=10):
    with torch.cuda.stream(stream):
        for k in range(3):
            eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(sweeps):
            eng.sweep(x[(k + 1) % 2], x[k % 2], sc, err)
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / sweeps
res = {"draw_best_us": info.get("draw_best_us"), "draws_timed": info.get("draws_timed")}
res["null"] = [round(timed(torch.cuda.default_stream()), 4) for _ in range(2)]
res["pool"] = [round(timed(torch.cuda.Stream()), 4) for _ in range(10)]
res["high_prio"] = [round(timed(torch.cuda.Stream(priority=-1)), 4) for _ in range(3)]
res["null_again"] = [round(timed(torch.cuda.default_stream()), 4)]
print(json.dumps(res), flush=True)
