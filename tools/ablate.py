#!/usr/bin/env python3
"""What bounds the two propagation-blocking kernels?  One resident graph, the sweep timed under each
GM_PB_ABLATE measurement variant (pagerank_pb.hip; variants other than 0 compute wrong results
by design).  Usage: tools/ablate.py [scale] [codes...]; the product (0) is re-timed between variants
because the clock drifts over a run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
# the variants live in the MEASUREMENT library only (make -C graph_amd/csrc measure: the sources with -DGM_MEASURE)
import subprocess
subprocess.check_call(["make", "-C", os.path.join(ROOT, "graph_amd", "csrc"), "measure", "-j8"], stdout=subprocess.DEVNULL)
os.environ["GRAPH_MI355X_LIB"] = os.path.join(ROOT, "graph_amd", "libgraph_mi355x_measure.so")
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
codes = [int(c) for c in sys.argv[2:]] or [1, 3, 4, 5, 30, 40, 50, 60]
NAMES = {0: "product", 1: "bin: no LDS gather", 3: "bin: no stores", 4: "bin: no x tile load", 5: "bin: x tile load only",
         30: "accum: no epilogue", 40: "accum: no streaming loops", 50: "accum: hot edges + epilogue only",
         60: "accum: value stream + epilogue only"}
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
del src, dst
torch.cuda.empty_cache()
eng = PageRankEngine(csr.handle, n, 0, od, 0.85, engine=2)
x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
sc = torch.zeros(n, device="cuda")
eng.init(sc, x[0])


def timed(code, reps=20):
    os.environ["GM_PB_ABLATE"] = str(code)
    for k in range(5):
        eng.sweep_tiles(x[k % 2], x[1 - k % 2], sc)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for k in range(reps):
        eng.sweep_tiles(x[k % 2], x[1 - k % 2], sc)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


for _ in range(8):
    timed(0)
for code in codes:
    base = timed(0)
    t = timed(code)
    base2 = timed(0)
    b = 0.5 * (base + base2)
    print(f"ablate {code:3d} {NAMES.get(code, '?'):32s}: {t:.4f} ms   product {b:.4f} ms   delta {t - b:+.4f} ms", flush=True)
del eng, csr
