#!/usr/bin/env python3
"""Unsorted layout at scale S: the PB engine's hub rows against the REFORDER engine (every row left to right in f32, one lane
per row) sweep by sweep ON THE SAME INPUTS, until they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from graph_amd import prelude as P, synth
from graph_amd.engine import PageRankEngine
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 130
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
od = torch.bincount(src, minlength=n).to(torch.int32)
inc = synth.build_csr(n, src, dst, P.Direction.Incoming, P.CsrLayout.Unsorted)
del src, dst
deg = inc.degrees() if hasattr(inc, "degrees") else None
ioff = inc.host()[0]
deg = torch.from_numpy(np.diff(ioff.astype(np.int64))).cuda()
hub = deg >= 4096
pb = PageRankEngine(inc.handle, n, 0, od, 0.85, engine=PageRankEngine.PB)
ro = PageRankEngine(inc.handle, n, 0, od, 0.85, engine=PageRankEngine.REFORDER)
sc = torch.zeros(n, device="cuda"); x = [torch.zeros(n, device="cuda") for _ in range(2)]; err = torch.zeros(1, dtype=torch.float64, device="cuda")
sc2 = torch.zeros(n, device="cuda"); x2 = torch.zeros(n, device="cuda"); err2 = torch.zeros(1, dtype=torch.float64, device="cuda")
pb.init(sc, x[0])
for k in range(sweeps):
    sc2.copy_(sc)
    ro.sweep(x[k % 2], x2, sc2, err2)
    pb.sweep(x[k % 2], x[1 - k % 2], sc, err)
    bad = torch.nonzero(hub & (sc != sc2)).flatten()
    if bad.numel():
        print(f"sweep {k}: {bad.numel()} hub rows differ:", [(int(r), int(deg[r]), float(sc[r]), float(sc2[r])) for r in bad[:10].tolist()], flush=True)
        if bad.numel() > 0 and k > 60:
            break
    elif k % 10 == 0:
        print(f"sweep {k}: hub rows equal, error {float(err.item()):.3e}", flush=True)
print("done", float(err.item()))
