#!/usr/bin/env python3
"""how many long chains there are and how many fall back to the sequential walk, per scale"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from graph_amd import synth
from graph_amd.engine import PageRankEngine
from graph_amd.prelude import CsrLayout, Direction
for scale in (20, 22, 24, 26):
    n = 1 << scale
    src, dst = synth.rmat_edges(scale, 42)
    od = torch.bincount(src, minlength=n).to(torch.int32)
    csr = synth.build_csr(n, src, dst, Direction.Incoming, CsrLayout.Sorted)
    del src, dst
    eng = PageRankEngine(csr.handle, n, 0, od, 0.85, engine=2)
    x = [torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")]
    sc = torch.zeros(n, device="cuda"); err = torch.zeros(1, dtype=torch.float64, device="cuda")
    eng.init(sc, x[0])
    out = []
    for k in range(12):
        eng.sweep(x[k % 2], x[1 - k % 2], sc, err)
        torch.cuda.synchronize()
        i = eng.plan_info()
        out.append(i["long_chains_fell_back"])
    print(f"scale {scale}: {i['hub_groups']} hub groups, {i['long_chain_groups']} of one or two rows with {i['long_chain_blocks']} blocks; "
          f"fell back per sweep: {out}", flush=True)
    del eng, csr
