#!/usr/bin/env python3
"""device CSR build at scale 21 with the arena's temporaries against the oracle's build"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from graph_amd import prelude as P, synth
from oracle import oracle as O
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
s, d = src.cpu().numpy().view(np.uint32), dst.cpu().numpy().view(np.uint32)
for direction, oname in ((P.Direction.Outgoing, O.OUTGOING), (P.Direction.Incoming, O.INCOMING)):
    for rep in range(3):
        g = synth.build_csr(n, src, dst, direction, P.CsrLayout.Sorted)
        off, tgt, _ = g.host()
        roff, rtgt = O.csr_build(n, s, d, oname, O.SORTED)
        bad_off = int((off != roff).sum()); bad_tgt = int((tgt != rtgt).sum())
        first = int(np.flatnonzero(tgt != rtgt)[0]) if bad_tgt else -1
        print(f"direction {direction} build {rep}: offsets differing {bad_off}, targets differing {bad_tgt} (first at {first} of {tgt.size})", flush=True)
        del g
