#!/usr/bin/env python3
"""Repeats delta_stepping on one RMAT graph (several start nodes and deltas) and checks that every run gives
the bits of the first one: the schedule is racy by design (atomicMin / flag protocol), the result must not be."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from graph_amd import prelude as P, synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << scale
src, dst = synth.rmat_edges(scale, 42)
w = synth.rmat_weights(src.numel(), 44)
g_out = synth.build_csr(n, src, dst, P.Direction.Outgoing, P.CsrLayout.Sorted, w)
g = P.DirectedCsrGraph(g_out, g_out, P.CsrLayout.Sorted)
deg = g_out.degrees()
starts = [int(np.flatnonzero(deg > 0)[0]), int(np.argmax(deg)), int(np.flatnonzero(deg > 0)[-1])]
bad = 0
for start in starts:
    for delta in (0.1, 0.5):
        ref = P.delta_stepping(g, P.DeltaSteppingConfig(start, delta))
        for r in range(reps):
            got = P.delta_stepping(g, P.DeltaSteppingConfig(start, delta))
            if not np.array_equal(got, ref):
                bad += 1
                print(f"MISMATCH start={start} delta={delta} rep={r}: {int((got != ref).sum())} distances differ", flush=True)
        print(f"start {start} delta {delta}: reached {int((ref < 3e38).sum())}, {reps} repeats identical" if not bad else "...", flush=True)
print("FAILED" if bad else "ALL IDENTICAL")
sys.exit(1 if bad else 0)
