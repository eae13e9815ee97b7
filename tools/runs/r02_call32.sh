#!/bin/bash
# round 2, GPU call 32: plan phases of a second build in the same process (one-time costs excluded)
OUT=gpurun_out/r02ag; mkdir -p $OUT; export TMPDIR=/tmp
GM_LOG=1 GM_PB_NOCACHE=1 timeout 600 python - > $OUT/twice.log 2>&1 <<PY
import time, torch
from graph_amd import prelude as P, synth
sc=26; n=1<<sc
src,dst=synth.rmat_edges(sc,42)
g=P.DirectedCsrGraph(synth.build_csr(n,src,dst,P.Direction.Outgoing,P.CsrLayout.Sorted), synth.build_csr(n,src,dst,P.Direction.Incoming,P.CsrLayout.Sorted), P.CsrLayout.Sorted)
del src,dst
for k in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    P.page_rank(g, P.PageRankConfig(1, 0.0, 0.85), P.PageRankMode.JacobiPB)
    print('CALL', k, (time.perf_counter()-t)*1e3, 'ms', flush=True)
PY
grep -a "pb plan\|CALL" $OUT/twice.log
