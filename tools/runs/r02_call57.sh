#!/bin/bash
# round 2, GPU call 57: size of the hot-source block filter of the keys kernel (LDS per workgroup vs false positives)
OUT=gpurun_out/r02bd; mkdir -p $OUT; export TMPDIR=/tmp
for fb in 18 18; do
GM_LOG=1 GM_PB_NOCACHE=1 GM_PB_FILTER_BITS=$fb timeout 600 python - > $OUT/f$fb.log 2>&1 <<PY
import time, torch
from graph_amd import prelude as P, synth
sc=26; n=1<<sc
src,dst=synth.rmat_edges(sc,42)
g=P.DirectedCsrGraph(synth.build_csr(n,src,dst,P.Direction.Outgoing,P.CsrLayout.Sorted), synth.build_csr(n,src,dst,P.Direction.Incoming,P.CsrLayout.Sorted), P.CsrLayout.Sorted)
del src,dst
torch.cuda.empty_cache()
for k in range(2):
    r = P.page_rank(g, P.PageRankConfig(1, 0.0, 0.85), P.PageRankMode.JacobiPB)
print('CHECK', float(r[0].sum()), r[0][:3])
PY
echo "filter bits $fb: $(grep -a 'edge keys took' $OUT/f$fb.log | tail -1 | cut -c16-80) | $(grep -a CHECK $OUT/f$fb.log)"
done
