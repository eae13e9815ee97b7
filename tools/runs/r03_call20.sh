#!/bin/bash
# round 3, GPU call 20: PageRank against the oracle at scale 21 / 22 / 24, tiers automatic and forced to one
OUT=gpurun_out/r03t; mkdir -p $OUT; export TMPDIR=/tmp
for sc in 21 22 24; do for t in 0 1; do
GM_PB_TIERS=$t timeout 300 python tools/parity_pagerank.py --scale $sc --mode pb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale $sc tiers $t: max rel', d['max_rel_vs_reference'], 'rows over', d['rows_over_1e-5'], 'device iterations', d['device']['iterations'], 'worst', d['worst_rows'][:2])"
done; done
