#!/bin/bash
# round 3, GPU call 21: which ingredient produces NaN at scale 21
OUT=gpurun_out/r03u; mkdir -p $OUT; export TMPDIR=/tmp
for env in "X=1" "GM_ARENA=0" "GM_PB_DRAWS=0" "GM_PB_HUB_DEG=0" "GM_PB_HUB_GRADED=0" "GM_PB_TIERS=1 GM_PB_HOT=0" "GM_PB_WGS=1"; do
env $env timeout 300 python tools/parity_pagerank.py --scale 21 --mode pb --iterations 5 --tolerance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env: max rel', d['max_rel_vs_reference'], 'rows over', d['rows_over_1e-5'], 'it', d['device']['iterations'])"
done
