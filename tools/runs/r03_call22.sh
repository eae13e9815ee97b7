#!/bin/bash
# round 3, GPU call 22: which arena user produces NaN at scale 21
export TMPDIR=/tmp
for env in "GM_ARENA_SITES=1" "GM_ARENA_SITES=2" "GM_ARENA_SITES=4" "GM_ARENA_SITES=8" "GM_ARENA_SITES=15"; do
env $env timeout 300 python tools/parity_pagerank.py --scale 21 --mode pb --iterations 5 --tolerance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env: max rel', d['max_rel_vs_reference'], 'rows over', d['rows_over_1e-5'])"
done
