#!/bin/bash
export TMPDIR=/tmp
for env in "GM_POISON=0,99999999999" "GM_POISON=0,1000000" "GM_POISON=1000001,16000000" "GM_POISON=16000001,99999999999"; do
env GM_ARENA=0 $env timeout 300 python tools/parity_pagerank.py --scale 21 --mode pb --iterations 5 --tolerance 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env: max rel', d['max_rel_vs_reference'], 'rows over', d['rows_over_1e-5'])"
done
