#!/bin/bash
# round 3, GPU call 26: arena with its own address space: parity at scale 21 / 22 / 24, then the whole GPU suite
OUT=gpurun_out/r03v; mkdir -p $OUT; export TMPDIR=/tmp
for sc in 21 22 24; do
timeout 300 python tools/parity_pagerank.py --scale $sc --mode pb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scale $sc: max rel', d['max_rel_vs_reference'], 'rows over', d['rows_over_1e-5'], 'device iterations', d['device']['iterations'])"
done
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|^real|^E  " $OUT/pytest_gpu.log | tail -12
