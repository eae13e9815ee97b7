#!/bin/bash
# round 5, GPU call 23: the boundary sums of a long row's items read once by one thread: hub tests incl. the hopping-boundary
# stress, then the Unsorted scale-26 parity three times and the Sorted one
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed\|Error\|assert" $OUT/pytest.txt | tail -5
show() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['max_rel_vs_reference'], d['rows_over_1e-5'], d['device']['iterations'], d['device']['next_call_s'])"; }
for i in 1 2 3; do timeout 900 python tools/parity_pagerank.py --scale 26 --layout unsorted --mode pb 2>/dev/null | tee $OUT/parity26_unsorted_$i.json | show "scale 26 unsorted ($i):"; done
timeout 900 python tools/parity_pagerank.py --scale 26 --layout sorted --mode pb 2>/dev/null | show "scale 26 sorted:"
