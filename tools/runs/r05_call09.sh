#!/bin/bash
# round 5, GPU call 9: hub rows of Unsorted layouts in CSR order
OUT=gpurun_out/r05i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py -x -q -m gpu -s > $OUT/pytest.txt 2>&1; grep -a "Unsorted\|passed\|failed\|Error\|error\|assert" $OUT/pytest.txt | cut -c1-300 | tail -12
timeout 900 python tools/parity_pagerank.py --scale 24 --layout unsorted --mode pb > $OUT/parity24_unsorted.json 2> $OUT/parity24.err; python -c "
import json; d=json.loads(open('$OUT/parity24_unsorted.json').read().strip().splitlines()[-1]); print('scale 24 unsorted:', d['max_rel_vs_reference'], d['rows_over_1e-5'], d['device'], [ (c['in_degree'], c['max_rel']) for c in d['by_in_degree']])" || tail -5 $OUT/parity24.err
