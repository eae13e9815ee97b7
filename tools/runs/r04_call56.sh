#!/bin/bash
# round 4, GPU call 56: 2-byte hot records (GM_PB_HOT16=1): parity, then the sweep against 4-byte records
OUT=gpurun_out/r04zy; mkdir -p $OUT; export TMPDIR=/tmp
GM_PB_HOT16=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -x -q -m gpu -k "not sssp and not wcc and not triangle" > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2; grep -a "Error\|assert" $OUT/pytest.txt | head -5
line() { python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1:', d['ms_per_step'], d['roofline']['frac'], d['config']['value_stream_placement'].get('level'), 'plan', d['config']['plan_build_ms'], d['config']['plan_bytes'], 'parity', (d['config'].get('parity') or {}).get('max_rel_vs_reference'))"; }
for sc in 22 26; do for rep in 1 2; do for h in 0 1; do
  GM_PB_HOT16=$h timeout 300 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | line "scale $sc hot16 $h"
done; done; done
