#!/bin/bash
# round 4, GPU call 26: the hub groups' hot keys sorted behind the ordinary hot keys (no compaction pass): tests, plan time, bench
OUT=gpurun_out/r04x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_robustness.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2; grep -a -B30 "Error\|assert " $OUT/pytest.txt | head -50 | cut -c1-200
for rep in 1 2; do
  timeout 200 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('scale 26:', d['ms_per_step'], d['roofline']['frac'], c['value_stream_placement'].get('draw_best_us'), c['value_stream_placement'].get('level'), c['plan_build_ms'], c['plan_rebuild_ms'], c['hub_rows_in_reference_order']['hub_hot_edges'])"
done
for sc in 22 24; do timeout 200 python bench.py --cpu-sweeps 0 --algos 0 --scale $sc 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale $sc:', d['ms_per_step'], d['roofline']['frac'], d['config']['plan_rebuild_ms'])"; done
