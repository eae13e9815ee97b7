#!/bin/bash
# round 4, GPU call 23: hub groups' hot terms off the value stream (hot records + gathers from hot_x): hub tests, bench A/B
OUT=gpurun_out/r04u; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest.txt 2>&1; grep -a "passed\|failed" $OUT/pytest.txt | tail -2; grep -a -B30 "Error\|assert " $OUT/pytest.txt | head -60 | cut -c1-200
for hh in 1 0 1 0; do
  GM_PB_HUB_HOT=$hh timeout 200 python bench.py --cpu-sweeps 0 --algos 0 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); c = d['config']; print('hub hot $hh:', d['ms_per_step'], d['roofline']['frac'], c['value_stream_placement'].get('draw_best_us'), c['value_entries'], c['hub_rows_in_reference_order']['hub_hot_edges'], c['plan_build_ms'], c['plan_rebuild_ms'])"
done
