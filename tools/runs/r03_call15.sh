#!/bin/bash
# round 3, GPU call 15: hub kernel replay for every group, virtual-rank exchange ordered like a collective: hub / multi tests, then bench twice
OUT=gpurun_out/r03o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_hub_adversarial.py tests/test_gpu_hub_order.py tests/test_gpu_multi.py -q -s > $OUT/pytest_hub.log 2>&1; grep -E "passed|failed" $OUT/pytest_hub.log | tail -3
grep -E "max rel|worst|virtual ranks|Unsorted|seed|emulated|long2" $OUT/pytest_hub.log | sed 's/^\.*//' | sort | uniq > $OUT/hub_numbers.txt; grep -E "seed|Unsorted|worst|scale|virtual" $OUT/hub_numbers.txt
grep -E "^E  |Error" $OUT/pytest_hub.log | head -12
for k in 1 2; do
timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/bench$k.err | tail -1 > $OUT/bench$k.json
python - <<PY
import json
d=json.load(open('$OUT/bench$k.json')); c=d['config']
print('run $k', d['ms_per_step'], d['roofline']['frac'], 'plan_build_ms', c['plan_build_ms'], 'tiers', c['hot_tiers'])
PY
done
