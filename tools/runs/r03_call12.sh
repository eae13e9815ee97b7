#!/bin/bash
# round 3, GPU call 12: hot batches pipelined across tier switches: tiers.py at scale 26 / 24, parity tests, two bench runs
OUT=gpurun_out/r03l; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/tiers.py 26 1 0 8 12 16 1 > $OUT/tiers26.txt 2>&1; cat $OUT/tiers26.txt
timeout 300 python tools/tiers.py 24 1 0 8 > $OUT/tiers24.txt 2>&1; cat $OUT/tiers24.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "page_rank" --ignore=tests/test_gpu_fullsize.py > $OUT/pytest_pr.log 2>&1; grep -E "passed|failed" $OUT/pytest_pr.log
for k in 1 2; do
GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/bench$k.err | tail -1 > $OUT/bench$k.json
python - <<PY
import json
d=json.load(open('$OUT/bench$k.json')); c=d['config']
print('run $k', d['ms_per_step'], d['roofline']['frac'], 'plan_build_ms', c['plan_build_ms'], 'rebuild', c['plan_rebuild_ms'], 'tiers', c['hot_tiers'], 'hot', c['hot_sources'], c['hot_edges'], 'stream', c['value_entries'])
PY
grep -E "draw" $OUT/bench$k.err | head -4
done
