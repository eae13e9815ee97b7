#!/bin/bash
# round 3, GPU call 11: automatic tier count, adaptive arena growth: tiers.py at three scales, PageRank parity tests, two bench runs
OUT=gpurun_out/r03k; mkdir -p $OUT; export TMPDIR=/tmp
GM_LOG=1 timeout 600 python tools/tiers.py 26 0 1 8 16 > $OUT/tiers26.txt 2> $OUT/tiers26.err; cat $OUT/tiers26.txt; grep -E "draw" $OUT/tiers26.err | head -12
timeout 300 python tools/tiers.py 24 0 1 > $OUT/tiers24.txt 2>&1; cat $OUT/tiers24.txt
timeout 300 python tools/tiers.py 22 0 1 > $OUT/tiers22.txt 2>&1; cat $OUT/tiers22.txt
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
