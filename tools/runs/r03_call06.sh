#!/bin/bash
# round 3, GPU call 6: arena + spread streams + timed draws: parity tests of the PageRank / CSR / TC paths, then four
# fresh bench processes (sweep time, plan build time, draw times)
OUT=gpurun_out/r03f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "page_rank or csr or triangle or robust" --ignore=tests/test_gpu_fullsize.py > $OUT/pytest_subset.log 2>&1; tail -5 $OUT/pytest_subset.log
for k in 1 2 3 4; do
GM_LOG=1 timeout 300 python bench.py --cpu-sweeps 0 2> $OUT/bench$k.err | tail -1 > $OUT/bench$k.json
python - <<PY
import json
d=json.load(open('$OUT/bench$k.json')); c=d['config']
print('run $k', d['ms_per_step'], d['roofline']['frac'], 'plan_build_ms', c['plan_build_ms'], 'rebuild', c['plan_rebuild_ms'], 'csr_build_s', c['csr_build_s'])
PY
grep -E "draw|spread" $OUT/bench$k.err | head -8
done
