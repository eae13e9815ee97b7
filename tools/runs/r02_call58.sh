#!/bin/bash
# round 2, GPU call 58: plan build after the keys-kernel and sampling changes; PageRank tests; sweep time / hot coverage
OUT=gpurun_out/r02be; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "page_rank or pagerank or pb or hub or multi or smoke or prelude" > $OUT/pytest_pr.log 2>&1; grep -a "passed\|failed" $OUT/pytest_pr.log | tail -2
for k in 1 2; do
GM_LOG=1 timeout 600 python bench.py --cpu-sweeps 0 > $OUT/b$k.json 2> $OUT/b$k.err
python -c "
import json; d=json.loads(open('$OUT/b$k.json').read().strip().splitlines()[-1]); c=d['config']; print('run $k', d['ms_per_step'], 'frac', d['roofline']['frac'], 'plan', c['plan_build_ms'], c['plan_rebuild_ms'], 'hot', c['hot_sources'], 'entries', c['value_entries'])"
grep -a "pb plan" $OUT/b$k.err | sed -n '12,24p' | cut -c16-90
done
